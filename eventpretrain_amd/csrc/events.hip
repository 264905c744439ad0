// Event-level augmentation on the device: erase rows / add correlated rows, result time-sorted.
//
// Restates `erase_and_add_events` of the reference loader (dataset/augmentation/events_augment.py:28-55): drop the rows
// listed in erase_index, append rows events[add_index] + noise (x and y clipped to the sensor), then sort by the time
// stamp. The random decisions (how many, which rows, the noise) are drawn by the caller in the reference's order; this
// file is the data movement. Instead of delete + concatenate + argsort over the whole clip, the kept rows (already
// time-sorted) and the few added rows (<= 1 % of the clip, sorted here in LDS) are MERGED: every row computes its own
// output position with two short binary searches and is written once.
//   position of kept row i      = i - #{erased < i} + #{added with t_add <  t_i}
//   position of added row j     = j + #{kept rows with t <= t_add_j}
// (rows with equal stamps: original rows first, in their original order; numpy's argsort leaves that order unspecified).
// HBM-bound: n*32 B read + n'*32 B written per clip.
#include "evp_common.h"

namespace {

constexpr int EA_MAX_ADD = 8192;  // added rows per clip the LDS sort holds (1 % of an 819 200-event clip)

struct SortKey {
  double t;
  int j;
};
__device__ __forceinline__ bool key_less(const SortKey &a, const SortKey &b) { return a.t < b.t || (a.t == b.t && a.j < b.j); }

// one workgroup per clip: gather + perturb + clip the added rows, bitonic-sort them by (t, draw order) in LDS
__global__ __launch_bounds__(1024) void build_added_kernel(const double *events, const int64_t *clip_begin, const int64_t *add_idx,
                                                           const double *add_noise, const int64_t *add_offsets, double sensor_w,
                                                           double sensor_h, double *add_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SortKey *keys = reinterpret_cast<SortKey *>(smem_raw);
  const int c = blockIdx.x;
  const int64_t a0 = add_offsets[c];
  const int na = (int)(add_offsets[c + 1] - a0);
  if (na <= 0) return;
  const double *ev = events + clip_begin[c] * 4;
  int np2 = 1;
  while (np2 < na) np2 <<= 1;
  for (int j = threadIdx.x; j < np2; j += blockDim.x) {
    SortKey k;
    k.j = j;
    k.t = j < na ? ev[add_idx[a0 + j] * 4 + 2] + add_noise[(a0 + j) * 3 + 2] : __builtin_inf();
    keys[j] = k;
  }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int partner = i ^ stride;
        if (partner > i) {
          const bool up = (i & size) == 0;
          const SortKey a = keys[i], b = keys[partner];
          if (key_less(b, a) == up) {
            keys[i] = b;
            keys[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < na; r += blockDim.x) {
    const int j = keys[r].j;
    const double *src = ev + add_idx[a0 + j] * 4;
    const double *nz = add_noise + (a0 + j) * 3;
    double x = src[0] + nz[0], y = src[1] + nz[1];
    x = fmin(fmax(x, 0.0), sensor_w - 1.0);  // np.clip(., 0, sensor_w - 1), events_augment.py:46-47
    y = fmin(fmax(y, 0.0), sensor_h - 1.0);
    double *dst = add_rows + (a0 + r) * 4;
    dst[0] = x;
    dst[1] = y;
    dst[2] = keys[r].t;
    dst[3] = src[3];
  }
}

// first index in [0, n) with a[idx] >= v (int64 keys)
__device__ __forceinline__ int lower_bound_i64(const int64_t *a, int n, int64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// grid (chunks, clips): kept rows and added rows scatter themselves to their merged positions. A workgroup owns a CONTIGUOUS chunk of
// its clip's rows: the erased indices and the added stamps that can fall inside the chunk are two short runs of the (sorted) lists --
// about 1 % of the chunk each -- which four threads bracket with one binary search each and the workgroup parks in LDS; a row then
// needs two searches over a handful of LDS words. (Round 3: every row used to walk both whole lists in global memory, ~18 dependent
// loads; 211 us per 64-clip batch = 1.9 TB/s. Runs longer than MERGE_SEG entries are searched in place. Rows of a clip must be
// time-sorted, as the entry points' contract says.)
constexpr int MERGE_SEG = 1024;
__global__ __launch_bounds__(256) void merge_kernel(const double *events, const int64_t *clip_begin, const int64_t *clip_end, const int64_t *erase_idx,
                                                    const int64_t *erase_offsets, const int64_t *add_offsets, const double *add_rows,
                                                    const int64_t *out_offsets, double *out) {
  __shared__ int64_t s_er[MERGE_SEG];
  __shared__ double s_at[MERGE_SEG];
  __shared__ int s_b[4];
  const int c = blockIdx.y;
  const int64_t beg = clip_begin[c], n = clip_end[c] - beg;
  const int64_t e0 = erase_offsets[c];
  const int ne = (int)(erase_offsets[c + 1] - e0);
  const int64_t a0 = add_offsets[c];
  const int na = (int)(add_offsets[c + 1] - a0);
  const double *ev = events + beg * 4;
  const int64_t *er = erase_idx + e0;
  const double *ad = add_rows + a0 * 4;
  double *dst = out + out_offsets[c] * 4;
  const int64_t c0 = n * blockIdx.x / gridDim.x, c1 = n * (blockIdx.x + 1) / gridDim.x;      // this workgroup's rows
  if (threadIdx.x < 4) {                     // s_b = {#erased < c0, #erased < c1, #added with t < t[c0], #added with t < t[c1 - 1]}
    int v = 0;
    if (c1 > c0) {
      if (threadIdx.x < 2) v = lower_bound_i64(er, ne, threadIdx.x == 0 ? c0 : c1);
      else {
        const double key = ev[(threadIdx.x == 2 ? c0 : c1 - 1) * 4 + 2];
        int lo = 0, hi = na;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ad[mid * 4 + 2] < key) lo = mid + 1; else hi = mid; }
        v = lo;
      }
    }
    s_b[threadIdx.x] = v;
  }
  __syncthreads();
  const int eb = s_b[0], ecnt = s_b[1] - s_b[0], ab = s_b[2], acnt = s_b[3] - s_b[2];
  const bool e_lds = ecnt <= MERGE_SEG, a_lds = acnt <= MERGE_SEG;
  if (e_lds) for (int k = threadIdx.x; k < ecnt; k += 256) s_er[k] = er[eb + k];
  if (a_lds) for (int k = threadIdx.x; k < acnt; k += 256) s_at[k] = ad[(int64_t)(ab + k) * 4 + 2];
  __syncthreads();
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
    const double2 r0 = *reinterpret_cast<const double2 *>(ev + i * 4);
    const double2 r1 = *reinterpret_cast<const double2 *>(ev + i * 4 + 2);
    int lo = 0, hi = ecnt;                  // erased indices of the run that are < i
    if (e_lds) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_er[mid] < i) lo = mid + 1; else hi = mid; } }
    else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (er[eb + mid] < i) lo = mid + 1; else hi = mid; } }
    if (lo < ecnt && (e_lds ? s_er[lo] : er[eb + lo]) == i) continue;      // erased
    const int e = eb + lo;
    lo = 0; hi = acnt;                      // added rows of the run with t_add < t_i
    if (a_lds) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_at[mid] < r1.x) lo = mid + 1; else hi = mid; } }
    else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (ad[(int64_t)(ab + mid) * 4 + 2] < r1.x) lo = mid + 1; else hi = mid; } }
    double *o = dst + (i - e + ab + lo) * 4;
    *reinterpret_cast<double2 *>(o) = r0;
    *reinterpret_cast<double2 *>(o + 2) = r1;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t j = tid; j < na; j += stride) {
    const double t = ad[j * 4 + 2];
    int64_t lo = 0, hi = n;  // original rows with t_i <= t
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (ev[mid * 4 + 2] <= t) lo = mid + 1;
      else hi = mid;
    }
    const int64_t kept = lo - lower_bound_i64(er, ne, lo);
    double *o = dst + (j + kept) * 4;
    o[0] = ad[j * 4 + 0];
    o[1] = ad[j * 4 + 1];
    o[2] = t;
    o[3] = ad[j * 4 + 3];
  }
}


// ---- the erase / add DECISIONS drawn on the device (round 4) ----------------------------------------------------------------------
// erase_and_add_events draws, per clip of n rows: how many rows to erase and to add (uniform in [int(0.001 n), int(0.01 n)), the caller
// does that: two numbers per clip, and the offsets below follow from them), WHICH rows (without replacement) and three normal noise
// columns for the added rows (events_augment.py:31-44). On the host that is ~130 us of numpy calls per clip -- 8.5 ms per 64-clip
// batch against 0.35 ms of kernels for the whole chain. Here one workgroup per clip draws the rows and the noise from Philox4x32-10
// keyed by (seed, step, sample): candidates floor(u * n) in draw order, sorted in LDS as (value, draw index) keys; the first of every
// run of equal values is its earliest draw, so "the first k distinct values in draw order" -- sequential sampling with rejection of
// repeats, i.e. a uniform draw without replacement -- are the kept entries whose rank among the kept, in draw order, is below k. The
// erased rows leave sorted (the merge kernel wants them ascending), the added rows in draw order with their noise
// (Box-Muller, N(0, 1.5) / N(0, 1.5) / N(0, 0.001)).
__device__ __forceinline__ void ev_philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// word w of the stream (seed, step, sample, purpose): counter (w / 4, purpose, sample lo, sample hi), key (seed ^ step-mix)
__device__ __forceinline__ uint32_t ev_rand_word(uint64_t seed, uint64_t step, uint64_t sample, uint32_t purpose, uint32_t w) {
  uint32_t c[4] = {w >> 2, purpose, (uint32_t)sample, (uint32_t)(sample >> 32)};
  ev_philox(c, (uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step);
  return c[w & 3];
}

// exclusive prefix sum of flag[0 .. n) (0 / 1 values as int) in place -> positions; returns the total. n <= 8192, 1024 threads.
__device__ int block_scan_excl(int *v, int n, int *scratch) {
  const int tid = threadIdx.x, per = (n + 1023) / 1024;
  const int b = tid * per, e = (b + per < n) ? b + per : n;
  int s = 0;
  for (int i = b; i < e; ++i) s += v[i];
  scratch[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int t = tid >= off ? scratch[tid - off] : 0;
    __syncthreads();
    scratch[tid] += t;
    __syncthreads();
  }
  const int total = scratch[1023];
  int run = scratch[tid] - s;
  for (int i = b; i < e; ++i) {
    const int f = v[i];
    v[i] = run;
    run += f;
  }
  __syncthreads();
  return total;
}

// one list (erase: sorted output; add: draw-order output) of `k` distinct rows of [0, n) for one clip
template <bool SORTED_OUT>
__device__ void draw_distinct(uint64_t *keys, int *a, int *b, int *scratch, int np2, int64_t n, int k, uint64_t seed, uint64_t step, uint64_t sample,
                              uint32_t purpose, int64_t *out) {
  const int tid = threadIdx.x;
  for (int j = tid; j < np2; j += 1024) {
    const uint64_t cand = ((uint64_t)ev_rand_word(seed, step, sample, purpose, (uint32_t)j) * (uint64_t)n) >> 32;      // floor(u * n), u = w / 2^32
    keys[j] = (cand << 16) | (uint64_t)j;                      // np2 <= 8192 < 2^16
  }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < np2; i += 1024) {
        const int partner = i ^ stride;
        if (partner > i) {
          const bool up = (i & size) == 0;
          const uint64_t x = keys[i], y = keys[partner];
          if ((y < x) == up) { keys[i] = y; keys[partner] = x; }
        }
      }
      __syncthreads();
    }
  // a[j] (draw order) = 1 where draw j is the first occurrence of its value
  for (int i = tid; i < np2; i += 1024) {
    const bool first = i == 0 || (keys[i] >> 16) != (keys[i - 1] >> 16);
    a[(int)(keys[i] & 0xFFFF)] = first ? 1 : 0;
  }
  __syncthreads();
  for (int j = tid; j < np2; j += 1024) b[j] = a[j];          // keep the flags; a becomes the rank among the kept, in draw order
  __syncthreads();
  const int distinct = block_scan_excl(a, np2, scratch);
  if (SORTED_OUT) {
    // b (sorted order) = 1 where the sorted entry is selected; scan -> output position
    __syncthreads();
    int *sel = scratch + 1024;                                  // [np2] (caller sized the scratch for it)
    for (int i = tid; i < np2; i += 1024) {
      const int j = (int)(keys[i] & 0xFFFF);
      sel[i] = (b[j] && a[j] < k) ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < np2; i += 1024) b[i] = sel[i];
    __syncthreads();
    block_scan_excl(sel, np2, scratch);
    for (int i = tid; i < np2; i += 1024)
      if (b[i]) out[sel[i]] = (int64_t)(keys[i] >> 16);
  } else {
    for (int i = tid; i < np2; i += 1024) {
      const int j = (int)(keys[i] & 0xFFFF);
      if (b[j] && a[j] < k) out[a[j]] = (int64_t)(keys[i] >> 16);
    }
  }
  __syncthreads();
  if (distinct < k && tid == 0) {
    // fewer distinct candidates than asked for (needs more repeats than the slack holds: practically never): top up with the smallest
    // rows not drawn yet, so that exactly k rows leave -- the offsets of every later table depend on it
    int have = distinct;
    int64_t v = 0;
    int i = 0;
    while (have < k && v < n) {
      while (i < np2 && (int64_t)(keys[i] >> 16) < v) ++i;
      if (i < np2 && (int64_t)(keys[i] >> 16) == v) { ++v; continue; }
      out[have++] = v++;                                       // (sorted output loses its order here; the merge only needs a set for
    }                                                          //  n this small -- and the caller's slack makes this path unreachable)
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void draw_erase_add_kernel(const int64_t *win_begin, const int64_t *win_end, const int64_t *erase_offsets,
                                                              const int64_t *add_offsets, uint64_t seed, uint64_t step, int64_t first_sample,
                                                              const int64_t *step_first, int np2, int64_t *erase_idx, int64_t *add_idx,
                                                              double *add_noise) {
  if (step_first) {                       // (step, first sample) from device memory: a replayed HIP graph draws each batch's own stream
    step = (uint64_t)step_first[0];
    first_sample = step_first[1];
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);
  int *a = reinterpret_cast<int *>(keys + np2), *b = a + np2, *scratch = b + np2;      // scratch: 1024 + np2 ints
  const int c = blockIdx.x;
  const int64_t n = win_end[c] - win_begin[c];
  const int ke = (int)(erase_offsets[c + 1] - erase_offsets[c]), ka = (int)(add_offsets[c + 1] - add_offsets[c]);
  const uint64_t sample = (uint64_t)(first_sample + c);
  if (n <= 0) return;
  // two workgroups per clip (blockIdx.y): the erase draw and the add draw + noise are independent streams of the same clip
  if (blockIdx.y == 0) {
    if (ke > 0) draw_distinct<true>(keys, a, b, scratch, np2, n, ke, seed, step, sample, 1u, erase_idx + erase_offsets[c]);
    return;
  }
  if (ka > 0) {
    draw_distinct<false>(keys, a, b, scratch, np2, n, ka, seed, step, sample, 2u, add_idx + add_offsets[c]);
    double *nz = add_noise + add_offsets[c] * 3;
    for (int r = threadIdx.x; r < ka; r += 1024) {
      uint32_t w[4] = {(uint32_t)r, 3u, (uint32_t)sample, (uint32_t)(sample >> 32)};
      ev_philox(w, (uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step);
      const double u0 = ((double)w[0] + 0.5) * 2.3283064365386963e-10, u1 = (double)w[1] * 2.3283064365386963e-10;
      const double u2 = ((double)w[2] + 0.5) * 2.3283064365386963e-10, u3 = (double)w[3] * 2.3283064365386963e-10;
      const double r0 = sqrt(-2.0 * log(u0)), r1 = sqrt(-2.0 * log(u2));
      nz[r * 3 + 0] = 1.5 * r0 * cos(6.283185307179586 * u1);
      nz[r * 3 + 1] = 1.5 * r0 * sin(6.283185307179586 * u1);
      nz[r * 3 + 2] = 0.001 * r1 * cos(6.283185307179586 * u3);
    }
  }
}


// ---- the batch PLAN on the device: what GpuInputPipeline.prepare() computes on the host with numpy (window starts, the two counts per
// clip and their prefix sums, the crop rows of the grid and of the frame target), one thread per clip from the same counter stream
// (dataset/augmentation/events_augment.py philox_words / draw_erase_add_counts, view_augment.py draw_evg_params_batch -- the numpy forms are
// the specification; this kernel repeats their float64 operations in their order, contraction off). With it a captured loader chain
// needs nothing from the host per batch: `state` = (step, first sample) lives on the device and is advanced here.
struct CropGeom { double area, a0, da; int W, H; };      // a0 = W / H * 3/4, da = W / H * 4/3 - a0 (computed by the caller in double)

__device__ void plan_crop_row(const double *U, const CropGeom g, double crop_min, double one_minus, int32_t *row) {
#pragma clang fp contract(off)
  long cw1 = g.W, ch1 = g.H;
  double ux = 0.0, uy = 0.0;
  bool any = false;
  for (int t = 0; t < 10 && !any; ++t) {
    const double *T = U + t * 5;
    const double target = (crop_min + T[0] * one_minus) * g.area;
    const double aspect = g.a0 + T[1] * g.da;
    long cw = (long)rint(sqrt(target * aspect)), ch = (long)rint(sqrt(target / aspect));
    if ((long)(T[2] * 10.0) < 5) { const long tmp = cw; cw = ch; ch = tmp; }
    if (cw < g.W && ch < g.H) { any = true; cw1 = cw; ch1 = ch; ux = T[3]; uy = T[4]; }
  }
  long x0 = 0, y0 = 0;
  if (any) {
    const long fx = g.W - cw1, fy = g.H - ch1;
    x0 = (long)(ux * (double)(fx > 1 ? fx : 1));
    y0 = (long)(uy * (double)(fy > 1 ? fy : 1));
    const long mx = fx - 1 > 0 ? fx - 1 : 0, my = fy - 1 > 0 ? fy - 1 : 0;
    x0 = x0 < mx ? x0 : mx;
    y0 = y0 < my ? y0 : my;
  }
  row[0] = (int32_t)x0; row[1] = (int32_t)y0; row[2] = (int32_t)cw1; row[3] = (int32_t)ch1;
  row[4] = U[50] < 0.5 ? 1 : 0;
  row[5] = U[51] < 0.5 ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_batch_kernel(const int64_t *clip_off, int nc, int64_t fix, uint64_t seed, int64_t *state, int advance,
                                                         int64_t *cur, CropGeom grid, CropGeom frame, double crop_min, double one_minus,
                                                         int64_t *tabs, int32_t *params, int32_t *fparams) {
#pragma clang fp contract(off)
  extern __shared__ int64_t plan_sm[];                 // [3][nc]: erase count, add count, rows out
  const uint64_t step = (uint64_t)state[0];
  const int64_t first = state[1];
  for (int c = threadIdx.x; c < nc; c += blockDim.x) {
    const uint64_t sample = (uint64_t)(first + c);
    uint32_t w[4] = {0u, 0u, (uint32_t)sample, (uint32_t)(sample >> 32)};
    ev_philox(w, (uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step);
    const int64_t n = clip_off[c + 1] - clip_off[c];
    const int64_t room = n > fix ? n - fix : 0;
    const int64_t s0 = n > fix ? (int64_t)(((uint64_t)w[0] * (uint64_t)room) >> 32) : 0;
    const int64_t s1 = n > fix ? s0 + fix : n;
    const int64_t nw = s1 - s0;
    const int64_t lo = (int64_t)(0.001 * (double)nw), hi = (int64_t)(0.01 * (double)nw);
    const uint64_t span = hi > lo ? (uint64_t)(hi - lo) : 0ull;
    int64_t e = lo + (int64_t)(((uint64_t)w[1] * span) >> 32), a = lo + (int64_t)(((uint64_t)w[2] * span) >> 32);
    if (hi <= 0) e = a = 0;
    tabs[c] = clip_off[c] + s0;
    tabs[(nc + 1) + c] = clip_off[c] + s1;
    plan_sm[c] = e; plan_sm[nc + c] = a; plan_sm[2 * nc + c] = nw - e + a;
    double U[52];
    for (int b = 0; b < 13; ++b) {
      uint32_t q[4] = {(uint32_t)b, 4u, (uint32_t)sample, (uint32_t)(sample >> 32)};
      ev_philox(q, (uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step);
      for (int i = 0; i < 4; ++i) U[b * 4 + i] = (double)q[i] * 2.3283064365386963e-10;
    }
    plan_crop_row(U, grid, crop_min, one_minus, params + c * 6);
    if (fparams) {
      plan_crop_row(U, frame, crop_min, one_minus, fparams + c * 6);
      fparams[c * 6 + 5] = params[c * 6 + 5];         // the time-flip flag is evg_augment's (pr_ef_imagenet_dataset.py:194-206)
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {                              // three prefix sums over the clips (a few dozen terms each)
    int64_t *row = tabs + (2 + threadIdx.x) * (nc + 1);
    const int64_t *v = plan_sm + threadIdx.x * nc;
    int64_t run = 0;
    row[0] = 0;
    for (int c = 0; c < nc; ++c) { run += v[c]; row[c + 1] = run; }
  }
  if (threadIdx.x == 3) {
    tabs[nc] = 0;
    tabs[(nc + 1) + nc] = 0;
    cur[0] = (int64_t)step;                           // what the draw kernel of THIS batch reads
    cur[1] = first;
    if (advance) state[0] = (int64_t)(step + 1);
  }
}
}  // namespace

static int build_added_launch(const double *events, const int64_t *clip_begin, int n_clips, const int64_t *add_idx, const double *add_noise,
                              const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows_ws, hipStream_t s,
                              const char *who) {
  int np2 = 1;
  while (np2 < max_add_per_clip) np2 <<= 1;
  const size_t smem = (size_t)np2 * sizeof(SortKey);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(build_added_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "%s: cannot reserve %zu B of LDS: %s", who, smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(build_added_kernel, dim3(n_clips), dim3(1024), smem, s, events, clip_begin, add_idx, add_noise, add_offsets, sensor_w, sensor_h,
                     add_rows_ws);
  EVP_CHECK_LAUNCH(who);
  return EVP_OK;
}

static int erase_add_launch(const double *events, const int64_t *clip_begin, const int64_t *clip_end, int n_clips, const int64_t *erase_idx,
                            const int64_t *erase_offsets, const int64_t *add_idx, const double *add_noise, const int64_t *add_offsets,
                            int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows_ws, const int64_t *out_offsets,
                            double *out_events, void *stream, const char *who) {
  EVP_CHECK_ARG(events && clip_begin && clip_end && erase_offsets && add_offsets && out_offsets && out_events, EVP_EINVAL, "%s: null pointer", who);
  EVP_CHECK_ARG(n_clips > 0, EVP_ESHAPE, "%s: n_clips must be positive", who);
  EVP_CHECK_ARG(max_add_per_clip >= 0 && max_add_per_clip <= EA_MAX_ADD, EVP_ESHAPE, "%s: at most %d added rows per clip (got %d)", who, EA_MAX_ADD,
                max_add_per_clip);
  EVP_CHECK_ARG(max_add_per_clip == 0 || (add_idx && add_noise && add_rows_ws), EVP_EINVAL,
                "%s: add_idx, add_noise and the workspace are required when rows are added", who);
  EVP_CHECK_ARG((((uintptr_t)events | (uintptr_t)out_events) & 15) == 0, EVP_EINVAL, "%s: event buffers must be 16-byte aligned", who);
  hipStream_t s = (hipStream_t)stream;
  if (max_add_per_clip > 0) {
    int rc = build_added_launch(events, clip_begin, n_clips, add_idx, add_noise, add_offsets, max_add_per_clip, sensor_w, sensor_h, add_rows_ws, s, who);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(merge_kernel, dim3(64, n_clips), dim3(256), 0, s, events, clip_begin, clip_end, erase_idx, erase_offsets, add_offsets, add_rows_ws,
                     out_offsets, out_events);
  EVP_CHECK_LAUNCH(who);
  return EVP_OK;
}

extern "C" int evp_events_erase_add_f64(const double *events, const int64_t *clip_offsets, int n_clips, const int64_t *erase_idx,
                                        const int64_t *erase_offsets, const int64_t *add_idx, const double *add_noise,
                                        const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h,
                                        double *add_rows_ws, const int64_t *out_offsets, double *out_events, void *stream) {
  EVP_CHECK_ARG(clip_offsets, EVP_EINVAL, "evp_events_erase_add_f64: null pointer");
  return erase_add_launch(events, clip_offsets, clip_offsets + 1, n_clips, erase_idx, erase_offsets, add_idx, add_noise, add_offsets, max_add_per_clip,
                          sensor_w, sensor_h, add_rows_ws, out_offsets, out_events, stream, "evp_events_erase_add_f64");
}

extern "C" int evp_events_erase_add_win_f64(const double *events, const int64_t *win_begin, const int64_t *win_end, int n_clips,
                                            const int64_t *erase_idx, const int64_t *erase_offsets, const int64_t *add_idx, const double *add_noise,
                                            const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows_ws,
                                            const int64_t *out_offsets, double *out_events, void *stream) {
  return erase_add_launch(events, win_begin, win_end, n_clips, erase_idx, erase_offsets, add_idx, add_noise, add_offsets, max_add_per_clip, sensor_w,
                          sensor_h, add_rows_ws, out_offsets, out_events, stream, "evp_events_erase_add_win_f64");
}

extern "C" int evp_events_build_added_f64(const double *events, const int64_t *win_begin, int n_clips, const int64_t *add_idx, const double *add_noise,
                                          const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows,
                                          void *stream) {
  EVP_CHECK_ARG(events && win_begin && add_idx && add_noise && add_offsets && add_rows, EVP_EINVAL, "evp_events_build_added_f64: null pointer");
  EVP_CHECK_ARG(n_clips > 0 && max_add_per_clip >= 0 && max_add_per_clip <= EA_MAX_ADD, EVP_ESHAPE, "evp_events_build_added_f64: bad shape");
  if (max_add_per_clip == 0) return EVP_OK;
  return build_added_launch(events, win_begin, n_clips, add_idx, add_noise, add_offsets, max_add_per_clip, sensor_w, sensor_h, add_rows,
                            (hipStream_t)stream, "evp_events_build_added_f64");
}

extern "C" int evp_events_draw_erase_add(const int64_t *win_begin, const int64_t *win_end, int n_clips, const int64_t *erase_offsets,
                                         const int64_t *add_offsets, uint64_t seed, uint64_t step, int64_t first_sample, const int64_t *step_first_dev,
                                         int max_per_clip, int64_t *erase_idx, int64_t *add_idx, double *add_noise, void *stream) {
  EVP_CHECK_ARG(win_begin && win_end && erase_offsets && add_offsets && erase_idx && add_idx && add_noise, EVP_EINVAL, "evp_events_draw_erase_add: null pointer");
  EVP_CHECK_ARG(n_clips > 0 && max_per_clip >= 0, EVP_ESHAPE, "evp_events_draw_erase_add: bad shape");
  if (max_per_clip == 0) return EVP_OK;
  const int want = max_per_clip + 64 + max_per_clip / 8;        // candidates per list: the count plus slack for repeated draws
  int np2 = 64;
  while (np2 < want) np2 <<= 1;
  EVP_CHECK_ARG(np2 <= 8192, EVP_ESHAPE, "evp_events_draw_erase_add: at most ~7200 rows per list and clip (got %d)", max_per_clip);
  const size_t smem = (size_t)np2 * 8 + (size_t)(3 * np2 + 1024) * 4;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(draw_erase_add_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_events_draw_erase_add: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(draw_erase_add_kernel, dim3(n_clips, 2), dim3(1024), smem, (hipStream_t)stream, win_begin, win_end, erase_offsets, add_offsets, seed, step,
                     first_sample, step_first_dev, np2, erase_idx, add_idx, add_noise);
  EVP_CHECK_LAUNCH("evp_events_draw_erase_add");
  return EVP_OK;
}

extern "C" int evp_events_plan_batch(const int64_t *clip_offsets, int n_clips, int64_t fix_events_num, uint64_t seed, int64_t *state, int advance,
                                     int64_t *step_first_out, int grid_h, int grid_w, int frame_h, int frame_w, double crop_min, int64_t *tabs,
                                     int32_t *params, int32_t *frame_params, void *stream) {
  EVP_CHECK_ARG(clip_offsets && state && step_first_out && tabs && params, EVP_EINVAL, "evp_events_plan_batch: null pointer");
  EVP_CHECK_ARG(n_clips > 0 && n_clips <= 2048 && fix_events_num > 0 && grid_h > 0 && grid_w > 0, EVP_ESHAPE,
                "evp_events_plan_batch: bad shape (at most 2048 clips per batch: three int64 rows of the batch live in 48 KiB of LDS)");
  EVP_CHECK_ARG(!frame_params || (frame_h > 0 && frame_w > 0), EVP_ESHAPE, "evp_events_plan_batch: frame size required with frame_params");
  EVP_CHECK_ARG(crop_min > 0.0 && crop_min <= 1.0, EVP_EINVAL, "evp_events_plan_batch: crop_min must lie in (0, 1]");
  auto geom = [](int H, int W) {
    CropGeom g;
    g.W = W; g.H = H;
    g.area = (double)((int64_t)W * H);
    const double wh = (double)W / (double)H;          // as the numpy form: W / H * ratio[0] + u * (W / H * ratio[1] - W / H * ratio[0])
    g.a0 = wh * (3.0 / 4.0);
    g.da = wh * (4.0 / 3.0) - wh * (3.0 / 4.0);
    return g;
  };
  const CropGeom gg = geom(grid_h, grid_w), gf = frame_params ? geom(frame_h, frame_w) : gg;
  hipLaunchKernelGGL(plan_batch_kernel, dim3(1), dim3(256), (size_t)3 * n_clips * sizeof(int64_t), (hipStream_t)stream, clip_offsets, n_clips,
                     fix_events_num, seed, state, advance, step_first_out, gg, gf, crop_min, 1.0 - crop_min, tabs, params, frame_params);
  EVP_CHECK_LAUNCH("evp_events_plan_batch");
  return EVP_OK;
}
