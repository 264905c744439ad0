// K1: event stream -> voxel grid (bilinear-in-time polarity histogram) for a batch of clips, gfx950.
//
// Restates dataset/dataset_utils/events_to_voxel_grid.py:4-61 of the reference (two index_add_ scatters) as
//   1. voxel_cuts_kernel : per clip, a 64-ary wave search over the (time-sorted) stamps for the first event with
//                          ts >= k, k = 0..bins  -> the contiguous event slab that can touch bin plane b.
//   2. voxel_bin_kernel  : workgroups of equal work per (clip, plane pair / plane, y-tile) -- see the schedule at the
//                          kernel. The tile of the bin plane lives in LDS (ds_add_f32), the workgroup streams its
//                          event slab with coalesced 32-byte rows, and the tile is flushed ONCE with coalesced
//                          16-byte stores -- no global atomics, no memset, every output byte written exactly once.
//      Blocks of one clip are mapped to one XCD (blockIdx % 8) so the slab re-reads (one per plane / y-tile) are
//      served by that XCD's L2, not by HBM.
// algo 1 keeps the plain global-atomic formulation (memset + 2 atomics per event), algo 2 a decode-once two-pass form
// (12-byte packed records); both measured slower than the default (5.3x and 1.3x) and stay for A/B measurements.
//
// HBM-bound. Algorithmic bytes per clip: n*32 B of events read + bins*H*W*4 B written.
#include "evp_common.h"

namespace {

struct Event { double x, y, t, p; };
// rows are (x,y,t,p), or (t,x,y,p) when is_txyp (events_to_voxel_grid.py:14-34)
// sx, sy: the sensor -> input rescale of the loader (events_augment.py:22-26), applied to x and y in float64 before the
// truncation exactly as `events[:, 0] *= input_w / sensor_w` does (1.0 = none; x * 1.0 is x)
__device__ __forceinline__ Event load_event(const double *ev, int64_t i, int is_txyp, double sx, double sy) {
  const double2 a = *reinterpret_cast<const double2 *>(ev + i * 4);
  const double2 b = *reinterpret_cast<const double2 *>(ev + i * 4 + 2);
  Event e;
  e.x = (is_txyp ? a.y : a.x) * sx;
  e.y = (is_txyp ? b.x : a.y) * sy;
  e.t = is_txyp ? a.x : b.x;
  e.p = b.y;
  return e;
}
__device__ __forceinline__ double stamp(const double *ev, int64_t i, int is_txyp) { return ev[i * 4 + (is_txyp ? 0 : 2)]; }

// ts exactly as events_to_voxel_grid.py:30/34 computes it (float64): (bins-1) * (t - t0) / dT
__device__ __forceinline__ double ts_of(double t, double t0, double dT, int bins) { return (double)(bins - 1) * (t - t0) / dT; }

// cuts[c][k], k = 0..bins: first row i (clip-relative) with ts_i >= k; cuts[c][bins+1] = n (unused sentinel)
// flags (verify mode, else NULL): flags[c] = 1 when the cuts partition the clip the way sorted stamps do (cuts[0] == 0, non-
// decreasing, cuts[bins] == n) -- the bin kernel then clears it if it meets a row outside the region its position says.
__global__ __launch_bounds__(512) void voxel_cuts_kernel(const double *events, const int64_t *offsets, int bins, int is_txyp,
                                                         int64_t *cuts, int32_t *flags) {
  const int c = blockIdx.x;
  const int64_t beg = offsets[c], n = offsets[c + 1] - beg;
  const double *ev = events + beg * 4;
  int64_t *out = cuts + (int64_t)c * (bins + 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (n <= 0) {
    for (int k = threadIdx.x; k <= bins + 1; k += blockDim.x) out[k] = 0;
    if (flags && threadIdx.x == 0) flags[c] = 1;
    return;
  }
  const double t0 = stamp(ev, 0, is_txyp), t1 = stamp(ev, n - 1, is_txyp);
  double dT = t1 - t0;
  if (dT == 0) dT = 1.0;
  for (int k = wave; k <= bins; k += nw) {
    int64_t lo = 0, hi = n;  // answer in [lo, hi]
    while (hi > lo) {
      const int64_t step = (hi - lo + 63) / 64;
      const int64_t idx = lo + (int64_t)lane * step;
      bool pred = false;
      if (idx < hi) pred = ts_of(stamp(ev, idx, is_txyp), t0, dT, bins) >= (double)k;
      const unsigned long long m = __ballot(pred);
      // lanes whose probe is out of range report false; they sit after every in-range probe
      if (m == 0ull) {
        const int64_t last = lo + ((hi - 1 - lo) / step) * step;  // last in-range probe
        lo = last + 1;
      } else {
        const int f = __ffsll((long long)m) - 1;
        const int64_t pf = lo + (int64_t)f * step;
        if (f > 0) lo = pf - step + 1;
        hi = pf;
      }
    }
    if (lane == 0) out[k] = lo;
  }
  if (threadIdx.x == 0) out[bins + 1] = n;
  if (flags) {
    __syncthreads();
    if (threadIdx.x == 0) {
      bool ok = out[0] == 0 && out[bins] == n;
      for (int k = 0; k < bins; ++k) ok = ok && out[k] <= out[k + 1];
      flags[c] = ok ? 1 : 0;
    }
  }
}

constexpr int VB_THREADS = 1024;
constexpr int VB_UNROLL = 4;

// a / d for the clip-constant divisor d, r = RN(1/d): two Newton corrections of q = a*r with exact FMA residuals, the
// last one is Markstein's final step (q faithful + r correctly rounded => RN(q + (a - q*d)*r) == RN(a/d)), so the
// result is the IEEE quotient the reference's float64 division produces -- at 5 full-rate FMAs instead of the ~14
// instruction v_div_scale/v_rcp_f64/v_div_fmas/v_div_fixup sequence (quarter-rate parts included) per event.
// Valid while nothing under- or overflows; callers route operands outside [2^-400, 2^400] to the plain division.
__device__ __forceinline__ double div_by_clip_constant(double a, double d, double r) {
  double q = a * r;
  double e = __builtin_fma(-q, d, a);
  q = __builtin_fma(e, r, q);
  e = __builtin_fma(-q, d, a);
  return __builtin_fma(e, r, q);
}
__device__ __forceinline__ bool in_safe_range(double v) {
  const double m = __builtin_fabs(v);
  return m >= 0x1p-400 && m <= 0x1p400;  // false for 0, NaN, inf, denormals
}

// Per event visit: (1) x, y -> flat pixel with 32-bit conversions and the y-tile test, which rejects about half of the
// visits after ~10 instructions; (2) only for events inside the tile the float64 time normalisation, with the division
// above. Measured: the kernel is bound by the bytes it pulls through L2 (every event row is visited by ~4 workgroups),
// not by this arithmetic, so the schedule below is what matters.
//
// Schedule. With sorted stamps the clip splits into regions R_k = [cuts[k], cuts[k+1]) (floor(ts) == k); plane b takes
// the left contributions of R_b and the right contributions of R_(b-1). A clip is served by (bins-1) * n_yt workgroups
// of EQUAL work (two regions each):
//   j = 0        : plane 0 over R_0, flush, then plane bins-1 over R_(bins-2) (+ the few rows with ts == bins-1)
//   j = 1..bins-2: plane j over R_(j-1) then R_j for odd j, R_j then R_(j-1) for even j (rows ascending in both)
// so that, for odd `bins`, the two planes that need a region stream it during the same half of their lifetime and the
// second reader finds the rows in the XCD's L2 instead of fetching them again (all blocks of a clip share an XCD).
// CELL: the accumulator type of the LDS tile: float (default) or double (algo 3, A/B). Round 3 found `ds_add_f32` to serialise its lanes
// on gfx950 (3 cycles per lane, 192 per full wave instruction; `ds_add_f64` ~22, `ds_add_u32` ~8 per wave instruction;
// tools/native/lds_atomic_probe.hip: 197 / 1431 / 2992 G lane-adds/s chip-wide) and SQ_LDS_IDX_ACTIVE at 74 % of this kernel's cycles --
// yet the atomics are NOT what the kernel waits for: without them it runs 101 -> 93 us, and with f64 cells (three y-tiles instead of two
// at 224 x 224, i.e. 6 instead of 4 visits per row) 124 us. What it does wait for is the rows themselves: "rows only loaded" is 76 of the
// 101 us (tools/voxel_parts.py) -- 820 MB through the L1s at 11 TB/s, 70 % of what the eight L2s deliver.
template <bool TXYP, typename CELL, bool DBG>
__global__ __launch_bounds__(VB_THREADS) void voxel_bin_kernel(const double *events, const int64_t *offsets, const int64_t *cuts,
                                                               int n_clips, int bins, int H, int W, int mode, int tile_rows,
                                                               int n_yt, double sx, double sy, float *out, int32_t *flags, int dbg_arg) {
  const int dbg = DBG ? dbg_arg : 0;          // the measurement build is a separate instantiation: the product kernel carries no test
  // dbg (measurement aid, evp_voxel_set_debug; results are garbage when set): 1 = no LDS atomics, 2 = no time normalisation (every row
  // in the tile adds 1), 3 = rows are only loaded
  // mode 0: any row order (every block scans its whole clip); 1: sorted stamps promised (slabs between the cuts);
  // 2: as 1, but every row that is normalised is checked to lie in the region its POSITION says (floor(ts) == k for a row
  //    of slab [cuts[k], cuts[k+1])) -- the only property of sortedness the slab schedule uses -- and flags[clip] is
  //    cleared otherwise; 3: repair pass = mode 0 for the clips whose flag is 0, nothing for the others.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  CELL *tile = reinterpret_cast<CELL *>(smem_raw);
  const int n_j = bins > 1 ? bins - 1 : 1;
  const int per_clip = n_j * n_yt;
  int clip, sub;
  if ((n_clips & 7) == 0) {  // keep one clip's blocks on one XCD (blocks b and b+8 share an XCD)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    clip = (slot / per_clip) * 8 + xcd;
    sub = slot % per_clip;
  } else {
    clip = blockIdx.x / per_clip;
    sub = blockIdx.x % per_clip;
  }
  if (mode >= 2) {
    const int ok = flags[clip];
    if (mode == 2 ? !ok : ok) return;      // 2: the cuts are no partition, the repair pass takes the clip; 3: nothing to repair
  }
  const bool assume_sorted = mode == 1 || mode == 2;
  bool inconsistent = false;
  const int j = sub / n_yt, yt = sub % n_yt;
  const int y0 = yt * tile_rows, y1 = (y0 + tile_rows < H) ? y0 + tile_rows : H;
  const int tile_elems = (y1 - y0) * W;
  const int64_t pix0 = (int64_t)y0 * W, pix1 = (int64_t)y1 * W;
  const int64_t beg = offsets[clip], n = offsets[clip + 1] - beg;
  const double *ev = events + beg * 4;
  const int64_t *cc = cuts + (int64_t)clip * (bins + 2);
  double t0 = 0.0, dT = 1.0;
  if (n > 0) {
    t0 = stamp(ev, 0, TXYP);
    dT = stamp(ev, n - 1, TXYP) - t0;
    if (dT == 0) dT = 1.0;
  }
  const double rT = 1.0 / dT, scale = (double)(bins - 1);
  const bool clip_fast = in_safe_range(dT);

  const int n_jobs = (j == 0 && bins > 1) ? 2 : 1;
  for (int job = 0; job < n_jobs; ++job) {
    const int b = job ? bins - 1 : j;
    const bool swapped = (j > 0) && ((j & 1) == 0);
    const double bd = (double)b;
    if (job) __syncthreads();  // the previous plane's flush has read the tile
    for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) tile[i] = (CELL)0;
    __syncthreads();
    if (n > 0) {
      // sorted: only rows with floor(ts) in {b-1, b} can touch plane b -- two regions, each walked upward, the upper
      // one first for even j; unsorted: scan the whole clip
      for (int part = 0; part < (assume_sorted ? 2 : 1); ++part) {
        const bool upper = (part == 0) == swapped;
        const int64_t lo = assume_sorted ? (upper ? cc[b] : cc[b > 0 ? b - 1 : 0]) : 0;
        const int64_t hi = assume_sorted ? (upper ? cc[b + 1] : cc[b]) : n;
        const double region = upper ? bd : bd - 1.0;      // floor(ts) of every row of this slab if the stamps are sorted
        // (Round 3, measured and removed: reading each 64-row block with two fully coalesced instructions -- lane l takes the 16 bytes at
        // 16 l, then at 1024 + 16 l -- and swapping the halves between lane pairs by DPP. Half the cache-line accesses per instruction, yet
        // slower: rows only loaded 84 against 76 us, whole kernel 124 against 101 us; tools/voxel_parts.py.
        // Also: software pipelining with a second batch of 4 rows per thread in flight (unconditional prefetch, so that hipcc's counted
        // waits read vmcnt(14) / (12) / (10) / (8)): 50 against 47 us for 64 workgroups, 105 against 98 us for 512 -- no gain. A workgroup
        // streams its 1.6 MB in ~45 us whether 64 or 256 CUs are busy (tools/voxel_clips_probe.py): ~40 GB/s per CU is what one CU's
        // vector-memory path sustains on rows that mostly miss its L1, and more requests queued behind it do not raise that.)
        for (int64_t k0 = threadIdx.x; k0 < hi - lo; k0 += (int64_t)VB_THREADS * VB_UNROLL) {
          double2 ra[VB_UNROLL], rb[VB_UNROLL];
          bool live[VB_UNROLL];
  #pragma unroll
          for (int u = 0; u < VB_UNROLL; ++u) {
            const int64_t k = k0 + (int64_t)u * VB_THREADS;
            live[u] = k < hi - lo;
            const double *row = ev + (lo + (live[u] ? k : 0)) * 4;
            ra[u] = *reinterpret_cast<const double2 *>(row);
            rb[u] = *reinterpret_cast<const double2 *>(row + 2);
          }
  #pragma unroll
          for (int u = 0; u < VB_UNROLL; ++u) {
            const double x = (TXYP ? ra[u].y : ra[u].x) * sx, y = (TXYP ? rb[u].x : ra[u].y) * sy;
            const double t = TXYP ? ra[u].x : rb[u].x, pd = rb[u].y;
            if (dbg == 3) { if (x == 1.2345e300 && pd == t) inconsistent = true; continue; }
            int64_t pix;
            if (__builtin_fabs(x) < 2147483648.0 && __builtin_fabs(y) < 2147483648.0)
              pix = (int64_t)(int)x + (int64_t)(int)y * (int64_t)W;  // same truncation as the int64 conversion below
            else
              pix = (int64_t)x + (int64_t)y * (int64_t)W;
            if (!live[u] || pix < pix0 || pix >= pix1) continue;
            if (dbg == 2) { atomicAdd(&tile[pix - pix0], (CELL)1); continue; }
            const double a = scale * (t - t0);
            const double ts = (clip_fast && in_safe_range(a)) ? div_by_clip_constant(a, dT, rT) : a / dT;
            const double tf = floor(ts);
            if (mode == 2 && !(tf == region)) inconsistent = true;   // NaN included
            if (!(tf >= 0.0)) continue;                   // also rejects NaN
            float p = (float)pd;
            if (p == 0.0f) p = -1.0f;
            const float dt = (float)(ts - tf);
            float val;
            if (tf == bd) val = p * (1.0f - dt);          // left neighbour, valid since b < bins
            else if (tf + 1.0 == bd) val = p * dt;         // right neighbour
            else continue;
            if (dbg == 1) { if (val == 1.2345e30f) inconsistent = true; continue; }
            atomicAdd(&tile[pix - pix0], (CELL)val);
          }
        }
      }
    }
    __syncthreads();
    float *dst = out + (((int64_t)clip * bins + b) * H + y0) * W;
    if ((W & 3) == 0) {
      for (int i = threadIdx.x; i < tile_elems / 4; i += VB_THREADS)
        reinterpret_cast<float4 *>(dst)[i] = make_float4((float)tile[4 * i], (float)tile[4 * i + 1], (float)tile[4 * i + 2], (float)tile[4 * i + 3]);
    } else {
      for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) dst[i] = (float)tile[i];
    }
  }
  if (inconsistent) flags[clip] = 0;       // every writer stores 0: no atomic needed
}

// ---- K1 fused with the event-level augmentation of the loader chain --------------------------------------------------------------
// The chain's merge kernel writes the augmented clip (window minus the erased rows plus the added rows, time-sorted) only for K1 to
// read it back four times. The voxel grid does not need that array: it is a SUM over the kept rows, and the only things the order
// decides are t0 / t1 (first / last stamp of the merged clip). So: the workgroups stream the ORIGINAL window rows through the same
// slab schedule, skip the erased ones by a bitmap in LDS (one bit per window row, behind the tile), and then run over the (at most
// 1 %) added rows, which arrive built and time-sorted from build_added_kernel. t0 = min(first kept row, first added row), t1 likewise.
// Verified like the plain form: a kept row outside the region its position implies flags the clip, the repair pass scans it whole.
struct FusedInfo { double t0, dT; };

__global__ __launch_bounds__(512) void voxel_cuts_fused_kernel(const double *events, const int64_t *win_begin, const int64_t *win_end,
                                                               const int64_t *erase_idx, const int64_t *erase_off, const double *added,
                                                               const int64_t *add_off, int bins, int64_t *cuts, int32_t *flags, FusedInfo *info) {
  __shared__ double sh_t0, sh_dT;
  __shared__ int64_t sh_first, sh_last;
  const int c = blockIdx.x;
  const int64_t beg = win_begin[c], n = win_end[c] - beg;
  const double *ev = events + beg * 4;
  int64_t *out = cuts + (int64_t)c * (bins + 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t e0 = erase_off[c], ke = erase_off[c + 1] - e0, a0 = add_off[c], ka = add_off[c + 1] - a0;
  if (threadIdx.x == 0) {
    const int64_t *E = erase_idx + e0;
    int64_t first = 0, last = n - 1;
    while (first < ke && E[first] == first) ++first;                        // leading rows that are all erased
    for (int64_t q = 0; q < ke && E[ke - 1 - q] == last; ++q) --last;       // trailing ones
    double t0 = 0.0, t1 = 0.0;
    bool have = false;
    if (n > 0 && first <= last) { t0 = ev[first * 4 + 2]; t1 = ev[last * 4 + 2]; have = true; }
    if (ka > 0) {
      const double ta = added[a0 * 4 + 2], tb = added[(a0 + ka - 1) * 4 + 2];
      t0 = have ? fmin(t0, ta) : ta;
      t1 = have ? fmax(t1, tb) : tb;
      have = true;
    }
    double dT = t1 - t0;
    if (dT == 0) dT = 1.0;
    sh_t0 = t0; sh_dT = dT;
    sh_first = first; sh_last = last;
    info[c].t0 = t0; info[c].dT = dT;
  }
  __syncthreads();
  const double t0 = sh_t0, dT = sh_dT;
  if (n <= 0) {
    for (int k = threadIdx.x; k <= bins + 1; k += blockDim.x) out[k] = 0;
    if (threadIdx.x == 0) flags[c] = 1;
    return;
  }
  for (int k = wave; k <= bins; k += nw) {
    int64_t lo = 0, hi = n;
    while (hi > lo) {
      const int64_t step = (hi - lo + 63) / 64;
      const int64_t idx = lo + (int64_t)lane * step;
      bool pred = false;
      if (idx < hi) pred = ts_of(ev[idx * 4 + 2], t0, dT, bins) >= (double)k;
      const unsigned long long m = __ballot(pred);
      if (m == 0ull) {
        const int64_t last = lo + ((hi - 1 - lo) / step) * step;
        lo = last + 1;
      } else {
        const int f = __ffsll((long long)m) - 1;
        const int64_t pf = lo + (int64_t)f * step;
        if (f > 0) lo = pf - step + 1;
        hi = pf;
      }
    }
    if (lane == 0) out[k] = lo;
  }
  if (threadIdx.x == 0) out[bins + 1] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    // Sorted stamps put nothing but ERASED rows in front of cuts[0] (ts < 0: earlier than the first kept row) and behind cuts[bins]
    // (later than the last kept one) -- the slabs never visit those two ranges, so a kept row there must send the clip to the repair
    // pass; between them the bin kernel checks every kept row it normalises.
    bool ok = out[0] <= sh_first && out[bins] >= sh_last + 1;
    for (int k = 0; k < bins; ++k) ok = ok && out[k] <= out[k + 1];
    flags[c] = ok ? 1 : 0;
  }
}

// one visit of a row (original or added) by the workgroup of plane b / tile [pix0, pix1); returns false when a verified row is outside
// `region` (region < -1: no check)
__device__ __forceinline__ bool fused_visit(double x, double y, double t, double pd, int W, int64_t pix0, int64_t pix1, double t0, double dT,
                                            double rT, bool clip_fast, double scale, double bd, double region, float *tile) {
  int64_t pix;
  if (__builtin_fabs(x) < 2147483648.0 && __builtin_fabs(y) < 2147483648.0) pix = (int64_t)(int)x + (int64_t)(int)y * (int64_t)W;
  else pix = (int64_t)x + (int64_t)y * (int64_t)W;
  if (pix < pix0 || pix >= pix1) return true;
  const double a = scale * (t - t0);
  const double ts = (clip_fast && in_safe_range(a)) ? div_by_clip_constant(a, dT, rT) : a / dT;
  const double tf = floor(ts);
  const bool good = region < -1.5 || tf == region;
  if (!(tf >= 0.0)) return good;
  float p = (float)pd;
  if (p == 0.0f) p = -1.0f;
  const float dt = (float)(ts - tf);
  if (tf == bd) atomicAdd(&tile[pix - pix0], p * (1.0f - dt));
  else if (tf + 1.0 == bd) atomicAdd(&tile[pix - pix0], p * dt);
  return good;
}

__global__ __launch_bounds__(VB_THREADS) void voxel_bin_fused_kernel(const double *events, const int64_t *win_begin, const int64_t *win_end,
                                                                     const int64_t *erase_idx, const int64_t *erase_off, const double *added,
                                                                     const int64_t *add_off, const int64_t *cuts, const FusedInfo *info, int n_clips,
                                                                     int bins, int H, int W, int mode, int tile_rows, int n_yt, int bm_words,
                                                                     double sx, double sy, float *out, int32_t *flags, const int32_t *view_params,
                                                                     int Hout, int Wout, int negate) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *tile = reinterpret_cast<float *>(smem_raw);
  uint32_t *bm = reinterpret_cast<uint32_t *>(smem_raw + (size_t)tile_rows * W * sizeof(float));     // erased-row bitmap of the window
  const int n_j = bins > 1 ? bins - 1 : 1;
  const int per_clip = n_j * n_yt;
  int clip, sub;
  if ((n_clips & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    clip = (slot / per_clip) * 8 + xcd;
    sub = slot % per_clip;
  } else {
    clip = blockIdx.x / per_clip;
    sub = blockIdx.x % per_clip;
  }
  const int ok = flags[clip];
  if (mode == 2 ? !ok : ok) return;        // 2: fast pass for the clips whose cuts are a partition; 3: repair pass for the others
  const bool sorted = mode == 2;
  bool inconsistent = false;
  const int j = sub / n_yt, yt = sub % n_yt;
  const int y0 = yt * tile_rows, y1 = (y0 + tile_rows < H) ? y0 + tile_rows : H;
  const int tile_elems = (y1 - y0) * W;
  const int64_t pix0 = (int64_t)y0 * W, pix1 = (int64_t)y1 * W;
  const int64_t beg = win_begin[clip], n = win_end[clip] - beg;
  const double *ev = events + beg * 4;
  const int64_t *cc = cuts + (int64_t)clip * (bins + 2);
  const double t0 = info[clip].t0, dT = info[clip].dT;
  const double rT = 1.0 / dT, scale = (double)(bins - 1);
  const bool clip_fast = in_safe_range(dT);
  const int64_t e0 = erase_off[clip], ke = erase_off[clip + 1] - e0, a0 = add_off[clip], ka = add_off[clip + 1] - a0;
  for (int i = threadIdx.x; i < bm_words; i += VB_THREADS) bm[i] = 0u;
  __syncthreads();
  for (int64_t q = threadIdx.x; q < ke; q += VB_THREADS) {
    const int64_t r = erase_idx[e0 + q];
    if (r >= 0 && r < n && (r >> 5) < bm_words) atomicOr(&bm[r >> 5], 1u << (r & 31));
  }
  const int n_jobs = (j == 0 && bins > 1) ? 2 : 1;
  for (int job = 0; job < n_jobs; ++job) {
    const int b = job ? bins - 1 : j;
    const bool swapped = (j > 0) && ((j & 1) == 0);
    const double bd = (double)b;
    __syncthreads();                       // the bitmap is complete / the previous plane's flush has read the tile
    for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) tile[i] = 0.f;
    __syncthreads();
    if (n > 0) {
      for (int part = 0; part < (sorted ? 2 : 1); ++part) {
        const bool upper = (part == 0) == swapped;
        const int64_t lo = sorted ? (upper ? cc[b] : cc[b > 0 ? b - 1 : 0]) : 0;
        // the last plane also takes what lies behind cuts[bins]: kept rows with ts == bins - 1 sit in front of it, rows behind it are erased
        const int64_t hi = sorted ? (upper ? cc[b + 1] : cc[b]) : n;
        const double region = sorted ? (upper ? bd : bd - 1.0) : -2.0;
        for (int64_t k0 = threadIdx.x; k0 < hi - lo; k0 += (int64_t)VB_THREADS * VB_UNROLL) {
          double2 ra[VB_UNROLL], rb[VB_UNROLL];
          bool live[VB_UNROLL];
#pragma unroll
          for (int u = 0; u < VB_UNROLL; ++u) {
            const int64_t k = k0 + (int64_t)u * VB_THREADS;
            live[u] = k < hi - lo;
            const int64_t r = lo + (live[u] ? k : 0);
            const double *row = ev + r * 4;
            ra[u] = *reinterpret_cast<const double2 *>(row);
            rb[u] = *reinterpret_cast<const double2 *>(row + 2);
            if (live[u] && ((bm[r >> 5] >> (r & 31)) & 1u)) live[u] = false;       // erased
          }
#pragma unroll
          for (int u = 0; u < VB_UNROLL; ++u) {
            if (!live[u]) continue;
            if (!fused_visit(ra[u].x * sx, ra[u].y * sy, rb[u].x, rb[u].y, W, pix0, pix1, t0, dT, rT, clip_fast, scale, bd, region, tile)) inconsistent = true;
          }
        }
      }
    }
    // the added rows (built, clipped to the sensor and time-sorted by build_added_kernel): every workgroup of the clip looks at all of them
    for (int64_t q = threadIdx.x; q < ka; q += VB_THREADS) {
      const double *row = added + (a0 + q) * 4;
      fused_visit(row[0] * sx, row[1] * sy, row[2], row[3], W, pix0, pix1, t0, dT, rT, clip_fast, scale, bd, -2.0, tile);
    }
    __syncthreads();
    if (view_params) {
      // flush THROUGH the view augmentation (view_augment.py:9-77 as csrc/augment.hip restates it: crop box, float32 nearest resize, the
      // horizontal flip on the resized view, time flip = reversed plane order and, for polarity grids, the sign): an output pixel reads one
      // grid pixel, so this workgroup writes exactly the output rows whose source row lies in its tile -- the raw grid is never stored
      const int32_t *pr = view_params + (int64_t)clip * 6;
      const int px0 = pr[0], py0 = pr[1], pw = pr[2], ph = pr[3], hflip = pr[4], tflip = pr[5];
      const float vsy = (float)ph / (float)Hout, vsx = (float)pw / (float)Wout;
      const float sgn = (tflip && negate) ? -1.0f : 1.0f;
      float *plane = out + ((int64_t)clip * bins + (tflip ? bins - 1 - b : b)) * Hout * Wout;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int y = wave; y < Hout; y += VB_THREADS / 64) {          // a wave per output row: the row test is wave-uniform
        int ys = (int)floorf((float)y * vsy);
        ys = py0 + (ys < ph - 1 ? ys : ph - 1);
        if (ys < y0 || ys >= y1) continue;
        const float *src = tile + (ys - y0) * W;
        float *dst = plane + (int64_t)y * Wout;
        for (int x = lane; x < Wout; x += 64) {
          const int xr = hflip ? Wout - 1 - x : x;
          int xs = (int)floorf((float)xr * vsx);
          xs = px0 + (xs < pw - 1 ? xs : pw - 1);
          dst[x] = sgn * src[xs];
        }
      }
    } else {
      float *dst = out + (((int64_t)clip * bins + b) * H + y0) * W;
      if ((W & 3) == 0) {
        for (int i = threadIdx.x; i < tile_elems / 4; i += VB_THREADS)
          reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(tile)[i];
      } else {
        for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) dst[i] = tile[i];
      }
    }
  }
  if (inconsistent) flags[clip] = 0;
}

// ---- two-pass form (algo 2): decode once, bin from packed records ------------------------------------------------
// Pass A decodes every event exactly once (the float64 time normalisation with its division is the expensive part)
// into three 4-byte streams: key = pix | floor(ts) << 24 | invalid << 31, and the two float32 contributions
// val_left = p*(1-dt), val_right = p*dt. Pass B (one workgroup per clip x bin x y-tile, tile in LDS) then only reads
// the 4-byte key of every event of its slab and touches a value only when the event lands in its tile.
constexpr uint32_t KEY_INVALID = 0x80000000u;
__global__ __launch_bounds__(256) void voxel_pack_kernel(const double *events, const int64_t *offsets, int bins, int H, int W, int is_txyp, double sx, double sy,
                                                         uint32_t *keys, float *vleft, float *vright) {
  const int clip = blockIdx.y;
  const int64_t beg = offsets[clip], n = offsets[clip + 1] - beg;
  if (n <= 0) return;
  const double *ev = events + beg * 4;
  const double t0 = stamp(ev, 0, is_txyp), t1 = stamp(ev, n - 1, is_txyp);
  double dT = t1 - t0;
  if (dT == 0) dT = 1.0;
  const int64_t plane = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const Event e = load_event(ev, i, is_txyp, sx, sy);
    const double ts = ts_of(e.t, t0, dT, bins);
    const double tf = floor(ts);
    float p = (float)e.p;
    if (p == 0.0f) p = -1.0f;
    const float dt = (float)(ts - tf);
    const int64_t pix = (int64_t)e.x + (int64_t)e.y * (int64_t)W;
    uint32_t key = KEY_INVALID;
    if (tf >= 0.0 && tf < (double)bins && pix >= 0 && pix < plane) key = (uint32_t)pix | ((uint32_t)tf << 24);
    keys[beg + i] = key;
    vleft[beg + i] = p * (1.0f - dt);
    vright[beg + i] = p * dt;
  }
}

__global__ __launch_bounds__(VB_THREADS) void voxel_bin_packed_kernel(const uint32_t *keys, const float *vleft, const float *vright,
                                                                      const int64_t *offsets, const int64_t *cuts, int n_clips, int bins,
                                                                      int H, int W, int assume_sorted, int tile_rows, int n_yt, float *out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *tile = reinterpret_cast<float *>(smem_raw);
  const int per_clip = bins * n_yt;
  int clip, sub;
  if ((n_clips & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    clip = (slot / per_clip) * 8 + xcd;
    sub = slot % per_clip;
  } else {
    clip = blockIdx.x / per_clip;
    sub = blockIdx.x % per_clip;
  }
  const int b = sub / n_yt, yt = sub % n_yt;
  const int y0 = yt * tile_rows, y1 = (y0 + tile_rows < H) ? y0 + tile_rows : H;
  const int tile_elems = (y1 - y0) * W;
  for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) tile[i] = 0.f;
  __syncthreads();
  const int64_t beg = offsets[clip], n = offsets[clip + 1] - beg;
  float *dst = out + (((int64_t)clip * bins + b) * H + y0) * W;
  if (n > 0) {
    const int64_t *cc = cuts + (int64_t)clip * (bins + 2);
    const int64_t lo = assume_sorted ? cc[b > 0 ? b - 1 : 0] : 0;
    const int64_t hi = assume_sorted ? cc[b + 1] : n;
    const uint32_t pix0 = (uint32_t)(y0 * W), pix1 = (uint32_t)(y1 * W);
    const uint32_t *kk = keys + beg;
    const float *vl = vleft + beg, *vr = vright + beg;
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)VB_THREADS * 8) {
      uint32_t k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = i0 + (int64_t)u * VB_THREADS;
        k[u] = i < hi ? kk[i] : KEY_INVALID;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k[u] & KEY_INVALID) continue;
        const uint32_t pix = k[u] & 0xFFFFFFu;
        const int tf = (int)(k[u] >> 24);
        if (pix < pix0 || pix >= pix1) continue;
        const int64_t i = i0 + (int64_t)u * VB_THREADS;
        if (tf == b) atomicAdd(&tile[pix - pix0], vl[i]);             // left neighbour (tf < bins by construction)
        else if (tf + 1 == b) atomicAdd(&tile[pix - pix0], vr[i]);    // right neighbour
      }
    }
  }
  __syncthreads();
  if ((W & 3) == 0) {
    for (int i = threadIdx.x; i < tile_elems / 4; i += VB_THREADS)
      reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(tile)[i];
  } else {
    for (int i = threadIdx.x; i < tile_elems; i += VB_THREADS) dst[i] = tile[i];
  }
}

// algo 1: two global float atomics per event into a pre-zeroed grid
__global__ __launch_bounds__(256) void voxel_atomic_kernel(const double *events, const int64_t *offsets, int bins, int H, int W,
                                                           int is_txyp, double sx, double sy, float *out) {
  const int clip = blockIdx.y;
  const int64_t beg = offsets[clip], n = offsets[clip + 1] - beg;
  if (n <= 0) return;
  const double *ev = events + beg * 4;
  const double t0 = stamp(ev, 0, is_txyp), t1 = stamp(ev, n - 1, is_txyp);
  double dT = t1 - t0;
  if (dT == 0) dT = 1.0;
  const int64_t plane = (int64_t)H * W;
  float *grid = out + (int64_t)clip * bins * plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const Event e = load_event(ev, i, is_txyp, sx, sy);
    const double ts = ts_of(e.t, t0, dT, bins);
    const double tf = floor(ts);
    if (!(tf >= 0.0)) continue;
    float p = (float)e.p;
    if (p == 0.0f) p = -1.0f;
    const float dt = (float)(ts - tf);
    const int64_t pix = (int64_t)e.x + (int64_t)e.y * (int64_t)W;
    if (pix < 0 || pix >= plane) continue;
    if (tf < (double)bins) atomicAdd(&grid[(int64_t)tf * plane + pix], p * (1.0f - dt));
    if (tf + 1.0 < (double)bins) atomicAdd(&grid[((int64_t)tf + 1) * plane + pix], p * dt);
  }
}

__global__ __launch_bounds__(256) void sorted_check_kernel(const double *events, const int64_t *offsets, int is_txyp, int32_t *flags) {
  __shared__ int bad;
  const int clip = blockIdx.x;
  const int64_t beg = offsets[clip], n = offsets[clip + 1] - beg;
  const int ct = is_txyp ? 0 : 2;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const double *ev = events + beg * 4;
  int mybad = 0;
  for (int64_t i = threadIdx.x + 1; i < n; i += blockDim.x)
    if (!(ev[i * 4 + ct] >= ev[(i - 1) * 4 + ct])) mybad = 1;
  if (mybad) atomicOr(&bad, 1);
  __syncthreads();
  if (threadIdx.x == 0) flags[clip] = bad ? 0 : 1;
}

}  // namespace

static int g_voxel_dbg = 0;
extern "C" int evp_voxel_set_debug(int v) { const int old = g_voxel_dbg; g_voxel_dbg = v; return old; }

extern "C" int evp_voxel_scatter_scaled_f32(const double *events, const int64_t *clip_offsets, int n_clips, int64_t n_events_total,
                                            int bins, int H, int W, int is_txyp, int assume_sorted, int algo, int tile_rows,
                                            double scale_x, double scale_y, int64_t *workspace, float *out, void *stream) {
  EVP_CHECK_ARG(events && clip_offsets && out, EVP_EINVAL, "evp_voxel_scatter_f32: null pointer");
  EVP_CHECK_ARG(n_clips > 0 && bins > 0 && bins <= 64 && H > 0 && W > 0, EVP_ESHAPE,
                "evp_voxel_scatter_f32: need n_clips>0, 0<bins<=64, H,W>0 (got %d,%d,%d,%d)", n_clips, bins, H, W);
  EVP_CHECK_ARG(((uintptr_t)events & 15) == 0, EVP_EINVAL, "evp_voxel_scatter_f32: events must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (algo == 1) {
    hipError_t e = evp_zero_async(out, sizeof(float) * (size_t)n_clips * bins * H * W, s);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_voxel_scatter_f32: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(voxel_atomic_kernel, dim3(64, n_clips), dim3(256), 0, s, events, clip_offsets, bins, H, W, is_txyp, scale_x,
                       scale_y, out);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_f32(atomic)");
    return EVP_OK;
  }
  EVP_CHECK_ARG(algo == 0 || algo == 2 || algo == 3, EVP_EINVAL, "evp_voxel_scatter_f32: unknown algo %d", algo);
  const bool f64_cells = algo == 3;                 // algo 3: float64 cells in the LDS tile (A/B; see voxel_bin_kernel)
  const size_t cell = f64_cells ? sizeof(double) : sizeof(float);
  EVP_CHECK_ARG(workspace, EVP_EINVAL, "evp_voxel_scatter_f32: workspace required for the LDS-binned algorithms");
  EVP_CHECK_ARG(algo != 2 || (n_events_total > 0 && (int64_t)H * W <= (1 << 24)), EVP_ESHAPE,
                "evp_voxel_scatter_f32: algo 2 needs n_events_total and H*W <= 2^24");
  if (tile_rows <= 0) {
    // f32 cells: <= 100 KiB of LDS (big tiles halve the slab re-reads; measured best); f64 cells: <= 150 KiB, three y-tiles at 224 x 224
    const int max_rows = (int)(((f64_cells ? 150 : 100) * 1024) / ((size_t)W * cell));
    tile_rows = max_rows < 1 ? 1 : (max_rows > H ? H : max_rows);
    const int nyt = (H + tile_rows - 1) / tile_rows;
    tile_rows = (H + nyt - 1) / nyt;  // balance the tiles
  }
  EVP_CHECK_ARG((size_t)tile_rows * W * cell <= 160 * 1024, EVP_ESHAPE, "evp_voxel_scatter_f32: tile of %d rows x %d exceeds LDS", tile_rows, W);
  const int n_yt = (H + tile_rows - 1) / tile_rows;
  const size_t smem = (size_t)tile_rows * W * (algo == 2 ? sizeof(float) : cell);
  EVP_CHECK_ARG(assume_sorted >= 0 && assume_sorted <= 2, EVP_EINVAL, "evp_voxel_scatter_f32: assume_sorted must be 0, 1 or 2");
  EVP_CHECK_ARG(assume_sorted != 2 || algo == 0 || algo == 3, EVP_EUNSUPPORTED, "evp_voxel_scatter_f32: the verified mode (assume_sorted = 2) is built for algo 0 / 3");
  int32_t *flags = assume_sorted == 2 ? reinterpret_cast<int32_t *>(workspace + (int64_t)n_clips * (bins + 2)) : nullptr;
  if (assume_sorted) {
    hipLaunchKernelGGL(voxel_cuts_kernel, dim3(n_clips), dim3(512), 0, s, events, clip_offsets, bins, is_txyp, workspace, flags);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_f32(cuts)");
  }
  if (algo == 2) {
    uint32_t *keys = reinterpret_cast<uint32_t *>(workspace + (int64_t)n_clips * (bins + 2));
    float *vl = reinterpret_cast<float *>(keys + n_events_total), *vr = vl + n_events_total;
    hipLaunchKernelGGL(voxel_pack_kernel, dim3(64, n_clips), dim3(256), 0, s, events, clip_offsets, bins, H, W, is_txyp, scale_x,
                       scale_y, keys, vl, vr);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_f32(pack)");
    if (smem > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(voxel_bin_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_voxel_scatter_f32: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(voxel_bin_packed_kernel, dim3(n_clips * bins * n_yt), dim3(VB_THREADS), smem, s, keys, vl, vr, clip_offsets, workspace,
                       n_clips, bins, H, W, assume_sorted, tile_rows, n_yt, out);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_f32(bin packed)");
    return EVP_OK;
  }
  auto kern = f64_cells ? (is_txyp ? voxel_bin_kernel<true, double, false> : voxel_bin_kernel<false, double, false>)
                        : (is_txyp ? voxel_bin_kernel<true, float, false> : voxel_bin_kernel<false, float, false>);
  if (g_voxel_dbg)
    kern = f64_cells ? (is_txyp ? voxel_bin_kernel<true, double, true> : voxel_bin_kernel<false, double, true>)
                     : (is_txyp ? voxel_bin_kernel<true, float, true> : voxel_bin_kernel<false, float, true>);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_voxel_scatter_f32: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  const int n_blocks = n_clips * (bins > 1 ? bins - 1 : 1) * n_yt;  // plane 0 and plane bins-1 share a workgroup
  for (int pass = 0; pass < (assume_sorted == 2 ? 2 : 1); ++pass) {
    const int mode = pass ? 3 : assume_sorted;      // verified mode: fast pass, then the repair pass for flagged clips
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(VB_THREADS), smem, s, events, clip_offsets, workspace, n_clips, bins, H, W, mode, tile_rows,
                       n_yt, scale_x, scale_y, out, flags, g_voxel_dbg);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_f32(bin)");
  }
  return EVP_OK;
}

extern "C" int evp_voxel_scatter_f32(const double *events, const int64_t *clip_offsets, int n_clips, int64_t n_events_total, int bins,
                                     int H, int W, int is_txyp, int assume_sorted, int algo, int tile_rows, int64_t *workspace,
                                     float *out, void *stream) {
  return evp_voxel_scatter_scaled_f32(events, clip_offsets, n_clips, n_events_total, bins, H, W, is_txyp, assume_sorted, algo, tile_rows,
                                      1.0, 1.0, workspace, out, stream);
}

extern "C" int evp_events_sorted_check(const double *events, const int64_t *clip_offsets, int n_clips, int is_txyp,
                                       int32_t *sorted_flags, void *stream) {
  EVP_CHECK_ARG(events && clip_offsets && sorted_flags && n_clips > 0, EVP_EINVAL, "evp_events_sorted_check: bad argument");
  hipLaunchKernelGGL(sorted_check_kernel, dim3(n_clips), dim3(256), 0, (hipStream_t)stream, events, clip_offsets, is_txyp, sorted_flags);
  EVP_CHECK_LAUNCH("evp_events_sorted_check");
  return EVP_OK;
}

extern "C" int evp_voxel_scatter_fused_f32(const double *events, const int64_t *win_begin, const int64_t *win_end, int n_clips,
                                           const int64_t *erase_idx, const int64_t *erase_offsets, const double *added_rows,
                                           const int64_t *add_offsets, int64_t max_window, int bins, int H, int W, double scale_x, double scale_y,
                                           const int32_t *view_params, int view_h, int view_w, int negate_on_time_flip, int64_t *workspace,
                                           float *out, void *stream) {
  EVP_CHECK_ARG(events && win_begin && win_end && erase_idx && erase_offsets && added_rows && add_offsets && workspace && out, EVP_EINVAL,
                "evp_voxel_scatter_fused_f32: null pointer");
  EVP_CHECK_ARG(n_clips > 0 && bins > 0 && bins <= 64 && H > 0 && W > 0 && max_window > 0, EVP_ESHAPE, "evp_voxel_scatter_fused_f32: bad shape");
  EVP_CHECK_ARG(((uintptr_t)events & 15) == 0, EVP_EINVAL, "evp_voxel_scatter_fused_f32: events must be 16-byte aligned");
  EVP_CHECK_ARG(!view_params || (view_h > 0 && view_w > 0), EVP_ESHAPE, "evp_voxel_scatter_fused_f32: view size required with view_params");
  hipStream_t s = (hipStream_t)stream;
  const int bm_words = (int)((max_window + 31) / 32);
  const size_t bm_bytes = (size_t)bm_words * 4;
  EVP_CHECK_ARG(bm_bytes <= 48 * 1024, EVP_ESHAPE, "evp_voxel_scatter_fused_f32: windows of at most %d rows (got %lld)", 48 * 1024 * 8, (long long)max_window);
  const int max_rows = (int)((100 * 1024) / ((size_t)W * sizeof(float)));
  int tile_rows = max_rows < 1 ? 1 : (max_rows > H ? H : max_rows);
  const int n_yt = (H + tile_rows - 1) / tile_rows;
  tile_rows = (H + n_yt - 1) / n_yt;
  const size_t smem = (size_t)tile_rows * W * sizeof(float) + bm_bytes;
  EVP_CHECK_ARG(smem <= 160 * 1024, EVP_ESHAPE, "evp_voxel_scatter_fused_f32: tile of %d rows x %d + bitmap exceeds LDS", tile_rows, W);
  int64_t *cuts = workspace;
  FusedInfo *info = reinterpret_cast<FusedInfo *>(workspace + (int64_t)n_clips * (bins + 2));
  int32_t *flags = reinterpret_cast<int32_t *>(workspace + (int64_t)n_clips * (bins + 2) + 2 * (int64_t)n_clips);
  hipLaunchKernelGGL(voxel_cuts_fused_kernel, dim3(n_clips), dim3(512), 0, s, events, win_begin, win_end, erase_idx, erase_offsets, added_rows, add_offsets, bins,
                     cuts, flags, info);
  EVP_CHECK_LAUNCH("evp_voxel_scatter_fused_f32(cuts)");
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(voxel_bin_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_voxel_scatter_fused_f32: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  const int n_blocks = n_clips * (bins > 1 ? bins - 1 : 1) * n_yt;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(voxel_bin_fused_kernel, dim3(n_blocks), dim3(VB_THREADS), smem, s, events, win_begin, win_end, erase_idx, erase_offsets, added_rows,
                       add_offsets, cuts, info, n_clips, bins, H, W, pass ? 3 : 2, tile_rows, n_yt, bm_words, scale_x, scale_y, out, flags, view_params,
                       view_h, view_w, negate_on_time_flip);
    EVP_CHECK_LAUNCH("evp_voxel_scatter_fused_f32(bin)");
  }
  return EVP_OK;
}
