// Row-wise (LayerNorm family) and column-wise (bias-gradient, BatchNorm) reductions for gfx950.
// All HBM-bound: one 64-lane wave owns one row, holds it in registers as float4 chunks (lane-strided, so every
// wave-instruction moves 1 KiB contiguous), reduces with cross-lane shuffles, writes once.
#include <cstdlib>
#include "evp_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves / 256 threads (8 waves per block measured 10-25 % slower)
constexpr int LN_THREADS = ROWS_PER_BLOCK * 64;
constexpr int LN_MAX_BLOCKS = 512;   // backward: also the number of dgamma / dbeta / column-sum partial rows. Round 3, same-box A/B of the
                                     // ViT-Base step: 1024 blocks 10.94-11.00 ms, 768 10.86-10.88, 512 10.84-10.88, 392 10.85-10.88 -- the
                                     // partial rows (nblk x 3 D floats per launch) are written here and read again by the grouped column sum
constexpr int LN_FWD_BLOCKS = 2048;  // forward has no partials: one row per wave up to 8192 rows

// Number of float4 chunks a lane holds for a row of D floats
static inline int vpl_for(int D) {
  const int chunks = D / 4;
  const int v = (chunks + 63) / 64;
  int r = 1;
  while (r < v) r <<= 1;
  return r;
}

template <int VPL> struct Row {
  float4 v[VPL];
  __device__ __forceinline__ void load(const float *p, int D, int lane) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c * 4 < D) ? *reinterpret_cast<const float4 *>(p + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void add(const float *p, int D, int lane) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        const float4 t = *reinterpret_cast<const float4 *>(p + c * 4);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
    }
  }
  __device__ __forceinline__ void load_any(const void *p, int dtype, int64_t off, int D, int lane) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        if (dtype == EVP_F32) v[i] = *reinterpret_cast<const float4 *>((const float *)p + off + c * 4);
        else {
          const uint2 u = *reinterpret_cast<const uint2 *>((const bf16_t *)p + off + c * 4);
          v[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                             __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
        }
      } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ float sum(int D, int lane) const {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if ((lane + 64 * i) * 4 < D) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    return wave_sum(s);
  }
  __device__ __forceinline__ float sumsq_centered(float mu, int D, int lane) const {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if ((lane + 64 * i) * 4 < D) {
        const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
        s += (a * a + b * b) + (c * c + d * d);
      }
    return wave_sum(s);
  }
};

__device__ __forceinline__ void store4_any(void *p, int dtype, int64_t off, float4 v) {
  if (dtype == EVP_F32) *reinterpret_cast<float4 *>((float *)p + off) = v;
  else {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2 *>((bf16_t *)p + off) = u;
  }
}

// ------------------------------------------------------------------------------------------------ LN forward
template <int VPL>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float *x, const float *x2, const float *x3, const float *gamma,
                                                     const float *beta, int64_t M, int D, float eps, void *y, int y_dtype,
                                                     float *mean, float *rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; r < M; r += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    Row<VPL> row;
    row.load(x + r * D, D, lane);
    if (x2) row.add(x2 + r * D, D, lane);
    if (x3) row.add(x3 + r * D, D, lane);
    const float mu = row.sum(D, lane) / (float)D;
    const float var = row.sumsq_centered(mu, D, lane) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
      if (mean) mean[r] = mu;
      if (rstd) rstd[r] = rs;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c * 4);
        const float4 b = *reinterpret_cast<const float4 *>(beta + c * 4);
        float4 o;
        o.x = (row.v[i].x - mu) * rs * g.x + b.x;
        o.y = (row.v[i].y - mu) * rs * g.y + b.y;
        o.z = (row.v[i].z - mu) * rs * g.z + b.z;
        o.w = (row.v[i].w - mu) * rs * g.w + b.w;
        store4_any(y, y_dtype, r * D + c * 4, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ LN backward
// partial layout: part[blk][0][D] = dgamma partial, part[blk][1][D] = dbeta partial and, with CS, part[blk][2][D] = column sums
// of the OUTPUT dx (= the residual-stream gradient: its column sum is the bias gradient of the Linear that fed the stream,
// so the grouped column-sum launch reduces these few rows instead of re-reading the whole f32 gradient)
template <int VPL, bool CS>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const void *dy, int dy_dtype, const float *x, const float *x2,
                                                     const float *x3, const float *gamma, const float *mean,
                                                     const float *rstd, const float *gres, int64_t M, int D, float *dx,
                                                     void *dx_lp, float *part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *sh = reinterpret_cast<float *>(smem_raw);  // [ROWS_PER_BLOCK][NP][D]
  constexpr int NP = CS ? 3 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dg[VPL], db[VPL], dc[CS ? VPL : 1];
#pragma unroll
  for (int i = 0; i < VPL; ++i) dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < (CS ? VPL : 1); ++i) dc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // TWO rows per wave and iteration, every load of both rows (x, dy, the residual-stream gradient, the row statistics) requested
  // before anything is reduced: the one-row form issued x and dy, ran two shuffle reductions, and only then asked for gres -- 6-9
  // loads in flight per wave and a dependent HBM round trip in the middle of every row (42 launches at 3.8-4.4 TB/s).
  // Same-box A/B against the one-row form (tools/ab_step.py, round 3): 11.366 against 11.387 ms per ViT-Base step -- inside the noise;
  // the kernel is not bound by its loads in flight.
  // (rows wider than 1024 floats per 64 lanes -- VPL > 4 -- keep one row per iteration: two would spill)
  constexpr int NR = VPL <= 4 ? 2 : 1;
  const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
  for (int64_t r0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; r0 < M; r0 += NR * stride) {
    const int64_t rr[2] = {r0, r0 + stride};
    const bool live[2] = {true, r0 + stride < M};
    Row<VPL> xr[NR], gr[NR], qr[NR];
    float mu[NR], rs[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int64_t r = live[k] ? rr[k] : r0;          // a dead second row re-reads the first (never stored, never summed)
      xr[k].load(x + r * D, D, lane);
      gr[k].load_any(dy, dy_dtype, r * D, D, lane);
      if (gres) qr[k].load(gres + r * D, D, lane);
      mu[k] = mean[r];
      rs[k] = rstd[r];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int64_t r = live[k] ? rr[k] : r0;
      if (x2) xr[k].add(x2 + r * D, D, lane);
      if (x3) xr[k].add(x3 + r * D, D, lane);
    }
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};      // (entries >= NR stay 0)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c * 4);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          float4 xh;
          xh.x = (xr[k].v[i].x - mu[k]) * rs[k]; xh.y = (xr[k].v[i].y - mu[k]) * rs[k];
          xh.z = (xr[k].v[i].z - mu[k]) * rs[k]; xh.w = (xr[k].v[i].w - mu[k]) * rs[k];
          const float4 d = gr[k].v[i];
          if (live[k]) {
            dg[i].x += d.x * xh.x; dg[i].y += d.y * xh.y; dg[i].z += d.z * xh.z; dg[i].w += d.w * xh.w;
            db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
          }
          float4 t;  // dy * gamma
          t.x = d.x * g.x; t.y = d.y * g.y; t.z = d.z * g.z; t.w = d.w * g.w;
          s1[k] += (t.x + t.y) + (t.z + t.w);
          s2[k] += (t.x * xh.x + t.y * xh.y) + (t.z * xh.z + t.w * xh.w);
          gr[k].v[i] = t;
          xr[k].v[i] = xh;
        }
      }
    }
    // four reductions interleaved: the shuffles of one hide under the others' latency
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float a0 = __shfl_xor(s1[0], o, 64), a1 = __shfl_xor(s2[0], o, 64), a2 = __shfl_xor(s1[1], o, 64), a3 = __shfl_xor(s2[1], o, 64);
      s1[0] += a0; s2[0] += a1; s1[1] += a2; s2[1] += a3;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (!live[k]) continue;
      const int64_t r = rr[k];
      const float m1 = s1[k] / (float)D, m2 = s2[k] / (float)D;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c * 4 < D) {
          float4 o;
          o.x = rs[k] * (gr[k].v[i].x - m1 - xr[k].v[i].x * m2);
          o.y = rs[k] * (gr[k].v[i].y - m1 - xr[k].v[i].y * m2);
          o.z = rs[k] * (gr[k].v[i].z - m1 - xr[k].v[i].z * m2);
          o.w = rs[k] * (gr[k].v[i].w - m1 - xr[k].v[i].w * m2);
          if (gres) {
            const float4 q = qr[k].v[i];
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
          }
          if (dx) *reinterpret_cast<float4 *>(dx + r * D + c * 4) = o;
          if (dx_lp) store4_any(dx_lp, EVP_BF16, r * D + c * 4, o);
          if constexpr (CS) { dc[i].x += o.x; dc[i].y += o.y; dc[i].z += o.z; dc[i].w += o.w; }
        }
      }
    }
  }
  // combine the 4 waves' column partials through LDS, then one store per block
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c * 4 < D) {
      *reinterpret_cast<float4 *>(sh + (wave * NP + 0) * D + c * 4) = dg[i];
      *reinterpret_cast<float4 *>(sh + (wave * NP + 1) * D + c * 4) = db[i];
      if constexpr (CS) *reinterpret_cast<float4 *>(sh + (wave * NP + 2) * D + c * 4) = dc[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NP * D; e += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ROWS_PER_BLOCK; ++w) s += sh[w * NP * D + e];
    part[(int64_t)blockIdx.x * NP * D + e] = s;
  }
}

// out0[d] = sum_blk part[blk][0][d], out1[d] = sum_blk part[blk][1][d].
// 1024 threads = 64 columns x 16 partial-row groups: coalesced 256-byte reads, 16-way parallel over the partials.
__global__ __launch_bounds__(1024) void ln_bwd_finalize(const float *part, int nblk, int D, float *out0, float *out1) {
  __shared__ float sh[16][65];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + c;
  float s = 0.f;
  if (e < 2 * D)
    for (int b = g; b < nblk; b += 16) s += part[(int64_t)b * 2 * D + e];
  sh[g][c] = s;
  __syncthreads();
  if (g == 0 && e < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][c];
    if (e < D) { if (out0) out0[e] = t; }
    else { if (out1) out1[e - D] = t; }
  }
}

// backward grid cap; EVP_LN_BWD_BLOCKS overrides it for A/B runs (read once)
static inline int ln_bwd_cap() {
  static const int v = [] {
    const char *e = getenv("EVP_LN_BWD_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n >= 64 && n <= 8192 ? n : LN_MAX_BLOCKS;
  }();
  return v;
}
static inline int ln_grid(int64_t M, int cap = 0) {
  if (cap <= 0) cap = ln_bwd_cap();
  int64_t g = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------ column sums
// out[n] += sum_m x[m,n]. Block = 16 column chunks (8 columns each, one 16/32-byte load per thread) x 16 row lanes
// over a slab of CS_ROWS rows; the 16 row lanes meet in LDS, then ONE f32 atomic per column and block.
constexpr int CS_ROWS = 256;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *x, int64_t M, int N, int64_t ld, float *out) {
  __shared__ float sh[16][129];
  const int cc = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.x * 128 + cc * 8;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS, r1 = r0 + CS_ROWS < M ? r0 + CS_ROWS : M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    const bool full = n + 7 < N;
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const T *p = x + r * ld + n;
      if (full) {
        if constexpr (sizeof(T) == 2) {
          const uint4 u = *reinterpret_cast<const uint4 *>(p);
          const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[2 * e] += __uint_as_float(w[e] << 16);
            acc[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u);
          }
        } else {
          const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
          acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
          acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
      } else {
        for (int e = 0; e < 8 && n + e < N; ++e) acc[e] += ElemIO<T>::ld(p + e);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[rl][cc * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][c];
    if (blockIdx.x * 128 + c < N) atomicAdd(out + blockIdx.x * 128 + c, t);
  }
}

// grouped form: many independent column sums in one launch (the deferred bias gradients of a whole step).
struct ColsumProblem { const void *x; float *out; long long M; int N, ld, dtype, pad; };
struct ColsumItem { int prob, col_block, row_slab, pad; };
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const ColsumProblem *__restrict__ probs, const ColsumItem *__restrict__ items) {
  __shared__ float sh[16][129];
  const ColsumItem it = items[blockIdx.x];
  const ColsumProblem g = probs[it.prob];
  const int cc = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = it.col_block * 128 + cc * 8;
  const int64_t r0 = (int64_t)it.row_slab * CS_ROWS, r1 = r0 + CS_ROWS < g.M ? r0 + CS_ROWS : g.M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < g.N) {
    const bool full = n + 7 < g.N;
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      if (g.dtype == EVP_BF16) {
        const bf16_t *p = (const bf16_t *)g.x + r * g.ld + n;
        if (full) {
          const uint4 u = *reinterpret_cast<const uint4 *>(p);
          const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[2 * e] += __uint_as_float(w[e] << 16);
            acc[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u);
          }
        } else
          for (int e = 0; e < 8 && n + e < g.N; ++e) acc[e] += bf16_to_f32(p[e]);
      } else {
        const float *p = (const float *)g.x + r * g.ld + n;
        if (full) {
          const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
          acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
          acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        } else
          for (int e = 0; e < 8 && n + e < g.N; ++e) acc[e] += p[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[rl][cc * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][c];
    if (it.col_block * 128 + c < g.N) atomicAdd(g.out + it.col_block * 128 + c, t);
  }
}

// ------------------------------------------------------------------------------------------------ patch-embed post-op
template <int VPL>
__global__ __launch_bounds__(LN_THREADS) void embed_post_fwd_kernel(const float *y, const float *gamma, const float *beta,
                                                             const float *pos, const int64_t *ids_keep, int64_t M,
                                                             int n_keep, int L, int D, float eps, float *out, float *mean,
                                                             float *rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; r < M; r += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    Row<VPL> row;
    row.load(y + r * D, D, lane);
    const float mu = row.sum(D, lane) / (float)D;
    const float var = row.sumsq_centered(mu, D, lane) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    const int64_t tok = ids_keep ? ids_keep[r] : (r % L);
    const float *pr = pos ? pos + tok * D : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c * 4);
        const float4 b = *reinterpret_cast<const float4 *>(beta + c * 4);
        const float4 pe = pr ? *reinterpret_cast<const float4 *>(pr + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;
        o.x = gelu_f((row.v[i].x - mu) * rs * g.x + b.x) + pe.x;
        o.y = gelu_f((row.v[i].y - mu) * rs * g.y + b.y) + pe.y;
        o.z = gelu_f((row.v[i].z - mu) * rs * g.z + b.z) + pe.z;
        o.w = gelu_f((row.v[i].w - mu) * rs * g.w + b.w) + pe.w;
        *reinterpret_cast<float4 *>(out + r * D + c * 4) = o;
      }
    }
  }
}

template <int VPL>
__global__ __launch_bounds__(LN_THREADS) void embed_post_bwd_kernel(const float *g_in, const float *y, const float *gamma,
                                                             const float *beta, const float *mean, const float *rstd,
                                                             int64_t M, int D, void *dy, int dy_dtype, float *part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *sh = reinterpret_cast<float *>(smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dg[VPL], db[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; r < M; r += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    Row<VPL> xr, gr;
    xr.load(y + r * D, D, lane);
    gr.load(g_in + r * D, D, lane);
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c * 4);
        const float4 b = *reinterpret_cast<const float4 *>(beta + c * 4);
        float4 xh;
        xh.x = (xr.v[i].x - mu) * rs; xh.y = (xr.v[i].y - mu) * rs;
        xh.z = (xr.v[i].z - mu) * rs; xh.w = (xr.v[i].w - mu) * rs;
        float4 d;  // gradient w.r.t. the LayerNorm output z = xh*gamma+beta
        d.x = gr.v[i].x * dgelu_f(xh.x * g.x + b.x);
        d.y = gr.v[i].y * dgelu_f(xh.y * g.y + b.y);
        d.z = gr.v[i].z * dgelu_f(xh.z * g.z + b.z);
        d.w = gr.v[i].w * dgelu_f(xh.w * g.w + b.w);
        dg[i].x += d.x * xh.x; dg[i].y += d.y * xh.y; dg[i].z += d.z * xh.z; dg[i].w += d.w * xh.w;
        db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        float4 t;
        t.x = d.x * g.x; t.y = d.y * g.y; t.z = d.z * g.z; t.w = d.w * g.w;
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += (t.x * xh.x + t.y * xh.y) + (t.z * xh.z + t.w * xh.w);
        gr.v[i] = t;
        xr.v[i] = xh;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c * 4 < D) {
        float4 o;
        o.x = rs * (gr.v[i].x - s1 - xr.v[i].x * s2);
        o.y = rs * (gr.v[i].y - s1 - xr.v[i].y * s2);
        o.z = rs * (gr.v[i].z - s1 - xr.v[i].z * s2);
        o.w = rs * (gr.v[i].w - s1 - xr.v[i].w * s2);
        store4_any(dy, dy_dtype, r * D + c * 4, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c * 4 < D) {
      *reinterpret_cast<float4 *>(sh + (wave * 2 + 0) * D + c * 4) = dg[i];
      *reinterpret_cast<float4 *>(sh + (wave * 2 + 1) * D + c * 4) = db[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ROWS_PER_BLOCK; ++w) s += sh[w * 2 * D + e];
    part[(int64_t)blockIdx.x * 2 * D + e] = s;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm over rows
constexpr int BN_ROWS = 64;
// Welford per (slab, column): part[slab][0][C] = mean, part[slab][1][C] = M2, count = rows in slab
__global__ __launch_bounds__(256) void bn_stats_partial(const void *x, int dtype, int64_t R, int C, float *part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < R ? r0 + BN_ROWS : R;
  float mean = 0.f, m2 = 0.f;
  int n = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = ld_any(x, dtype, r * C + c);
    ++n;
    const float d = v - mean;
    mean += d / (float)n;
    m2 += d * (v - mean);
  }
  part[((int64_t)blockIdx.y * 2 + 0) * C + c] = mean;
  part[((int64_t)blockIdx.y * 2 + 1) * C + c] = m2;
}
// 1024 threads = 64 columns x 16 slab groups: each thread folds every 16th slab (Chan's pairwise update, in double), the 16 partial
// (n, mean, M2) triples of a column then meet in LDS. (One thread per column walking all R / 64 slabs serially took 68 us at
// R = 12 608, C = 4096: 16 workgroups, a dependent chain of 197 double-precision divisions each.)
__global__ __launch_bounds__(1024) void bn_stats_finalize(const float *part, int nslab, int64_t R, int C, float eps, float momentum, float *mean_out,
                                                          float *invstd_out, float *running_mean, float *running_var) {
  __shared__ double sh[3][16][65];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double mean = 0.0, m2 = 0.0, n = 0.0;
  if (c < C)
    for (int s = g; s < nslab; s += 16) {
      const int64_t r0 = (int64_t)s * BN_ROWS;
      const double nb = (double)((r0 + BN_ROWS < R ? r0 + BN_ROWS : R) - r0);
      const double mb = part[((int64_t)s * 2 + 0) * C + c], m2b = part[((int64_t)s * 2 + 1) * C + c];
      const double delta = mb - mean, tot = n + nb;
      mean += delta * nb / tot;
      m2 += m2b + delta * delta * n * nb / tot;
      n = tot;
    }
  sh[0][g][cl] = n; sh[1][g][cl] = mean; sh[2][g][cl] = m2;
  __syncthreads();
  if (g == 0 && c < C) {
    for (int i = 1; i < 16; ++i) {
      const double nb = sh[0][i][cl];
      if (nb > 0.0) {
        const double delta = sh[1][i][cl] - mean, tot = n + nb;
        mean += delta * nb / tot;
        m2 += sh[2][i][cl] + delta * delta * n * nb / tot;
        n = tot;
      }
    }
    const float var_b = (float)(m2 / n);
    mean_out[c] = (float)mean;
    invstd_out[c] = 1.0f / sqrtf(var_b + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2 / (n > 1.0 ? n - 1.0 : 1.0));
  }
}
__global__ __launch_bounds__(256) void bn_apply(const void *x, int dtype, int64_t total, int C, const float *gamma,
                                                const float *beta, const float *mean, const float *invstd, int relu, void *y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = (ld_any(x, dtype, i) - mean[c]) * invstd[c];
    if (gamma) v = v * gamma[c] + beta[c];
    if (relu) v = fmaxf(v, 0.f);
    st_any(y, dtype, i, v);
  }
}
// part[slab][0][C] = sum dy', part[slab][1][C] = sum dy' * xhat
__global__ __launch_bounds__(256) void bn_bwd_partial(const void *dy, const void *x, const void *y, int dtype, int64_t R, int C,
                                                      const float *mean, const float *invstd, int relu, float *part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < R ? r0 + BN_ROWS : R;
  const float mu = mean[c], is = invstd[c];
  float s0 = 0.f, s1 = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    float d = ld_any(dy, dtype, r * C + c);
    if (relu && !(ld_any(y, dtype, r * C + c) > 0.f)) d = 0.f;
    s0 += d;
    s1 += d * (ld_any(x, dtype, r * C + c) - mu) * is;
  }
  part[((int64_t)blockIdx.y * 2 + 0) * C + c] = s0;
  part[((int64_t)blockIdx.y * 2 + 1) * C + c] = s1;
}
__global__ __launch_bounds__(1024) void bn_bwd_finalize(const float *part, int nslab, int C, float *sum_dy, float *sum_dy_xhat) {
  __shared__ float sh[2][16][65];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int s = g; s < nslab; s += 16) {
      a += part[((int64_t)s * 2 + 0) * C + c];
      b += part[((int64_t)s * 2 + 1) * C + c];
    }
  sh[0][g][cl] = a; sh[1][g][cl] = b;
  __syncthreads();
  if (g == 0 && c < C) {
    for (int i = 1; i < 16; ++i) { a += sh[0][i][cl]; b += sh[1][i][cl]; }
    sum_dy[c] = a;
    sum_dy_xhat[c] = b;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply(const void *dy, const void *x, const void *y, int dtype, int64_t R, int C,
                                                    const float *gamma, const float *mean, const float *invstd, int relu,
                                                    const float *sum_dy, const float *sum_dy_xhat, void *dx) {
  const int64_t total = R * C;
  const float invR = 1.0f / (float)R;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    float d = ld_any(dy, dtype, i);
    if (relu && !(ld_any(y, dtype, i) > 0.f)) d = 0.f;
    const float xh = (ld_any(x, dtype, i) - mean[c]) * invstd[c];
    const float g = gamma ? gamma[c] : 1.f;
    st_any(dx, dtype, i, g * invstd[c] * (d - sum_dy[c] * invR - xh * sum_dy_xhat[c] * invR));
  }
}

// ---- 4-column vector forms (C % 4 == 0): 16-byte (f32) / 8-byte (bf16) accesses instead of one element per lane; a block is 64
// column groups x 4 row lanes over one slab of BN_ROWS rows, the row lanes meet in LDS -------------------------------------------
__device__ __forceinline__ float4 ld4_any(const void *p, int dtype, int64_t i) {
  if (dtype == EVP_BF16) {
    const uint2 u = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(p) + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
  }
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + i);
}
__device__ __forceinline__ void st4_any(void *p, int dtype, int64_t i, float4 v) {
  if (dtype == EVP_BF16) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2 *>(reinterpret_cast<bf16_t *>(p) + i) = u;
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + i) = v;
  }
}
__global__ __launch_bounds__(256) void bn_stats_partial_v4(const void *x, int dtype, int64_t R, int C, float *part) {
  __shared__ float sh[3][4][260];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + cg * 4;
  const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < R ? r0 + BN_ROWS : R;
  float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
  int n = 0;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float4 v4 = ld4_any(x, dtype, r * C + c);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      ++n;
      const float inv = 1.0f / (float)n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[j] - mean[j];
        mean[j] += d * inv;
        m2[j] += d * (v[j] - mean[j]);
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh[0][rl][cg * 4 + j] = (float)n; sh[1][rl][cg * 4 + j] = mean[j]; sh[2][rl][cg * 4 + j] = m2[j]; }
  __syncthreads();
  const int cc = threadIdx.x;                      // one column per thread for the combine
  if (blockIdx.x * 256 + cc < C) {
    float nn = sh[0][0][cc], mu = sh[1][0][cc], q = sh[2][0][cc];
    for (int i = 1; i < 4; ++i) {
      const float nb = sh[0][i][cc];
      if (nb > 0.f) {
        const float delta = sh[1][i][cc] - mu, tot = nn + nb;
        mu += delta * nb / tot;
        q += sh[2][i][cc] + delta * delta * nn * nb / tot;
        nn = tot;
      }
    }
    part[((int64_t)blockIdx.y * 2 + 0) * C + blockIdx.x * 256 + cc] = mu;
    part[((int64_t)blockIdx.y * 2 + 1) * C + blockIdx.x * 256 + cc] = q;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_partial_v4(const void *dy, const void *x, const void *y, int dtype, int64_t R, int C,
                                                         const float *mean, const float *invstd, int relu, float *part) {
  __shared__ float sh[2][4][260];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + cg * 4;
  const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < R ? r0 + BN_ROWS : R;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
    const float mua[4] = {mu.x, mu.y, mu.z, mu.w}, isa[4] = {is.x, is.y, is.z, is.w};
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float4 d4 = ld4_any(dy, dtype, r * C + c), x4 = ld4_any(x, dtype, r * C + c);
      float d[4] = {d4.x, d4.y, d4.z, d4.w};
      const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
      if (relu) {
        const float4 y4 = ld4_any(y, dtype, r * C + c);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yv[j] > 0.f)) d[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[j] += d[j];
        s1[j] += d[j] * (xv[j] - mua[j]) * isa[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh[0][rl][cg * 4 + j] = s0[j]; sh[1][rl][cg * 4 + j] = s1[j]; }
  __syncthreads();
  const int cc = threadIdx.x;
  if (blockIdx.x * 256 + cc < C) {
    part[((int64_t)blockIdx.y * 2 + 0) * C + blockIdx.x * 256 + cc] = (sh[0][0][cc] + sh[0][1][cc]) + (sh[0][2][cc] + sh[0][3][cc]);
    part[((int64_t)blockIdx.y * 2 + 1) * C + blockIdx.x * 256 + cc] = (sh[1][0][cc] + sh[1][1][cc]) + (sh[1][2][cc] + sh[1][3][cc]);
  }
}
__global__ __launch_bounds__(256) void bn_apply_v4(const void *x, int dtype, int64_t total4, int C, const float *gamma, const float *beta,
                                                   const float *mean, const float *invstd, int relu, void *y) {
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
    const int64_t i = i4 * 4;
    const int c = (int)(i % C);
    const float4 v = ld4_any(x, dtype, i), mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
    float o[4] = {(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
    if (gamma) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
      o[0] = o[0] * g.x + b.x; o[1] = o[1] * g.y + b.y; o[2] = o[2] * g.z + b.z; o[3] = o[3] * g.w + b.w;
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    st4_any(y, dtype, i, make_float4(o[0], o[1], o[2], o[3]));
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_v4(const void *dy, const void *x, const void *y, int dtype, int64_t total4, int64_t R, int C,
                                                       const float *gamma, const float *mean, const float *invstd, int relu,
                                                       const float *sum_dy, const float *sum_dy_xhat, void *dx) {
  const float invR = 1.0f / (float)R;
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
    const int64_t i = i4 * 4;
    const int c = (int)(i % C);
    const float4 d4 = ld4_any(dy, dtype, i), x4 = ld4_any(x, dtype, i);
    float d[4] = {d4.x, d4.y, d4.z, d4.w};
    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
    if (relu) {
      const float4 y4 = ld4_any(y, dtype, i);
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(yv[j] > 0.f)) d[j] = 0.f;
    }
    const float4 is4 = *reinterpret_cast<const float4 *>(invstd + c), mu4 = *reinterpret_cast<const float4 *>(mean + c);
    const float4 s04 = *reinterpret_cast<const float4 *>(sum_dy + c), s14 = *reinterpret_cast<const float4 *>(sum_dy_xhat + c);
    const float4 g4 = gamma ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float isa[4] = {is4.x, is4.y, is4.z, is4.w}, mua[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
    const float s0a[4] = {s04.x, s04.y, s04.z, s04.w}, s1a[4] = {s14.x, s14.y, s14.z, s14.w}, ga[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xv[j] - mua[j]) * isa[j];
      o[j] = ga[j] * isa[j] * (d[j] - s0a[j] * invR - xh * s1a[j] * invR);
    }
    st4_any(dx, dtype, i, make_float4(o[0], o[1], o[2], o[3]));
  }
}

}  // namespace

#define DISPATCH_VPL(D, CALL)                                   \
  switch (vpl_for(D)) {                                         \
    case 1: { constexpr int V = 1; CALL; } break;               \
    case 2: { constexpr int V = 2; CALL; } break;               \
    case 4: { constexpr int V = 4; CALL; } break;               \
    case 8: { constexpr int V = 8; CALL; } break;               \
    default: { constexpr int V = 16; CALL; } break;             \
  }

extern "C" int evp_layernorm_fwd(const float *x, const float *x2, const float *x3, const float *gamma, const float *beta,
                                 int64_t M, int D, float eps, void *y, int y_dtype, float *mean, float *rstd, void *stream) {
  EVP_CHECK_ARG(x && gamma && beta && y, EVP_EINVAL, "evp_layernorm_fwd: null pointer");
  EVP_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, EVP_ESHAPE, "evp_layernorm_fwd: need M>0, D%%4==0, D<=4096 (M=%lld D=%d)", (long long)M, D);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_VPL(D, hipLaunchKernelGGL(ln_fwd_kernel<V>, dim3(ln_grid(M, LN_FWD_BLOCKS)), dim3(LN_THREADS), 0, s, x, x2, x3, gamma, beta, M, D, eps, y, y_dtype, mean, rstd));
  EVP_CHECK_LAUNCH("evp_layernorm_fwd");
  return EVP_OK;
}

extern "C" int evp_layernorm_bwd_nblk(int64_t M) { return ln_grid(M); }

extern "C" int evp_layernorm_bwd(const void *dy, int dy_dtype, const float *x, const float *x2, const float *x3,
                                 const float *gamma, const float *mean, const float *rstd, const float *gres, int64_t M, int D,
                                 float *dx, void *dx_lp, float *dgamma, float *dbeta, float *workspace, void *stream) {
  EVP_CHECK_ARG(dy && x && gamma && mean && rstd && workspace, EVP_EINVAL, "evp_layernorm_bwd: null pointer");
  EVP_CHECK_ARG(dx || dx_lp, EVP_EINVAL, "evp_layernorm_bwd: no output requested");
  EVP_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, EVP_ESHAPE, "evp_layernorm_bwd: need D%%4==0, D<=4096 (D=%d)", D);
  hipStream_t s = (hipStream_t)stream;
  const int g = ln_grid(M);
  const size_t sh = (size_t)ROWS_PER_BLOCK * 2 * D * sizeof(float);
  DISPATCH_VPL(D, {
    if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ln_bwd_kernel<V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((ln_bwd_kernel<V, false>), dim3(g), dim3(LN_THREADS), sh, s, dy, dy_dtype, x, x2, x3, gamma, mean, rstd, gres, M, D, dx, dx_lp, workspace);
  });
  EVP_CHECK_LAUNCH("evp_layernorm_bwd");
  if (dgamma || dbeta) {   // both NULL: the caller reduces the per-block partials workspace[g][2][D] itself (deferred, grouped)
    hipLaunchKernelGGL(ln_bwd_finalize, dim3((2 * D + 63) / 64), dim3(1024), 0, s, workspace, g, D, dgamma, dbeta);
    EVP_CHECK_LAUNCH("evp_layernorm_bwd(finalize)");
  }
  return EVP_OK;
}

extern "C" int evp_layernorm_bwd_cs(const void *dy, int dy_dtype, const float *x, const float *x2, const float *x3,
                                    const float *gamma, const float *mean, const float *rstd, const float *gres, int64_t M, int D,
                                    float *dx, void *dx_lp, float *workspace, void *stream) {
  EVP_CHECK_ARG(dy && x && gamma && mean && rstd && workspace, EVP_EINVAL, "evp_layernorm_bwd_cs: null pointer");
  EVP_CHECK_ARG(dx || dx_lp, EVP_EINVAL, "evp_layernorm_bwd_cs: no output requested");
  EVP_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, EVP_ESHAPE, "evp_layernorm_bwd_cs: need D%%4==0, D<=4096 (D=%d)", D);
  hipStream_t s = (hipStream_t)stream;
  const int g = ln_grid(M);
  const size_t sh = (size_t)ROWS_PER_BLOCK * 3 * D * sizeof(float);
  // three partial rows per wave live in LDS: 160 KiB per workgroup is the limit (D <= 3412 at 4 rows per block); wider rows take
  // evp_layernorm_bwd (two partial rows) plus a column sum of dx (ops.layernorm_bwd does that by itself)
  EVP_CHECK_ARG(sh <= 160 * 1024, EVP_ESHAPE, "evp_layernorm_bwd_cs: D=%d needs %zu bytes of LDS (> 160 KiB); use evp_layernorm_bwd", D, sh);
  DISPATCH_VPL(D, {
    if (sh > 48 * 1024) {
      hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(ln_bwd_kernel<V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      EVP_CHECK_ARG(ea == hipSuccess, EVP_ELAUNCH, "evp_layernorm_bwd_cs: hipFuncSetAttribute(%zu) failed: %s", sh, hipGetErrorString(ea));
    }
    hipLaunchKernelGGL((ln_bwd_kernel<V, true>), dim3(g), dim3(LN_THREADS), sh, s, dy, dy_dtype, x, x2, x3, gamma, mean, rstd, gres, M, D, dx, dx_lp, workspace);
  });
  EVP_CHECK_LAUNCH("evp_layernorm_bwd_cs");
  return EVP_OK;
}

extern "C" int evp_colsum_nblk(int64_t M) { return (int)((M + CS_ROWS - 1) / CS_ROWS); }

extern "C" int evp_colsum(const void *x, int x_dtype, int64_t M, int N, int64_t ld, float *out, float *workspace, void *stream) {
  (void)workspace;  // kept in the signature for ABI stability; the reduction now meets in LDS + one atomic per column
  EVP_CHECK_ARG(x && out, EVP_EINVAL, "evp_colsum: null pointer");
  EVP_CHECK_ARG(M > 0 && N > 0 && ld >= N && ld % 8 == 0, EVP_ESHAPE, "evp_colsum: bad shape (ld must be a multiple of 8)");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = evp_zero_async(out, sizeof(float) * N, s);
  EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_colsum: memset failed: %s", hipGetErrorString(e));
  dim3 grid((N + 127) / 128, evp_colsum_nblk(M));
  if (x_dtype == EVP_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)x, M, N, ld, out);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float *)x, M, N, ld, out);
  EVP_CHECK_LAUNCH("evp_colsum");
  return EVP_OK;
}

extern "C" int evp_colsum_grouped(const void *problems, const void *items, int n_items, void *stream) {
  EVP_CHECK_ARG(problems && items && n_items > 0, EVP_EINVAL, "evp_colsum_grouped: bad argument");
  hipLaunchKernelGGL(colsum_grouped_kernel, dim3((unsigned)n_items), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const ColsumProblem *>(problems), reinterpret_cast<const ColsumItem *>(items));
  EVP_CHECK_LAUNCH("evp_colsum_grouped");
  return EVP_OK;
}

extern "C" int evp_embed_post_fwd(const float *y, const float *gamma, const float *beta, const float *pos,
                                  const int64_t *ids_keep, int B, int n_keep, int L, int D, float eps, float *out, float *mean,
                                  float *rstd, void *stream) {
  EVP_CHECK_ARG(y && gamma && beta && out && mean && rstd, EVP_EINVAL, "evp_embed_post_fwd: null pointer");
  EVP_CHECK_ARG(B > 0 && n_keep > 0 && n_keep <= L && D > 0 && D % 4 == 0 && D <= 4096, EVP_ESHAPE, "evp_embed_post_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t M = (int64_t)B * n_keep;
  DISPATCH_VPL(D, hipLaunchKernelGGL(embed_post_fwd_kernel<V>, dim3(ln_grid(M)), dim3(LN_THREADS), 0, s, y, gamma, beta, pos, ids_keep, M, n_keep, L, D, eps, out, mean, rstd));
  EVP_CHECK_LAUNCH("evp_embed_post_fwd");
  return EVP_OK;
}

extern "C" int evp_embed_post_bwd(const float *g, const float *y, const float *gamma, const float *beta, const float *mean,
                                  const float *rstd, int64_t M, int D, void *dy, int dy_dtype, float *dgamma, float *dbeta,
                                  float *workspace, void *stream) {
  EVP_CHECK_ARG(g && y && gamma && beta && mean && rstd && dy && workspace, EVP_EINVAL, "evp_embed_post_bwd: null pointer");
  EVP_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, EVP_ESHAPE, "evp_embed_post_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int gsz = ln_grid(M);
  const size_t sh = (size_t)ROWS_PER_BLOCK * 2 * D * sizeof(float);
  DISPATCH_VPL(D, {
    if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(embed_post_bwd_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(embed_post_bwd_kernel<V>, dim3(gsz), dim3(LN_THREADS), sh, s, g, y, gamma, beta, mean, rstd, M, D, dy, dy_dtype, workspace);
  });
  EVP_CHECK_LAUNCH("evp_embed_post_bwd");
  if (dgamma || dbeta) {   // both NULL: the caller reduces the per-block partials workspace[g][2][D] itself (deferred, grouped)
    hipLaunchKernelGGL(ln_bwd_finalize, dim3((2 * D + 63) / 64), dim3(1024), 0, s, workspace, gsz, D, dgamma, dbeta);
    EVP_CHECK_LAUNCH("evp_embed_post_bwd(finalize)");
  }
  return EVP_OK;
}

extern "C" int evp_batchnorm_nblk(int64_t R) { return (int)((R + BN_ROWS - 1) / BN_ROWS); }

extern "C" int evp_batchnorm_fwd(const void *x, int dtype, int64_t R, int C, const float *gamma, const float *beta, float eps,
                                 float momentum, int relu, void *y, float *mean, float *invstd, float *running_mean,
                                 float *running_var, float *workspace, void *stream) {
  EVP_CHECK_ARG(x && y && mean && invstd && workspace, EVP_EINVAL, "evp_batchnorm_fwd: null pointer");
  EVP_CHECK_ARG(R > 1 && C > 0, EVP_ESHAPE, "evp_batchnorm_fwd: need R>1 rows");
  EVP_CHECK_ARG((gamma == nullptr) == (beta == nullptr), EVP_EINVAL, "evp_batchnorm_fwd: gamma and beta go together");
  hipStream_t s = (hipStream_t)stream;
  const int ns = evp_batchnorm_nblk(R);
  const bool v4 = (C % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  if (v4) hipLaunchKernelGGL(bn_stats_partial_v4, dim3((C + 255) / 256, ns), dim3(256), 0, s, x, dtype, R, C, workspace);
  else hipLaunchKernelGGL(bn_stats_partial, dim3((C + 255) / 256, ns), dim3(256), 0, s, x, dtype, R, C, workspace);
  EVP_CHECK_LAUNCH("evp_batchnorm_fwd(stats)");
  hipLaunchKernelGGL(bn_stats_finalize, dim3((C + 63) / 64), dim3(1024), 0, s, workspace, ns, R, C, eps, momentum, mean, invstd, running_mean, running_var);
  EVP_CHECK_LAUNCH("evp_batchnorm_fwd(finalize)");
  const int64_t total = R * C;
  int64_t g = ((v4 ? total / 4 : total) + 255) / 256; if (g > 4096) g = 4096;
  if (v4) hipLaunchKernelGGL(bn_apply_v4, dim3((int)g), dim3(256), 0, s, x, dtype, total / 4, C, gamma, beta, mean, invstd, relu, y);
  else hipLaunchKernelGGL(bn_apply, dim3((int)g), dim3(256), 0, s, x, dtype, total, C, gamma, beta, mean, invstd, relu, y);
  EVP_CHECK_LAUNCH("evp_batchnorm_fwd(apply)");
  return EVP_OK;
}

extern "C" int evp_batchnorm_bwd(const void *dy, const void *x, const void *y, int dtype, int64_t R, int C, const float *gamma,
                                 const float *mean, const float *invstd, int relu, void *dx, float *dgamma, float *dbeta,
                                 float *workspace, void *stream) {
  EVP_CHECK_ARG(dy && x && mean && invstd && dx && workspace, EVP_EINVAL, "evp_batchnorm_bwd: null pointer");
  EVP_CHECK_ARG(!relu || y, EVP_EINVAL, "evp_batchnorm_bwd: relu needs y");
  hipStream_t s = (hipStream_t)stream;
  const int ns = evp_batchnorm_nblk(R);
  float *sum_dy = workspace + (int64_t)ns * 2 * C, *sum_dy_xhat = sum_dy + C;
  const bool v4 = (C % 4 == 0) && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx) & 15) == 0;
  if (v4) hipLaunchKernelGGL(bn_bwd_partial_v4, dim3((C + 255) / 256, ns), dim3(256), 0, s, dy, x, y, dtype, R, C, mean, invstd, relu, workspace);
  else hipLaunchKernelGGL(bn_bwd_partial, dim3((C + 255) / 256, ns), dim3(256), 0, s, dy, x, y, dtype, R, C, mean, invstd, relu, workspace);
  EVP_CHECK_LAUNCH("evp_batchnorm_bwd(partial)");
  hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 63) / 64), dim3(1024), 0, s, workspace, ns, C, sum_dy, sum_dy_xhat);
  EVP_CHECK_LAUNCH("evp_batchnorm_bwd(finalize)");
  const int64_t total = R * C;
  int64_t g = ((v4 ? total / 4 : total) + 255) / 256; if (g > 4096) g = 4096;
  if (v4) hipLaunchKernelGGL(bn_bwd_apply_v4, dim3((int)g), dim3(256), 0, s, dy, x, y, dtype, total / 4, R, C, gamma, mean, invstd, relu, sum_dy, sum_dy_xhat, dx);
  else hipLaunchKernelGGL(bn_bwd_apply, dim3((int)g), dim3(256), 0, s, dy, x, y, dtype, R, C, gamma, mean, invstd, relu, sum_dy, sum_dy_xhat, dx);
  EVP_CHECK_LAUNCH("evp_batchnorm_bwd(apply)");
  hipError_t e = hipSuccess;
  if (dgamma) e = hipMemcpyAsync(dgamma, sum_dy_xhat, sizeof(float) * C, hipMemcpyDeviceToDevice, s);
  if (dbeta && e == hipSuccess) e = hipMemcpyAsync(dbeta, sum_dy, sizeof(float) * C, hipMemcpyDeviceToDevice, s);
  EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_batchnorm_bwd: copying the affine gradients failed: %s", hipGetErrorString(e));
  return EVP_OK;
}
