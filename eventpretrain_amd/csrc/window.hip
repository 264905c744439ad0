// Swin path: grouped window attention with the gathered relative-position bias, token row
// gathers (window grouping / merging / 2x2 patch merging), the stage-fusion patch gathers, and the host-side
// window grouping (greedy 0/1 knapsack).
//
// Reference behaviour restated (never copied): model/sub_module/swin_block.py:124-162 (WindowAttention.forward),
// :277-347 (knapsack / group_windows), :454-466 (group / merge), :179-209 (PatchMerging), model/backbone/swin.py:190-236
// (re-densify + stage convs + gather by ids_keep).
//
// Every head of Swin-T has d_h = 32 and a group holds at most 2*7*7 = 98 tokens, so one workgroup owns one
// (group instance, head): q/k/v and the full score matrix live in LDS and the arithmetic is plain f32 FMA --
// the whole stage is a few hundred MFLOP, far below anything worth an MFMA pipeline, and f32 keeps the -100 mask
// arithmetic identical to the reference's.
#include "evp_common.h"

#include <vector>

#define WA_THREADS 256
#define WA_DH 32
#define WA_LD 33  // padded LDS row stride (floats): k[j][d] reads across j hit distinct banks
#define WA_MAX_N 128
#define WA_MAX_R 512

template <typename T>
__device__ __forceinline__ void wa_load_rows(const T *src, float *dst, int N, int row_stride, float mul) {
  for (int e = threadIdx.x; e < N * WA_DH; e += WA_THREADS) {
    const int n = e >> 5, d = e & 31;
    dst[n * WA_LD + d] = ElemIO<T>::ld(src + (int64_t)n * row_stride + d) * mul;
  }
}

// S[i][j] = qs_i . k_j + (rel >= 0 ? table[rel][h] : -100); then row softmax in place.
__device__ __forceinline__ void wa_scores_softmax(const float *q, const float *k, float *S, const float *table, const int32_t *relg,
                                                  int N, int H, int h) {
  for (int e = threadIdx.x; e < N * N; e += WA_THREADS) {
    const int i = e / N, j = e - i * N;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < WA_DH; ++d) s = fmaf(q[i * WA_LD + d], k[j * WA_LD + d], s);
    const int r = relg[e];
    S[e] = s + (r >= 0 ? table[r * H + h] : -100.0f);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < N; i += WA_THREADS / 64) {
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, S[i * N + j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float p = expf(S[i * N + j] - m);
      S[i * N + j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < N; j += 64) S[i * N + j] *= inv;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(WA_THREADS) void win_attn_fwd_kernel(const T *__restrict__ qkv, const float *__restrict__ table,
                                                                  const int32_t *__restrict__ rel, T *__restrict__ out,
                                                                  float *__restrict__ probs, int nG, int N, int H, float scale,
                                                                  const uint8_t *__restrict__ keep, float keep_scale) {
  extern __shared__ __attribute__((aligned(16))) float wa_sm[];
  float *q = wa_sm, *k = q + N * WA_LD, *v = k + N * WA_LD, *S = v + N * WA_LD;
  const int bg = blockIdx.x / H, h = blockIdx.x - bg * H;
  const int g = bg % nG;
  const int row = 3 * H * WA_DH;
  const T *base = qkv + (int64_t)bg * N * row + h * WA_DH;
  wa_load_rows(base, q, N, row, scale);
  wa_load_rows(base + H * WA_DH, k, N, row, 1.0f);
  wa_load_rows(base + 2 * H * WA_DH, v, N, row, 1.0f);
  __syncthreads();
  wa_scores_softmax(q, k, S, table, rel + (int64_t)g * N * N, N, H, h);
  if (keep) {  // attn_drop (swin_block.py:113,152): P <- P * keep / (1 - p); the returned map is the dropped one (:157)
    const uint8_t *kp = keep + (int64_t)blockIdx.x * N * N;
    for (int e = threadIdx.x; e < N * N; e += WA_THREADS) S[e] = kp[e] ? S[e] * keep_scale : 0.f;
    __syncthreads();
  }
  if (probs) {
    float *pp = probs + (int64_t)blockIdx.x * N * N;
    for (int e = threadIdx.x; e < N * N; e += WA_THREADS) pp[e] = S[e];
  }
  for (int e = threadIdx.x; e < N * WA_DH; e += WA_THREADS) {
    const int i = e >> 5, d = e & 31;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(S[i * N + j], v[j * WA_LD + d], acc);
    ElemIO<T>::st(out + ((int64_t)bg * N + i) * H * WA_DH + h * WA_DH + d, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(WA_THREADS) void win_attn_bwd_kernel(const T *__restrict__ qkv, const float *__restrict__ table,
                                                                  const int32_t *__restrict__ rel, const T *__restrict__ out,
                                                                  const T *__restrict__ dout, T *__restrict__ dqkv,
                                                                  float *__restrict__ dtable, int nG, int N, int H, int R, float scale,
                                                                  const uint8_t *__restrict__ keep, float keep_scale) {
  extern __shared__ __attribute__((aligned(16))) float wa_sm[];
  float *q = wa_sm, *k = q + N * WA_LD, *v = k + N * WA_LD, *dO = v + N * WA_LD, *S = dO + N * WA_LD;
  float *delta = S + N * N, *tab = delta + N;
  const int bg = blockIdx.x / H, h = blockIdx.x - bg * H;
  const int g = bg % nG;
  const int row = 3 * H * WA_DH, orow = H * WA_DH;
  const T *base = qkv + (int64_t)bg * N * row + h * WA_DH;
  const int32_t *relg = rel + (int64_t)g * N * N;
  wa_load_rows(base, q, N, row, scale);
  wa_load_rows(base + H * WA_DH, k, N, row, 1.0f);
  wa_load_rows(base + 2 * H * WA_DH, v, N, row, 1.0f);
  wa_load_rows(dout + (int64_t)bg * N * orow + h * WA_DH, dO, N, orow, 1.0f);
  for (int r = threadIdx.x; r < R; r += WA_THREADS) tab[r] = 0.f;
  {  // delta_i = dO_i . O_i  (== sum_j P_ij dP_ij)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T *ob = out + (int64_t)bg * N * orow + h * WA_DH;
    const T *gb = dout + (int64_t)bg * N * orow + h * WA_DH;
    for (int i = wave; i < N; i += WA_THREADS / 64) {
      float t = 0.f;
      if (lane < WA_DH) t = ElemIO<T>::ld(ob + (int64_t)i * orow + lane) * ElemIO<T>::ld(gb + (int64_t)i * orow + lane);
      t = wave_sum(t);
      if (lane == 0) delta[i] = t;
    }
  }
  __syncthreads();
  wa_scores_softmax(q, k, S, table, relg, N, H, h);
  T *dbase = dqkv + (int64_t)bg * N * row + h * WA_DH;
  // with attn_drop: O = P' V, P' = P * keep / (1 - p); dV takes P', dP = keep / (1 - p) * dP', and delta_i = dO_i . O_i still equals
  // sum_j P_ij dP_ij
  const uint8_t *kp = keep ? keep + (int64_t)blockIdx.x * N * N : nullptr;
  // dV[j][d] = sum_i P'[i][j] dO[i][d]
  for (int e = threadIdx.x; e < N * WA_DH; e += WA_THREADS) {
    const int j = e >> 5, d = e & 31;
    float acc = 0.f;
    if (kp) {
      for (int i = 0; i < N; ++i) acc = fmaf(kp[i * N + j] ? S[i * N + j] * keep_scale : 0.f, dO[i * WA_LD + d], acc);
    } else
      for (int i = 0; i < N; ++i) acc = fmaf(S[i * N + j], dO[i * WA_LD + d], acc);
    ElemIO<T>::st(dbase + (int64_t)j * row + 2 * H * WA_DH + d, acc);
  }
  __syncthreads();
  // dS = P * (dO_i . v_j - delta_i), in place; allowed pairs also feed the relative-position table gradient
  for (int e = threadIdx.x; e < N * N; e += WA_THREADS) {
    const int i = e / N, j = e - i * N;
    float dp = 0.f;
#pragma unroll
    for (int d = 0; d < WA_DH; ++d) dp = fmaf(dO[i * WA_LD + d], v[j * WA_LD + d], dp);
    if (kp) dp = kp[e] ? dp * keep_scale : 0.f;
    const float ds = S[e] * (dp - delta[i]);
    S[e] = ds;
    const int r = relg[e];
    if (r >= 0) atomicAdd(&tab[r], ds);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * WA_DH; e += WA_THREADS) {
    const int i = e >> 5, d = e & 31;
    float aq = 0.f, ak = 0.f;
    for (int j = 0; j < N; ++j) {
      aq = fmaf(S[i * N + j], k[j * WA_LD + d], aq);   // dQ[i][d]
      ak = fmaf(S[j * N + i], q[j * WA_LD + d], ak);   // dK[i][d] = sum_j dS[j][i] qs[j][d]
    }
    ElemIO<T>::st(dbase + (int64_t)i * row + d, aq * scale);
    ElemIO<T>::st(dbase + (int64_t)i * row + H * WA_DH + d, ak);
  }
  for (int r = threadIdx.x; r < R; r += WA_THREADS) {
    const float t = tab[r];
    if (t != 0.f) atomicAdd(dtable + (int64_t)r * H + h, t);
  }
}

static int wa_check(const char *fn, int Bg, int nG, int N, int H, int R, int dtype) {
  EVP_CHECK_ARG(dtype == EVP_F32 || dtype == EVP_BF16, EVP_EINVAL, "%s: dtype %d", fn, dtype);
  EVP_CHECK_ARG(Bg > 0 && nG > 0 && Bg % nG == 0, EVP_EINVAL, "%s: Bg=%d must be a positive multiple of nG=%d", fn, Bg, nG);
  EVP_CHECK_ARG(N > 0 && N <= WA_MAX_N, EVP_EINVAL, "%s: N=%d outside 1..%d", fn, N, WA_MAX_N);
  EVP_CHECK_ARG(H > 0 && R > 0 && R <= WA_MAX_R, EVP_EINVAL, "%s: H=%d R=%d", fn, H, R);
  return EVP_OK;
}

extern "C" int evp_window_attention_fwd(const void *qkv, const float *table, const int32_t *rel, void *out, float *probs, int Bg, int nG,
                                        int N, int H, int R, float scale, int dtype, const void *keep, float keep_scale, void *stream) {
  int rc = wa_check("evp_window_attention_fwd", Bg, nG, N, H, R, dtype);
  if (rc) return rc;
  EVP_CHECK_ARG(qkv && table && rel && out, EVP_EINVAL, "evp_window_attention_fwd: null pointer");
  const size_t smem = (size_t)(3 * N * WA_LD + N * N) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(Bg * H));
  if (dtype == EVP_F32) {
    auto kfn = win_attn_fwd_kernel<float>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kfn, grid, dim3(WA_THREADS), smem, st, (const float *)qkv, table, rel, (float *)out, probs, nG, N, H, scale,
                       (const uint8_t *)keep, keep_scale);
  } else {
    auto kfn = win_attn_fwd_kernel<bf16_t>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kfn, grid, dim3(WA_THREADS), smem, st, (const bf16_t *)qkv, table, rel, (bf16_t *)out, probs, nG, N, H, scale,
                       (const uint8_t *)keep, keep_scale);
  }
  EVP_CHECK_LAUNCH("evp_window_attention_fwd");
  return EVP_OK;
}

extern "C" int evp_window_attention_bwd(const void *qkv, const float *table, const int32_t *rel, const void *out, const void *dout,
                                        void *dqkv, float *dtable, int Bg, int nG, int N, int H, int R, float scale, int dtype,
                                        const void *keep, float keep_scale, void *stream) {
  int rc = wa_check("evp_window_attention_bwd", Bg, nG, N, H, R, dtype);
  if (rc) return rc;
  EVP_CHECK_ARG(qkv && table && rel && out && dout && dqkv && dtable, EVP_EINVAL, "evp_window_attention_bwd: null pointer");
  const size_t smem = (size_t)(4 * N * WA_LD + N * N + N + R) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = evp_zero_async(dtable, (size_t)R * H * sizeof(float), st);
  EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_window_attention_bwd: memset: %s", hipGetErrorString(e));
  const dim3 grid((unsigned)(Bg * H));
  if (dtype == EVP_F32) {
    auto kfn = win_attn_bwd_kernel<float>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kfn, grid, dim3(WA_THREADS), smem, st, (const float *)qkv, table, rel, (const float *)out, (const float *)dout,
                       (float *)dqkv, dtable, nG, N, H, R, scale, (const uint8_t *)keep, keep_scale);
  } else {
    auto kfn = win_attn_bwd_kernel<bf16_t>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kfn, grid, dim3(WA_THREADS), smem, st, (const bf16_t *)qkv, table, rel, (const bf16_t *)out, (const bf16_t *)dout,
                       (bf16_t *)dqkv, dtable, nG, N, H, R, scale, (const uint8_t *)keep, keep_scale);
  }
  EVP_CHECK_LAUNCH("evp_window_attention_bwd");
  return EVP_OK;
}

// ---------------------------------------------------------------------------------------------------- row gathers
// out[b, s, :] = idx[s] >= 0 ? x[b, idx[s], :] : 0   (idx shared by the batch, or per sample when idx_bstride != 0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4 *__restrict__ x, const int32_t *__restrict__ idx,
                                                          float4 *__restrict__ out, int64_t total, int n_in, int n_out, int C4,
                                                          int idx_bstride) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C4);
  const int64_t rs = e / C4;
  const int s = (int)(rs % n_out);
  const int64_t b = rs / n_out;
  const int t = idx[b * idx_bstride + s];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t >= 0) v = x[((int64_t)b * n_in + t) * C4 + c];
  out[e] = v;
}

extern "C" int evp_gather_rows_f32(const float *x, const int32_t *idx, float *out, int B, int n_in, int n_out, int C, int idx_per_sample,
                                   void *stream) {
  EVP_CHECK_ARG(x && idx && out, EVP_EINVAL, "evp_gather_rows_f32: null pointer");
  EVP_CHECK_ARG(B > 0 && n_in > 0 && n_out > 0 && C > 0 && C % 4 == 0, EVP_EINVAL, "evp_gather_rows_f32: B=%d n_in=%d n_out=%d C=%d (C%%4==0)",
                B, n_in, n_out, C);
  const int64_t total = (int64_t)B * n_out * (C / 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, idx,
                     (float4 *)out, total, n_in, n_out, C / 4, idx_per_sample ? n_out : 0);
  EVP_CHECK_LAUNCH("evp_gather_rows_f32");
  return EVP_OK;
}

// ---------------------------------------------------------------------------------------------------- stage fusion
// A[(b*K + j), c*k*k + ky*k + kx] = token at dense position (cy*k+ky, cx*k+kx) of the R x R grid, or 0 when that
// position is hidden; (cy, cx) = cell ids_keep[b][j] of the g x g decoder grid (swin.py:201-208: zeros grid, scatter,
// Conv2d(k, stride k), gather by ids_keep -- only the gathered rows are ever formed).
__global__ __launch_bounds__(256) void fuse_gather_kernel(const float *__restrict__ x, const int32_t *__restrict__ tokmap,
                                                          const int64_t *__restrict__ ids_keep, float *__restrict__ A, int64_t total,
                                                          int n, int K, int C, int R, int k, int g) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int kk = k * k;
  const int kykx = (int)(e % kk);
  const int64_t r1 = e / kk;
  const int c = (int)(r1 % C);
  const int64_t bj = r1 / C;
  const int64_t b = bj / K;
  const int cell = (int)ids_keep[bj];
  const int cy = cell / g, cx = cell - cy * g;
  const int ky = kykx / k, kx = kykx - ky * k;
  const int t = tokmap[(cy * k + ky) * R + cx * k + kx];
  A[e] = t >= 0 ? x[((int64_t)b * n + t) * C + c] : 0.f;
}

// dx[b, t, c] = dA[(b*K + j), c*k*k + ky*k + kx] with j = ids_restore[b][cell(t)] when j < K, else 0.
__global__ __launch_bounds__(256) void fuse_gather_bwd_kernel(const float *__restrict__ dA, const int32_t *__restrict__ coords,
                                                              const int64_t *__restrict__ ids_restore, float *__restrict__ dx,
                                                              int64_t total, int n, int K, int C, int k, int g) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int64_t bt = e / C;
  const int t = (int)(bt % n);
  const int64_t b = bt / n;
  const int y = coords[2 * t], xx = coords[2 * t + 1];
  const int cell = (y / k) * g + xx / k;
  const int64_t j = ids_restore[b * g * g + cell];
  float v = 0.f;
  if (j < K) v = dA[((int64_t)b * K + j) * C * k * k + (int64_t)c * k * k + (y % k) * k + (xx % k)];
  dx[e] = v;
}

extern "C" int evp_swin_fuse_gather_f32(const float *x, const int32_t *tokmap, const int64_t *ids_keep, float *A, int B, int n, int K, int C,
                                        int R, int k, void *stream) {
  EVP_CHECK_ARG(x && tokmap && ids_keep && A, EVP_EINVAL, "evp_swin_fuse_gather_f32: null pointer");
  EVP_CHECK_ARG(B > 0 && n > 0 && K > 0 && C > 0 && k > 0 && R > 0 && R % k == 0, EVP_EINVAL,
                "evp_swin_fuse_gather_f32: B=%d n=%d K=%d C=%d R=%d k=%d", B, n, K, C, R, k);
  const int64_t total = (int64_t)B * K * C * k * k;
  hipLaunchKernelGGL(fuse_gather_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, tokmap, ids_keep, A,
                     total, n, K, C, R, k, R / k);
  EVP_CHECK_LAUNCH("evp_swin_fuse_gather_f32");
  return EVP_OK;
}

extern "C" int evp_swin_fuse_gather_bwd_f32(const float *dA, const int32_t *coords, const int64_t *ids_restore, float *dx, int B, int n, int K,
                                            int C, int R, int k, void *stream) {
  EVP_CHECK_ARG(dA && coords && ids_restore && dx, EVP_EINVAL, "evp_swin_fuse_gather_bwd_f32: null pointer");
  EVP_CHECK_ARG(B > 0 && n > 0 && K > 0 && C > 0 && k > 0 && R > 0 && R % k == 0, EVP_EINVAL,
                "evp_swin_fuse_gather_bwd_f32: B=%d n=%d K=%d C=%d R=%d k=%d", B, n, K, C, R, k);
  const int64_t total = (int64_t)B * n * C;
  hipLaunchKernelGGL(fuse_gather_bwd_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, dA, coords,
                     ids_restore, dx, total, n, K, C, k, R / k);
  EVP_CHECK_LAUNCH("evp_swin_fuse_gather_bwd_f32");
  return EVP_OK;
}

// ---------------------------------------------------------------------------------------------------- host grouping
// Greedy packing of windows into groups of at most `cap` tokens: repeat a 0/1 knapsack (value == weight) over the
// windows still unassigned; back-tracking walks from the last item and takes an item whenever the table value
// differs from the row above, so ties resolve exactly as in swin_block.py:303-317. Pure host code, no GPU.
extern "C" int evp_swin_group_windows(const int32_t *counts, int n_windows, int cap, int32_t *group_of_window, int32_t *group_sizes,
                                      int32_t *n_groups) {
  EVP_CHECK_ARG(counts && group_of_window && group_sizes && n_groups, EVP_EINVAL, "evp_swin_group_windows: null pointer");
  EVP_CHECK_ARG(n_windows >= 0 && cap > 0, EVP_EINVAL, "evp_swin_group_windows: n_windows=%d cap=%d", n_windows, cap);
  for (int i = 0; i < n_windows; ++i)
    EVP_CHECK_ARG(counts[i] > 0 && counts[i] <= cap, EVP_EINVAL, "evp_swin_group_windows: counts[%d]=%d outside 1..%d", i, counts[i], cap);
  std::vector<int> left(n_windows);
  for (int i = 0; i < n_windows; ++i) left[i] = i;
  std::vector<int> tab;
  int ng = 0;
  while (!left.empty()) {
    const int n = (int)left.size(), W = cap + 1;
    tab.assign((size_t)(n + 1) * W, 0);
    for (int i = 1; i <= n; ++i) {
      const int w = counts[left[i - 1]];
      const int *prev = &tab[(size_t)(i - 1) * W];
      int *cur = &tab[(size_t)i * W];
      for (int c = 0; c < W; ++c) {
        int best = prev[c];
        if (w <= c && prev[c - w] + w > best) best = prev[c - w] + w;
        cur[c] = best;
      }
    }
    int res = tab[(size_t)n * W + cap], c = cap;
    group_sizes[ng] = res;
    std::vector<char> taken(n, 0);
    for (int i = n; i > 0 && res > 0; --i) {
      if (res == tab[(size_t)(i - 1) * W + c]) continue;
      taken[i - 1] = 1;
      res -= counts[left[i - 1]];
      c -= counts[left[i - 1]];
    }
    std::vector<int> rest;
    for (int i = 0; i < n; ++i) {
      if (taken[i]) group_of_window[left[i]] = ng;
      else rest.push_back(left[i]);
    }
    left.swap(rest);
    ++ng;
  }
  *n_groups = ng;
  return EVP_OK;
}
