// Token bookkeeping kernels: masking ids (K2), patch gather (K3 front half), decoder unshuffle (K10) and small
// element-wise helpers. Integer outputs are bit-exact with the reference by construction (stable rank counting).
#include "evp_common.h"
#include <stdarg.h>

// ---- thread-local error string (shared by the whole library) ------------------------------------------------
static thread_local char g_err[512] = "";
void evp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *evp_last_error(void) { return g_err; }
extern "C" int evp_abi_version(void) { return EVP_ABI_VERSION; }
extern "C" const char *evp_target_arch(void) { return "gfx950"; }

namespace {

// a sorts strictly before b in torch's ascending order (NaN last)
__device__ __forceinline__ bool key_lt(float a, float b) { return (a < b) || (!(a != a) && (b != b)); }
__device__ __forceinline__ bool key_eq(float a, float b) { return (a == b) || ((a != a) && (b != b)); }

// One block per sample: rank[i] = #{j : noise[j] < noise[i]  or (== and j < i)} -> this IS ids_restore[i]
// (vit.py:92: argsort of the argsort); ids_shuffle[rank[i]] = i; mask[i] = rank[i] >= len_keep (vit.py:99-103).
__global__ __launch_bounds__(256) void mask_kernel(const float *noise, int L, int len_keep, int64_t *ids_keep, float *mask,
                                                   int64_t *ids_restore) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *sh = reinterpret_cast<float *>(smem_raw);
  const int b = blockIdx.x;
  const float *nz = noise + (int64_t)b * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) sh[i] = nz[i];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float v = sh[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float u = sh[j];
      rank += (key_lt(u, v) || (key_eq(u, v) && j < i)) ? 1 : 0;
    }
    ids_restore[(int64_t)b * L + i] = rank;
    mask[(int64_t)b * L + i] = rank >= len_keep ? 1.0f : 0.0f;
    if (rank < len_keep) ids_keep[(int64_t)b * len_keep + rank] = i;
  }
}

// vit.py:80-89: noise[b, gy*gw+gx] = sign * mean_{patch}( |sum_c x[b,c,y,x]| ).
// The ids that follow are an argsort of these values, and on real voxel grids |sum over bins| is near-integer, so ties
// and one-ulp near-ties are the norm: the value must be the reference's BIT FOR BIT. Its ATen composition on the CPU is
// torch.sum(dim=1) = bins added in order 0..C-1, abs, AvgPool2d = the window's p*p values added one after the other in
// row-major order in f32, then ONE division by p*p (checked bit-exact against the reference, tests/golden/
// masking_density.npz). So: the per-pixel values go to LDS and ONE lane adds them in that order -- a wave-tree sum
// differs in the last bit and flips ids.
__global__ __launch_bounds__(256) void density_kernel(const float *x, int C, int H, int W, int p, float sign, float *noise) {
  extern __shared__ float pix[];           // p*p per-pixel |sum over bins|
  const int gw = W / p, gh = H / p;
  const int cell = blockIdx.x % (gw * gh), b = blockIdx.x / (gw * gh);
  const int gy = cell / gw, gx = cell % gw;
  for (int e = threadIdx.x; e < p * p; e += blockDim.x) {
    const int py = e / p, px = e % p;
    const float *src = x + (((int64_t)b * C) * H + gy * p + py) * W + gx * p + px;
    float t = src[0];
    for (int c = 1; c < C; ++c) t += src[(int64_t)c * H * W];
    pix[e] = fabsf(t);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int e = 0; e < p * p; ++e) s += pix[e];
    const float m = s / (float)(p * p);
    noise[(int64_t)b * gw * gh + cell] = sign < 0.f ? -m : m;
  }
}

// cols[(b, j), c*p*p + py*p + px] = x[b, c, gy*p+py, gx*p+px], token = ids_keep[b,j] (or j) -> (gy, gx)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float *x, const int64_t *ids_keep, int C, int H, int W, int p,
                                                       int n_keep, T *cols) {
  const int row = blockIdx.x;  // b * n_keep + j
  const int b = row / n_keep;
  const int gw = W / p;
  const int tok = ids_keep ? (int)ids_keep[row] : (row % n_keep);
  const int gy = tok / gw, gx = tok % gw;
  const int Kc = C * p * p;
  const float *xb = x + (int64_t)b * C * H * W;
  T *o = cols + (int64_t)row * Kc;
  // 4 consecutive px per thread (p % 4 == 0): 16-byte loads, 8/16-byte stores
  for (int e = threadIdx.x * 4; e < Kc; e += blockDim.x * 4) {
    const int c = e / (p * p), rem = e % (p * p), py = rem / p, px = rem % p;
    const float4 v = *reinterpret_cast<const float4 *>(xb + ((int64_t)c * H + gy * p + py) * W + gx * p + px);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4 *>(o + e) = v;
    else {
      uint2 u;
      u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
      u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
      *reinterpret_cast<uint2 *>(o + e) = u;
    }
  }
}

// out[b,l,:] = (ids_restore[b,l] < n_keep ? emb[b, ids_restore[b,l], :] : mask_token) + pos[l]
__global__ __launch_bounds__(256) void unshuffle_fwd_kernel(const float *emb, const float *mask_token, const float *pos,
                                                            const int64_t *ids_restore, int n_keep, int L, int D, float *out) {
  const int64_t row = blockIdx.x;  // b*L + l
  const int l = (int)(row % L);
  const int64_t b = row / L;
  const int64_t src = ids_restore[row];
  const float *s = src < n_keep ? emb + (b * n_keep + src) * D : mask_token;
  const float *pe = pos + (int64_t)l * D;
  float *o = out + row * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    const float4 a = *reinterpret_cast<const float4 *>(s + d);
    const float4 q = *reinterpret_cast<const float4 *>(pe + d);
    *reinterpret_cast<float4 *>(o + d) = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
  }
}
// out[b,j,:] = x[b,j,:] + table[ids[b,j] (or j), :]
__global__ __launch_bounds__(256) void add_rows_gather_kernel(const float *x, const float *table, const int64_t *ids, int n, int L, int D,
                                                             float *out) {
  const int64_t row = blockIdx.x;
  const int64_t tok = ids ? ids[row] : (row % L);
  const float *s = x + row * D, *t = table + tok * D;
  float *o = out + row * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    const float4 a = *reinterpret_cast<const float4 *>(s + d), q = *reinterpret_cast<const float4 *>(t + d);
    *reinterpret_cast<float4 *>(o + d) = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
  }
}
// demb[b, ids_restore[b,l], :] = g[b,l,:] for kept positions (ids_restore is a permutation: no write conflicts)
__global__ __launch_bounds__(256) void unshuffle_bwd_scatter(const float *g, const int64_t *ids_restore, int n_keep, int L, int D,
                                                             float *demb) {
  const int64_t row = blockIdx.x;  // b*L + l
  const int64_t b = row / L;
  const int64_t j = ids_restore[row];
  if (j >= n_keep) return;
  const float *s = g + row * D;
  float *o = demb + (b * n_keep + j) * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4)
    *reinterpret_cast<float4 *>(o + d) = *reinterpret_cast<const float4 *>(s + d);
}
// partial sums of g over removed positions (ids_restore >= n_keep): part[blk][D]
constexpr int UM_ROWS = 64;
__global__ __launch_bounds__(256) void unshuffle_bwd_masktok(const float *g, const int64_t *ids_restore, int64_t rows, int n_keep,
                                                             int D, float *part) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int64_t r0 = (int64_t)blockIdx.y * UM_ROWS, r1 = r0 + UM_ROWS < rows ? r0 + UM_ROWS : rows;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r)
    if (ids_restore[r] >= n_keep) s += g[r * D + d];
  part[(int64_t)blockIdx.y * D + d] = s;
}
// out[n] = sum_b part[b][n]. 1024 threads = 64 columns x 16 partial-row groups: coalesced 256-byte reads, the nblk partial rows
// walked 16-way in parallel (one thread per column adding ~200 dependent loads took 46 us per step)
__global__ __launch_bounds__(1024) void sum_partials(const float *part, int nblk, int N, float *out) {
  __shared__ float sh[16][65];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float s = 0.f;
  if (n < N)
    for (int b = g; b < nblk; b += 16) s += part[(int64_t)b * N + n];
  sh[g][c] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][c];
    out[n] = t;
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float *a, const float *b, const float *c, int64_t n4, int64_t n, float *out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 x = reinterpret_cast<const float4 *>(a)[i];
    const float4 y = reinterpret_cast<const float4 *>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    if (c) {
      const float4 z = reinterpret_cast<const float4 *>(c)[i];
      x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w;
    }
    reinterpret_cast<float4 *>(out)[i] = x;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    out[i] = a[i] + b[i] + (c ? c[i] : 0.f);
  }
}

// x *= *scalar (device scalar; used for the upstream gradient of a scalar loss)
__global__ __launch_bounds__(256) void scale_kernel(float *x, const float *scalar, int64_t n) {
  const float s = scalar[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}

__global__ __launch_bounds__(256) void cast_kernel(const void *src, int sd, void *dst, int dd, int64_t n) {
  // 4 elements per thread where alignment allows (buffers from torch are >= 256-byte aligned)
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v;
    if (sd == EVP_F32) v = reinterpret_cast<const float4 *>(src)[i];
    else {
      const uint2 u = reinterpret_cast<const uint2 *>(src)[i];
      v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                      __uint_as_float(u.y & 0xFFFF0000u));
    }
    if (dd == EVP_F32) reinterpret_cast<float4 *>(dst)[i] = v;
    else {
      uint2 u;
      u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
      u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
      reinterpret_cast<uint2 *>(dst)[i] = u;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    st_any(dst, dd, i, ld_any(src, sd, i));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *src, T *dst, int64_t rows, int64_t cols) {
  __shared__ T tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < cols && r0 + tx < rows) dst[(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

static inline int ew_grid(int64_t n_items) {
  int64_t g = (n_items + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int evp_mask_from_noise(const float *noise, int B, int L, int len_keep, int64_t *ids_keep, float *mask,
                                   int64_t *ids_restore, void *stream) {
  EVP_CHECK_ARG(noise && (ids_keep || len_keep == 0) && mask && ids_restore, EVP_EINVAL, "evp_mask_from_noise: null pointer");
  EVP_CHECK_ARG(B > 0 && L > 0 && L <= 4096 && len_keep >= 0 && len_keep <= L, EVP_ESHAPE,
                "evp_mask_from_noise: need 0<L<=4096, 0<=len_keep<=L (B=%d L=%d keep=%d)", B, L, len_keep);
  hipLaunchKernelGGL(mask_kernel, dim3(B), dim3(256), (size_t)L * sizeof(float), (hipStream_t)stream, noise, L, len_keep,
                     ids_keep, mask, ids_restore);
  EVP_CHECK_LAUNCH("evp_mask_from_noise");
  return EVP_OK;
}

extern "C" int evp_density_noise(const float *x, int B, int C, int H, int W, int patch, float sign, float *noise, void *stream) {
  EVP_CHECK_ARG(x && noise, EVP_EINVAL, "evp_density_noise: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_density_noise: bad shape");
  EVP_CHECK_ARG(patch * patch * 4 <= 48 * 1024, EVP_ESHAPE, "evp_density_noise: patch %d too large", patch);
  hipLaunchKernelGGL(density_kernel, dim3(B * (H / patch) * (W / patch)), dim3(256), patch * patch * sizeof(float), (hipStream_t)stream, x, C, H, W, patch, sign, noise);
  EVP_CHECK_LAUNCH("evp_density_noise");
  return EVP_OK;
}

extern "C" int evp_patchify(const float *x, const int64_t *ids_keep, int B, int C, int H, int W, int patch, int n_keep,
                            void *cols, int dtype, void *stream) {
  EVP_CHECK_ARG(x && cols, EVP_EINVAL, "evp_patchify: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && patch % 4 == 0 && H % patch == 0 && W % patch == 0 && W % 4 == 0, EVP_ESHAPE,
                "evp_patchify: need patch%%4==0 and H,W multiples of patch");
  const int L = (H / patch) * (W / patch);
  EVP_CHECK_ARG(n_keep > 0 && n_keep <= L && (ids_keep || n_keep == L), EVP_ESHAPE, "evp_patchify: bad n_keep %d (L=%d)", n_keep, L);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EVP_F32) hipLaunchKernelGGL(patchify_kernel<float>, dim3(B * n_keep), dim3(256), 0, s, x, ids_keep, C, H, W, patch, n_keep, (float *)cols);
  else hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(B * n_keep), dim3(256), 0, s, x, ids_keep, C, H, W, patch, n_keep, (bf16_t *)cols);
  EVP_CHECK_LAUNCH("evp_patchify");
  return EVP_OK;
}

extern "C" int evp_add_rows_gather_f32(const float *x, const float *table, const int64_t *ids, int B, int n, int L, int D, float *out,
                                       void *stream) {
  EVP_CHECK_ARG(x && table && out, EVP_EINVAL, "evp_add_rows_gather_f32: null pointer");
  EVP_CHECK_ARG(B > 0 && n > 0 && L > 0 && D % 4 == 0 && (ids || n == L), EVP_ESHAPE, "evp_add_rows_gather_f32: bad shape");
  hipLaunchKernelGGL(add_rows_gather_kernel, dim3(B * n), dim3(D / 4 < 256 ? ((D / 4 + 63) / 64) * 64 : 256), 0, (hipStream_t)stream, x, table,
                     ids, n, L, D, out);
  EVP_CHECK_LAUNCH("evp_add_rows_gather_f32");
  return EVP_OK;
}

extern "C" int evp_unshuffle_fwd(const float *emb, const float *mask_token, const float *pos, const int64_t *ids_restore, int B,
                                 int n_keep, int L, int D, float *out, void *stream) {
  EVP_CHECK_ARG(emb && mask_token && pos && ids_restore && out, EVP_EINVAL, "evp_unshuffle_fwd: null pointer");
  EVP_CHECK_ARG(B > 0 && n_keep > 0 && n_keep <= L && D % 4 == 0, EVP_ESHAPE, "evp_unshuffle_fwd: bad shape");
  hipLaunchKernelGGL(unshuffle_fwd_kernel, dim3(B * L), dim3(D / 4 < 256 ? ((D / 4 + 63) / 64) * 64 : 256), 0, (hipStream_t)stream,
                     emb, mask_token, pos, ids_restore, n_keep, L, D, out);
  EVP_CHECK_LAUNCH("evp_unshuffle_fwd");
  return EVP_OK;
}

extern "C" int evp_unshuffle_bwd(const float *g, const int64_t *ids_restore, int B, int n_keep, int L, int D, float *demb,
                                 float *dmask_token, float *workspace, void *stream) {
  EVP_CHECK_ARG(g && ids_restore && demb, EVP_EINVAL, "evp_unshuffle_bwd: null pointer");
  EVP_CHECK_ARG(B > 0 && n_keep > 0 && n_keep <= L && D % 4 == 0, EVP_ESHAPE, "evp_unshuffle_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(unshuffle_bwd_scatter, dim3(B * L), dim3(D / 4 < 256 ? ((D / 4 + 63) / 64) * 64 : 256), 0, s, g, ids_restore,
                     n_keep, L, D, demb);
  EVP_CHECK_LAUNCH("evp_unshuffle_bwd");
  if (dmask_token) {
    EVP_CHECK_ARG(workspace, EVP_EINVAL, "evp_unshuffle_bwd: workspace needed for dmask_token");
    const int64_t rows = (int64_t)B * L;
    const int nb = (int)((rows + UM_ROWS - 1) / UM_ROWS);
    hipLaunchKernelGGL(unshuffle_bwd_masktok, dim3((D + 255) / 256, nb), dim3(256), 0, s, g, ids_restore, rows, n_keep, D, workspace);
    EVP_CHECK_LAUNCH("evp_unshuffle_bwd(mask_token)");
    hipLaunchKernelGGL(sum_partials, dim3((D + 63) / 64), dim3(1024), 0, s, workspace, nb, D, dmask_token);
    EVP_CHECK_LAUNCH("evp_unshuffle_bwd(finalize)");
  }
  return EVP_OK;
}

extern "C" int evp_add_f32(const float *a, const float *b, const float *c, int64_t n, float *out, void *stream) {
  EVP_CHECK_ARG(a && b && out && n > 0, EVP_EINVAL, "evp_add_f32: bad argument");
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, a, b, c, n >> 2, n, out);
  EVP_CHECK_LAUNCH("evp_add_f32");
  return EVP_OK;
}

extern "C" int evp_scale_f32(float *x, const float *scalar, int64_t n, void *stream) {
  EVP_CHECK_ARG(x && scalar && n > 0, EVP_EINVAL, "evp_scale_f32: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, n);
  EVP_CHECK_LAUNCH("evp_scale_f32");
  return EVP_OK;
}

extern "C" int evp_cast(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, void *stream) {
  EVP_CHECK_ARG(src && dst && n > 0, EVP_EINVAL, "evp_cast: bad argument");
  hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, dst, dst_dtype, n);
  EVP_CHECK_LAUNCH("evp_cast");
  return EVP_OK;
}

extern "C" int evp_transpose(const void *src, void *dst, int dtype, int64_t rows, int64_t cols, void *stream) {
  EVP_CHECK_ARG(src && dst && rows > 0 && cols > 0, EVP_EINVAL, "evp_transpose: bad argument");
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  if (dtype == EVP_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, (float *)dst, rows, cols);
  else hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)src, (bf16_t *)dst, rows, cols);
  EVP_CHECK_LAUNCH("evp_transpose");
  return EVP_OK;
}

// ------------------------------------------------------------------------------------------------ token mean pool
// Fine-tune classification head (model/finetune_cls/ft_cls_hub_model.py:136): emb_h.mean(dim=1) of f32 tokens
// [B, N, D] -> [B, D], and its backward dx[b,n,:] = g[b,:] / N. One thread per (b, 4 columns); the N-loop reads
// consecutive rows, so a wave reads 1 KiB-contiguous row segments. HBM-bound, 4 B per input element.
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ out, int N, int D4) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= D4) return;
  const float4 *p = x + (int64_t)b * N * D4 + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < N; ++n) {
    const float4 v = p[(int64_t)n * D4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = 1.0f / (float)N;
  out[(int64_t)b * D4 + c] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
}
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float4 *__restrict__ g, float4 *__restrict__ dx, int64_t total, int N, int D4) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % D4);
  const int64_t b = e / ((int64_t)N * D4);
  const float4 v = g[b * D4 + c];
  const float inv = 1.0f / (float)N;
  dx[e] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

extern "C" int evp_token_mean_fwd(const float *x, int B, int N, int D, float *out, void *stream) {
  EVP_CHECK_ARG(x && out, EVP_EINVAL, "evp_token_mean_fwd: null pointer");
  EVP_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && D > 0 && D % 4 == 0, EVP_ESHAPE, "evp_token_mean_fwd: B=%d N=%d D=%d (D%%4==0)", B, N, D);
  hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((unsigned)((D / 4 + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)x, (float4 *)out, N, D / 4);
  EVP_CHECK_LAUNCH("evp_token_mean_fwd");
  return EVP_OK;
}
extern "C" int evp_token_mean_bwd(const float *g, int B, int N, int D, float *dx, void *stream) {
  EVP_CHECK_ARG(g && dx, EVP_EINVAL, "evp_token_mean_bwd: null pointer");
  EVP_CHECK_ARG(B > 0 && N > 0 && D > 0 && D % 4 == 0, EVP_ESHAPE, "evp_token_mean_bwd: B=%d N=%d D=%d (D%%4==0)", B, N, D);
  const int64_t total = (int64_t)B * N * (D / 4);
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)g,
                     (float4 *)dx, total, N, D / 4);
  EVP_CHECK_LAUNCH("evp_token_mean_bwd");
  return EVP_OK;
}

// ------------------------------------------------------------------------------------------------ slice sum
// out[i] (+)= sum_s ws[s * numel + i]: the reduction of split-K weight-gradient partials (each K slice of a long-K
// problem is computed as its own problem of the grouped launch into its own slice of a workspace).
__global__ __launch_bounds__(256) void sum_slices_kernel(const float4 *__restrict__ ws, float4 *__restrict__ out, int n_slices, int64_t n4,
                                                         int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = accumulate ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < n_slices; ++k) {
    const float4 v = ws[(int64_t)k * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[i] = s;
}
extern "C" int evp_sum_slices_f32(const float *ws, float *out, int n_slices, int64_t numel, int accumulate, void *stream) {
  EVP_CHECK_ARG(ws && out, EVP_EINVAL, "evp_sum_slices_f32: null pointer");
  EVP_CHECK_ARG(n_slices > 0 && numel > 0 && numel % 4 == 0, EVP_ESHAPE, "evp_sum_slices_f32: n_slices=%d numel=%lld (numel%%4==0)", n_slices,
                (long long)numel);
  const int64_t n4 = numel / 4;
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)ws, (float4 *)out,
                     n_slices, n4, accumulate);
  EVP_CHECK_LAUNCH("evp_sum_slices_f32");
  return EVP_OK;
}
