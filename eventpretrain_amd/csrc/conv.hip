// ConvViT stage-1/2 kernels on channels-last token maps ([B,H,W,C], C contiguous), gfx950. All HBM-bound.
//   * patch gather / scatter for the non-overlapping strided convolutions (PatchEmbed k=s=2/4 and the multi-scale
//     fusion convs k=s=4/2): the conv itself is a GEMM on the gathered patch matrix (csrc/gemm.hip).
//   * depthwise 5x5 convolution (pad 2, groups=C) with the keep-mask multiply of ConvBlock fused into the input read.
// Replaces model/sub_module/conv_block.py:41-51 (ConvBlock.attn + mask) and the Conv2d(k=s) layers of
// model/backbone/convvit.py:20-25,48-49.
#include "evp_common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }

// cols[(b,j), c*p*p + py*p + px] = x[b, gy*p+py, gx*p+px, c]; token = ids_keep[b,j] (or j) -> (gy, gx)
// one thread per (row, c): p*p strided loads (coalesced over c), p*p contiguous stores
template <typename TO>
__global__ __launch_bounds__(256) void patchify_nhwc_kernel(const float *x, const int64_t *ids_keep, int H, int W, int C, int p,
                                                            int n_keep, int64_t rows, TO *cols) {
  const int gw = W / p, L = (H / p) * gw, pp = p * p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int64_t b = row / n_keep;
    const int tok = ids_keep ? (int)ids_keep[row] : (int)(row % L);
    const int gy = tok / gw, gx = tok % gw;
    TO *o = cols + row * ((int64_t)C * pp) + (int64_t)c * pp;
    for (int py = 0; py < p; ++py)
      for (int px = 0; px < p; ++px) {
        const float v = x[(((int64_t)b * H + gy * p + py) * W + gx * p + px) * C + c];
        ElemIO<TO>::st(o + py * p + px, v);
      }
  }
}
// dx[b, gy*p+py, gx*p+px, c] (+)= dcols[(b,j), c*p*p + py*p + px]; dx must be zero-initialised when ids_keep != NULL
template <typename TI>
__global__ __launch_bounds__(256) void unpatchify_nhwc_kernel(const TI *dcols, const int64_t *ids_keep, int H, int W, int C, int p,
                                                              int n_keep, int64_t rows, int accumulate, float *dx) {
  const int gw = W / p, L = (H / p) * gw, pp = p * p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int64_t b = row / n_keep;
    const int tok = ids_keep ? (int)ids_keep[row] : (int)(row % L);
    const int gy = tok / gw, gx = tok % gw;
    const TI *s = dcols + row * ((int64_t)C * pp) + (int64_t)c * pp;
    for (int py = 0; py < p; ++py)
      for (int px = 0; px < p; ++px) {
        float *d = dx + (((int64_t)b * H + gy * p + py) * W + gx * p + px) * C + c;
        const float v = ldf<TI>(s + py * p + px);
        *d = accumulate ? *d + v : v;
      }
  }
}

// keep factor of position (y,x): 1 - mask[b, (y/s)*gw + x/s]  (convvit.py:129-130,142-143), or 1 without a mask
__device__ __forceinline__ float keep_at(const float *mask, int b, int y, int x, int s, int gw, int L) {
  return mask ? 1.0f - mask[(int64_t)b * L + (y / s) * gw + x / s] : 1.0f;
}

// y[b,y,x,c] = bias[c] + sum_{i,j} w[c,i,j] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c]
// thread = 4 channels of one position (float4 / 8-byte loads); weights for its channels in registers
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T *in, const float *mask, const float *w, const float *bias, int B, int H,
                                                         int W, int C, int ms, int mgw, int mL, T *out) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[4] = {bias[c], bias[c + 1], bias[c + 2], bias[c + 3]};
#pragma unroll
    for (int ki = 0; ki < 5; ++ki) {
      const int yy = y + ki - 2;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kj = 0; kj < 5; ++kj) {
        const int xx = x + kj - 2;
        if (xx < 0 || xx >= W) continue;
        const float k = keep_at(mask, b, yy, xx, ms, mgw, mL);
        if (k == 0.f) continue;
        const T *p = in + (((int64_t)b * H + yy) * W + xx) * C + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w[(c + e) * 25 + ki * 5 + kj] * (k * ldf<T>(p + e));
      }
    }
    T *o = out + (((int64_t)b * H + y) * W + x) * C + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) ElemIO<T>::st(o + e, acc[e]);
  }
}

// din[b,y,x,c] = keep(y,x) * sum_{i,j} w[c,i,j] * dout[b,y-i+2,x-j+2,c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T *dout, const float *mask, const float *w, int B, int H, int W, int C,
                                                              int ms, int mgw, int mL, T *din) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float k = keep_at(mask, b, y, x, ms, mgw, mL);
    if (k != 0.f) {
#pragma unroll
      for (int ki = 0; ki < 5; ++ki) {
        const int yy = y - ki + 2;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
          const int xx = x - kj + 2;
          if (xx < 0 || xx >= W) continue;
          const T *p = dout + (((int64_t)b * H + yy) * W + xx) * C + c;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w[(c + e) * 25 + ki * 5 + kj] * ldf<T>(p + e);
        }
      }
    }
    T *o = din + (((int64_t)b * H + y) * W + x) * C + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) ElemIO<T>::st(o + e, k * acc[e]);
  }
}

// dw[c,i,j] = sum_{b,y,x} dout[b,y,x,c] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c];  dbias[c] = sum dout
// block = 64 channels x 4 position lanes over a slab of positions; partials part[slab][26][C] (25 taps + bias)
constexpr int DW_SLAB = 512;   // positions per block
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T *dout, const T *in, const float *mask, int B, int H, int W, int C,
                                                                int ms, int mgw, int mL, float *part) {
  __shared__ float sh[4][26][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t npos = (int64_t)B * H * W;
  const int64_t p0 = (int64_t)blockIdx.y * DW_SLAB, p1 = p0 + DW_SLAB < npos ? p0 + DW_SLAB : npos;
  float acc[26];
#pragma unroll
  for (int k = 0; k < 26; ++k) acc[k] = 0.f;
  if (c < C) {
    for (int64_t pos = p0 + pl; pos < p1; pos += 4) {
      const int x = (int)(pos % W);
      const int y = (int)((pos / W) % H);
      const int b = (int)(pos / ((int64_t)W * H));
      const float g = ldf<T>(dout + pos * C + c);
      acc[25] += g;
#pragma unroll
      for (int ki = 0; ki < 5; ++ki) {
        const int yy = y + ki - 2;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
          const int xx = x + kj - 2;
          if (xx < 0 || xx >= W) continue;
          const float k = keep_at(mask, b, yy, xx, ms, mgw, mL);
          if (k != 0.f) acc[ki * 5 + kj] += g * k * ldf<T>(in + (((int64_t)b * H + yy) * W + xx) * C + c);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 26; ++k) sh[pl][k][cl] = acc[k];
  __syncthreads();
  for (int e = threadIdx.x; e < 26 * 64; e += 256) {
    const int k = e / 64, cc = e % 64;
    if (blockIdx.x * 64 + cc < C)
      part[((int64_t)blockIdx.y * 26 + k) * C + blockIdx.x * 64 + cc] = sh[0][k][cc] + sh[1][k][cc] + sh[2][k][cc] + sh[3][k][cc];
  }
}
// dw[c*25 + k] = sum_slab part[slab][k][c] (k < 25), dbias[c] = sum_slab part[slab][25][c]
__global__ __launch_bounds__(256) void dwconv_bwd_weight_finalize(const float *part, int nslab, int C, float *dw, float *dbias) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 26 * C) return;
  const int k = e / C, c = e % C;
  float s = 0.f;
  for (int i = 0; i < nslab; ++i) s += part[((int64_t)i * 26 + k) * C + c];
  if (k < 25) dw[c * 25 + k] = s;
  else dbias[c] = s;
}

static inline int ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int evp_patchify_nhwc(const float *x, const int64_t *ids_keep, int B, int H, int W, int C, int patch, int n_keep, void *cols,
                                 int dtype, void *stream) {
  EVP_CHECK_ARG(x && cols, EVP_EINVAL, "evp_patchify_nhwc: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_patchify_nhwc: bad shape");
  const int L = (H / patch) * (W / patch);
  EVP_CHECK_ARG(n_keep > 0 && n_keep <= L && (ids_keep || n_keep == L), EVP_ESHAPE, "evp_patchify_nhwc: bad n_keep");
  const int64_t rows = (int64_t)B * n_keep;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EVP_F32) hipLaunchKernelGGL(patchify_nhwc_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, s, x, ids_keep, H, W, C, patch, n_keep, rows, (float *)cols);
  else hipLaunchKernelGGL(patchify_nhwc_kernel<bf16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, s, x, ids_keep, H, W, C, patch, n_keep, rows, (bf16_t *)cols);
  EVP_CHECK_LAUNCH("evp_patchify_nhwc");
  return EVP_OK;
}

extern "C" int evp_unpatchify_nhwc(const void *dcols, int dtype, const int64_t *ids_keep, int B, int H, int W, int C, int patch, int n_keep,
                                   int accumulate, float *dx, void *stream) {
  EVP_CHECK_ARG(dcols && dx, EVP_EINVAL, "evp_unpatchify_nhwc: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_unpatchify_nhwc: bad shape");
  const int L = (H / patch) * (W / patch);
  EVP_CHECK_ARG(n_keep > 0 && n_keep <= L && (ids_keep || n_keep == L), EVP_ESHAPE, "evp_unpatchify_nhwc: bad n_keep");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && n_keep < L) {
    hipError_t e = hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * H * W * C, s);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_unpatchify_nhwc: memset failed: %s", hipGetErrorString(e));
  }
  const int64_t rows = (int64_t)B * n_keep;
  if (dtype == EVP_F32) hipLaunchKernelGGL(unpatchify_nhwc_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, s, (const float *)dcols, ids_keep, H, W, C, patch, n_keep, rows, accumulate, dx);
  else hipLaunchKernelGGL(unpatchify_nhwc_kernel<bf16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, s, (const bf16_t *)dcols, ids_keep, H, W, C, patch, n_keep, rows, accumulate, dx);
  EVP_CHECK_LAUNCH("evp_unpatchify_nhwc");
  return EVP_OK;
}

#define DW_COMMON_CHECK(name)                                                                                              \
  EVP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVP_ESHAPE, name ": need C %% 4 == 0");                     \
  EVP_CHECK_ARG(!mask || (mask_scale > 0 && H % mask_scale == 0 && W % mask_scale == 0), EVP_ESHAPE, name ": bad mask scale"); \
  const int mgw = mask ? W / mask_scale : 1, mL = mask ? (H / mask_scale) * (W / mask_scale) : 1;

extern "C" int evp_dwconv5x5_fwd(const void *in, int dtype, const float *mask, int mask_scale, const float *w, const float *bias, int B,
                                 int H, int W, int C, void *out, void *stream) {
  EVP_CHECK_ARG(in && w && bias && out, EVP_EINVAL, "evp_dwconv5x5_fwd: null pointer");
  DW_COMMON_CHECK("evp_dwconv5x5_fwd")
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EVP_F32) hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (float *)out);
  else hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16_t *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)out);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_fwd");
  return EVP_OK;
}

extern "C" int evp_dwconv5x5_bwd_nslab(int B, int H, int W) { return (int)(((int64_t)B * H * W + DW_SLAB - 1) / DW_SLAB); }

extern "C" int evp_dwconv5x5_bwd(const void *dout, const void *in, int dtype, const float *mask, int mask_scale, const float *w, int B, int H,
                                 int W, int C, void *din, float *dw, float *dbias, float *workspace, void *stream) {
  EVP_CHECK_ARG(dout && in && w && din && dw && dbias && workspace, EVP_EINVAL, "evp_dwconv5x5_bwd: null pointer");
  DW_COMMON_CHECK("evp_dwconv5x5_bwd")
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  const int nslab = evp_dwconv5x5_bwd_nslab(B, H, W);
  dim3 wg((C + 63) / 64, nslab);
  if (dtype == EVP_F32) {
    hipLaunchKernelGGL(dwconv_bwd_data_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float *)dout, mask, w, B, H, W, C, mask_scale, mgw, mL, (float *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<float>, wg, dim3(256), 0, s, (const float *)dout, (const float *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  } else {
    hipLaunchKernelGGL(dwconv_bwd_data_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16_t *)dout, mask, w, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<bf16_t>, wg, dim3(256), 0, s, (const bf16_t *)dout, (const bf16_t *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  }
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd");
  hipLaunchKernelGGL(dwconv_bwd_weight_finalize, dim3((26 * C + 255) / 256), dim3(256), 0, s, workspace, nslab, C, dw, dbias);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd(finalize)");
  return EVP_OK;
}
