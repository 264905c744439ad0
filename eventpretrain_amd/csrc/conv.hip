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

// Forward:  y[b,y,x,c]   = bias[c] + sum_{i,j} w[c,i,j] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c]
// Backward: din[b,y,x,c] = keep(y,x) * sum_{i,j} w[c,i,j] * dout[b,y-i+2,x-j+2,c]     (the flipped kernel on dout)
// One kernel for both: a wave walks an image row left to right with a 5x5 register window per channel (lane = 2
// adjacent channels, 64 lanes = 128 channels = one contiguous 256-byte / 512-byte row segment), so a position costs
// 5 loads + 1 store instead of 25 + 1. BWD: window elements are not masked, the result is; the taps are read flipped.
template <typename T> __device__ __forceinline__ void ld2(const T *p, float &a, float &b);
template <> __device__ __forceinline__ void ld2<float>(const float *p, float &a, float &b) {
  const float2 v = *reinterpret_cast<const float2 *>(p);
  a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t *p, float &a, float &b) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void st2(T *p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t *p, float a, float b) {
  *reinterpret_cast<uint32_t *>(p) = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
constexpr int DWF_RPW = 2;                    // rows per wave
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const T *src, const float *mask, const float *w, const float *bias, int B, int H,
                                                          int W, int C, int ms, int mgw, int mL, T *dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  if (c >= C) return;
  const int nrows = B * H;
  float k0[5][5], k1[5][5];                   // taps of the two channels (flipped for the backward)
#pragma unroll
  for (int ki = 0; ki < 5; ++ki)
#pragma unroll
    for (int kj = 0; kj < 5; ++kj) {
      const int t = BWD ? (4 - ki) * 5 + (4 - kj) : ki * 5 + kj;
      k0[ki][kj] = w[c * 25 + t];
      k1[ki][kj] = w[(c + 1) * 25 + t];
    }
  const float b0 = BWD ? 0.f : bias[c], b1 = BWD ? 0.f : bias[c + 1];
  const int r0 = (blockIdx.y * 4 + wave) * DWF_RPW;
  for (int r = r0; r < r0 + DWF_RPW && r < nrows; ++r) {
    const int b = r / H, y = r - b * H;
    float w0[5][5], w1[5][5];
    auto load_col = [&](int xx, int kj) {
#pragma unroll
      for (int ki = 0; ki < 5; ++ki) {
        const int yy = y + ki - 2;
        float a_ = 0.f, b_ = 0.f;
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
          const float k = BWD ? 1.0f : keep_at(mask, b, yy, xx, ms, mgw, mL);
          if (k != 0.f) {
            ld2<T>(src + (((int64_t)b * H + yy) * W + xx) * C + c, a_, b_);
            a_ *= k; b_ *= k;
          }
        }
        w0[ki][kj] = a_; w1[ki][kj] = b_;
      }
    };
    load_col(-2, 0); load_col(-1, 1); load_col(0, 2); load_col(1, 3);
    for (int x = 0; x < W; ++x) {
      load_col(x + 2, 4);
      float s0 = b0, s1 = b1;
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
          s0 += k0[ki][kj] * w0[ki][kj];
          s1 += k1[ki][kj] * w1[ki][kj];
        }
      if (BWD) {
        const float k = keep_at(mask, b, y, x, ms, mgw, mL);
        s0 *= k; s1 *= k;
      }
      st2<T>(dst + ((int64_t)r * W + x) * C + c, s0, s1);
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
    }
  }
}

// dw[c,i,j] = sum_{b,y,x} dout[b,y,x,c] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c];  dbias[c] = sum dout
// A wave walks image rows (b, y) left to right with a 5x5 register window of the (masked) input per channel: one new
// column (5 loads) + one dout load per position instead of 25 + 1, and the keep mask is looked up once per loaded
// element. Lane = 2 adjacent channels (one 4-byte / 8-byte load: 64 lanes cover 128 channels = 256 contiguous bytes).
// block = 4 waves x DW_RPW rows each; partials part[slab][26][C] (25 taps + bias), slab = blockIdx.y.
constexpr int DW_RPW = 4;                     // rows per wave
constexpr int DW_ROWS = 4 * DW_RPW;           // rows per block (= slab)
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T *dout, const T *in, const float *mask, int B, int H, int W, int C,
                                                                int ms, int mgw, int mL, float *part) {
  __shared__ float sh[4][26][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  const int nrows = B * H;
  float acc0[26], acc1[26];
#pragma unroll
  for (int k = 0; k < 26; ++k) acc0[k] = acc1[k] = 0.f;
  if (c < C) {
    const int r0 = blockIdx.y * DW_ROWS + wave * DW_RPW;
    for (int r = r0; r < r0 + DW_RPW && r < nrows; ++r) {
      const int b = r / H, y = r - b * H;
      float w0[5][5], w1[5][5];          // window: w[ki][kj] = keep * in[b, y+ki-2, x+kj-2, c / c+1]
      auto load_col = [&](int xx, float (&o0)[5], float (&o1)[5]) {
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) {
          const int yy = y + ki - 2;
          o0[ki] = o1[ki] = 0.f;
          if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const float k = keep_at(mask, b, yy, xx, ms, mgw, mL);
            if (k != 0.f) {
              float a_, b_;
              ld2<T>(in + (((int64_t)b * H + yy) * W + xx) * C + c, a_, b_);
              o0[ki] = k * a_; o1[ki] = k * b_;
            }
          }
        }
      };
      float c0[5], c1[5];
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {     // columns x-2 .. x+1 for x = 0
        load_col(kj - 2, c0, c1);
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) { w0[ki][kj] = c0[ki]; w1[ki][kj] = c1[ki]; }
      }
      for (int x = 0; x < W; ++x) {
        load_col(x + 2, c0, c1);
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) { w0[ki][4] = c0[ki]; w1[ki][4] = c1[ki]; }
        float g0, g1;
        ld2<T>(dout + ((int64_t)r * W + x) * C + c, g0, g1);
        acc0[25] += g0; acc1[25] += g1;
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 5; ++kj) {
            acc0[ki * 5 + kj] += g0 * w0[ki][kj];
            acc1[ki * 5 + kj] += g1 * w1[ki][kj];
          }
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 26; ++k) { sh[wave][k][lane * 2] = acc0[k]; sh[wave][k][lane * 2 + 1] = acc1[k]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 26 * 128; e += 256) {
    const int k = e / 128, cc = e % 128;
    if (blockIdx.x * 128 + cc < C)
      part[((int64_t)blockIdx.y * 26 + k) * C + blockIdx.x * 128 + cc] = sh[0][k][cc] + sh[1][k][cc] + sh[2][k][cc] + sh[3][k][cc];
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
    hipError_t e = evp_zero_async(dx, sizeof(float) * (size_t)B * H * W * C, s);
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
  (void)n;
  const dim3 rg((unsigned)((C + 127) / 128), (unsigned)(((int64_t)B * H + 4 * DWF_RPW - 1) / (4 * DWF_RPW)));
  if (dtype == EVP_F32) hipLaunchKernelGGL((dwconv_rows_kernel<float, false>), rg, dim3(256), 0, s, (const float *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (float *)out);
  else hipLaunchKernelGGL((dwconv_rows_kernel<bf16_t, false>), rg, dim3(256), 0, s, (const bf16_t *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)out);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_fwd");
  return EVP_OK;
}

extern "C" int evp_dwconv5x5_bwd_nslab(int B, int H, int W) { (void)W; return (int)(((int64_t)B * H + DW_ROWS - 1) / DW_ROWS); }

extern "C" int evp_dwconv5x5_bwd(const void *dout, const void *in, int dtype, const float *mask, int mask_scale, const float *w, int B, int H,
                                 int W, int C, void *din, float *dw, float *dbias, float *workspace, void *stream) {
  EVP_CHECK_ARG(dout && in && w && din && dw && dbias && workspace, EVP_EINVAL, "evp_dwconv5x5_bwd: null pointer");
  DW_COMMON_CHECK("evp_dwconv5x5_bwd")
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  const int nslab = evp_dwconv5x5_bwd_nslab(B, H, W);
  dim3 wg((C + 127) / 128, nslab);
  (void)n;
  const dim3 rg((unsigned)((C + 127) / 128), (unsigned)(((int64_t)B * H + 4 * DWF_RPW - 1) / (4 * DWF_RPW)));
  if (dtype == EVP_F32) {
    hipLaunchKernelGGL((dwconv_rows_kernel<float, true>), rg, dim3(256), 0, s, (const float *)dout, mask, w, (const float *)nullptr, B, H, W, C, mask_scale, mgw, mL, (float *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<float>, wg, dim3(256), 0, s, (const float *)dout, (const float *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  } else {
    hipLaunchKernelGGL((dwconv_rows_kernel<bf16_t, true>), rg, dim3(256), 0, s, (const bf16_t *)dout, mask, w, (const float *)nullptr, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<bf16_t>, wg, dim3(256), 0, s, (const bf16_t *)dout, (const bf16_t *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  }
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd");
  hipLaunchKernelGGL(dwconv_bwd_weight_finalize, dim3((26 * C + 255) / 256), dim3(256), 0, s, workspace, nslab, C, dw, dbias);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd(finalize)");
  return EVP_OK;
}
