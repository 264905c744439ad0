// Internal helpers shared by the gfx950 kernels of libevtpretrain.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/evtpretrain.h"

#define EVP_WAVE 64

typedef uint16_t bf16_t;  // raw storage

void evp_set_error(const char *fmt, ...);

#define EVP_CHECK_ARG(cond, code, ...)      \
  do {                                      \
    if (!(cond)) {                          \
      evp_set_error(__VA_ARGS__);           \
      return (code);                        \
    }                                       \
  } while (0)

#define EVP_CHECK_LAUNCH(name)                                                       \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      evp_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));         \
      return EVP_ELAUNCH;                                                            \
    }                                                                                \
  } while (0)

// ---- bf16 <-> f32 (round to nearest even; NaN stays NaN via the compiler's own cast) ------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;  // hipcc lowers to v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float ld_any(const void *p, int dtype, int64_t i) {
  return dtype == EVP_BF16 ? bf16_to_f32(((const bf16_t *)p)[i]) : ((const float *)p)[i];
}
__device__ __forceinline__ void st_any(void *p, int dtype, int64_t i, float v) {
  if (dtype == EVP_BF16) ((bf16_t *)p)[i] = f32_to_bf16(v);
  else ((float *)p)[i] = v;
}

// ---- wave / block reductions (wave = 64 lanes) --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// exact erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- zero fill as a KERNEL ------------------------------------------------------------------------------------------------
// Accumulate-by-atomics outputs (split-K partial sums, column sums, the relative-position table gradient, scatter targets) are
// cleared by a kernel launch, not by hipMemsetAsync: inside a captured HIP graph the memset NODES of this ROCm release were
// observed not to take effect reliably on replay (the second replay of a Swin f32 step added its atomics onto whatever the
// graph's memory pool had left there: non-finite gradients in exactly the outputs cleared by memset nodes, first replay fine).
// A kernel node has ordinary stream-order semantics. Dwords; pitch/width in bytes, both multiples of 4.
namespace {
__global__ __launch_bounds__(256) void evp_zero_kernel(uint32_t *p, int64_t pitch_dw, int64_t width_dw, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / width_dw, c = e - r * width_dw;
  p[r * pitch_dw + c] = 0u;
}
// 16-byte stores, grid-stride (large contiguous buffers: a ConvViT patch-embed gradient is 205 MB)
__global__ __launch_bounds__(256) void evp_zero16_kernel(uint4 *p, int64_t n16) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n16; e += (int64_t)gridDim.x * 256) p[e] = make_uint4(0u, 0u, 0u, 0u);
}
inline hipError_t evp_zero2d_async(void *p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t s) {
  if (width_bytes == 0 || rows == 0) return hipSuccess;
  if ((pitch_bytes | width_bytes | (size_t)(uintptr_t)p) & 3) return hipErrorInvalidValue;
  if ((rows == 1 || pitch_bytes == width_bytes) && (((size_t)(uintptr_t)p | (width_bytes * rows)) & 15) == 0) {
    const int64_t n16 = (int64_t)(width_bytes * rows / 16);
    int64_t g = (n16 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(evp_zero16_kernel, dim3((unsigned)g), dim3(256), 0, s, (uint4 *)p, n16);
    return hipGetLastError();
  }
  const int64_t total = (int64_t)(width_bytes / 4) * (int64_t)rows;
  hipLaunchKernelGGL(evp_zero_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (uint32_t *)p, (int64_t)(pitch_bytes / 4),
                     (int64_t)(width_bytes / 4), total);
  return hipGetLastError();
}
inline hipError_t evp_zero_async(void *p, size_t bytes, hipStream_t s) { return evp_zero2d_async(p, bytes, bytes, 1, s); }
}  // namespace
