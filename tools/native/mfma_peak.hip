// Calibration, not part of the product: the MFMA rate this GPU sustains when nothing but v_mfma_f32_16x16x32_bf16 runs
// (16 independent accumulators per wave, W waves per CU), to put the 2.5 PFLOP/s nominal peak used by bench.py's roofline
// next to a measured ceiling.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)((threadIdx.x >> 3) & 7); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void mfma32_loop(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)((threadIdx.x >> 3) & 7); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  float *out;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu : {1, 2, 4}) {
    for (int iters : {2000, 20000, 200000}) {
      hipLaunchKernelGGL(mfma_loop, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 100);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 2.0 * 16 * 16 * 32 * 16.0 * iters * 4.0 * cus * wg_per_cu;
      printf("CUs %d, %d waves per SIMD, %6d x 16 MFMAs per wave: %8.3f ms  %7.1f TFLOP/s\n", cus, wg_per_cu, iters, ms, flops / ms / 1e9);
    }
  }
  for (int wg_per_cu : {1, 2}) {
    const int iters = 100000;
    hipLaunchKernelGGL(mfma32_loop, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma32_loop, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * 4.0 * cus * wg_per_cu;
    printf("32x32x16: CUs %d, %d waves per SIMD, %6d x 4 MFMAs per wave: %8.3f ms  %7.1f TFLOP/s\n", cus, wg_per_cu, iters, ms, flops / ms / 1e9);
  }
  return 0;
}
