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
// 32x32x16 with NACC accumulators (16 of them = 256 registers, as in the G4 GEMM body, which puts them in AGPRs) and NOPS
// distinct operand register sets used in the body's order (fa[i], fb[j]): is the 2.16 PFLOP/s of mfma32_loop still there?
template <int NI, int NJ> __global__ __launch_bounds__(256) void mfma32_tile_loop(float *out, const bf16x8 *ops, int iters) {
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 a[NI], b[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) a[i] = ops[threadIdx.x + 256 * i];
#pragma unroll
  for (int j = 0; j < NJ; ++j) b[j] = ops[threadIdx.x + 256 * (NI + j)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(b[j]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NI, int NJ> static void run_tile(float *out, const bf16x8 *ops, int cus, hipEvent_t e0, hipEvent_t e1, int iters = 20000, int reps = 1) {
  hipLaunchKernelGGL((mfma32_tile_loop<NI, NJ>), dim3(cus), dim3(256), 0, 0, out, ops, 100);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma32_tile_loop<NI, NJ>), dim3(cus), dim3(256), 0, 0, out, ops, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * NI * NJ * (double)iters * reps * 4.0 * cus;
  printf("32x32x16 tile loop %dx%d accumulators (%d registers), one wave per SIMD, %d launches of %d iterations: %8.3f ms  %7.1f TFLOP/s\n", NI, NJ, NI * NJ * 16, reps, iters, ms, flops / ms / 1e9);
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
  bf16x8 *ops;
  hipMalloc(&ops, 16 * 256 * 8 * 2);
  hipMemset(ops, 0x3c, 16 * 256 * 8);
  run_tile<1, 4>(out, ops, cus, e0, e1);
  run_tile<2, 2>(out, ops, cus, e0, e1);
  run_tile<2, 4>(out, ops, cus, e0, e1);
  run_tile<3, 4>(out, ops, cus, e0, e1);
  run_tile<4, 4>(out, ops, cus, e0, e1);
  // launch-length dependence (a 4096^3 GEMM tile = 256 iterations of 16 MFMAs)
  for (int it : {64, 256, 1024, 4096}) run_tile<4, 4>(out, ops, cus, e0, e1, it, 50);
  hipMemset(ops, 0, 16 * 256 * 8);
  run_tile<4, 4>(out, ops, cus, e0, e1, 256, 50);
  // operand DATA: the rate above is for constant bit patterns; with random bf16 values in [-1, 1) the same instruction stream
  // draws more power and the clock drops
  {
    unsigned short *h = new unsigned short[16 * 256 * 8];
    unsigned s_ = 12345u;
    for (int i = 0; i < 16 * 256 * 8; ++i) {
      s_ = s_ * 1664525u + 1013904223u;
      float f = (float)(s_ >> 8) / 8388608.f - 1.f;
      unsigned u;
      __builtin_memcpy(&u, &f, 4);
      h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(ops, h, 16 * 256 * 8 * 2, hipMemcpyHostToDevice);
    printf("random operands in [-1, 1):\n");
    run_tile<4, 4>(out, ops, cus, e0, e1, 256, 50);
    run_tile<4, 4>(out, ops, cus, e0, e1, 20000, 1);
    run_tile<2, 4>(out, ops, cus, e0, e1, 20000, 1);
    delete[] h;
  }
  return 0;
}
