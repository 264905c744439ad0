// LDS atomic rates on gfx950: one 1024-thread workgroup per CU, a 100 KB tile in LDS, every lane adds to pseudo-random cells.
// Forms: ds_add_f32 (what voxel_bin_kernel uses), ds_add_u32, ds_add_u64, ds_add_f64, and half the lanes masked off (the y-tile test).
//   hipcc --offload-arch=gfx950 -O2 -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int FORM, bool HALF>
__global__ __launch_bounds__(1024) void k(unsigned *sink, int iters, unsigned long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cells = FORM >= 2 ? 12500 : 25000;
  for (int i = threadIdx.x; i < 25000; i += 1024) reinterpret_cast<unsigned *>(smem)[i] = 0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned c = (s >> 8) % cells;
    if (HALF && ((s >> 3) & 1)) continue;
    if constexpr (FORM == 0) atomicAdd(reinterpret_cast<float *>(smem) + c, 0.37f);
    if constexpr (FORM == 1) atomicAdd(reinterpret_cast<unsigned *>(smem) + c, 3u);
    if constexpr (FORM == 2) atomicAdd(reinterpret_cast<unsigned long long *>(smem) + c, 0x100000003ull);
    if constexpr (FORM == 3) atomicAdd(reinterpret_cast<double *>(smem) + c, 0.37);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < 25000; i += 1024) acc ^= reinterpret_cast<unsigned *>(smem)[i];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int FORM, bool HALF> void run(const char *name, unsigned *sink, unsigned long long *dc) {
  auto kk = k<FORM, HALF>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kk, dim3(256), dim3(1024), 100 * 1024, 0, sink, iters, dc);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kk, dim3(256), dim3(1024), 100 * 1024, 0, sink, iters, dc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += (double)h[i];
  mean /= 256;
  const double wave_instrs = 16.0 * iters;                 // per CU
  printf("%-34s %7.1f us per launch | %8.0f s_memtime ticks per CU | %6.1f ticks per wave instruction per CU | %5.2f G lane-atomics/s chip\n", name, ms * 1e3, mean,
         mean / wave_instrs, 256.0 * 1024 * iters * (HALF ? 0.5 : 1.0) / (ms * 1e-3) * 1e-9);
}

int main() {
  unsigned *sink;
  unsigned long long *dc;
  hipMalloc(&sink, 64);
  hipMalloc(&dc, 256 * 8);
  run<0, false>("ds_add_f32, all lanes", sink, dc);
  run<0, true>("ds_add_f32, half the lanes", sink, dc);
  run<1, false>("ds_add_u32, all lanes", sink, dc);
  run<1, true>("ds_add_u32, half the lanes", sink, dc);
  run<2, false>("ds_add_u64, all lanes", sink, dc);
  run<2, true>("ds_add_u64, half the lanes", sink, dc);
  run<3, false>("ds_add_f64, all lanes", sink, dc);
  return 0;
}
