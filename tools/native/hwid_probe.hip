// Where do the workgroups of a launch land? 512 workgroups of 256 threads with 64 KB of LDS each (two fit a CU, like the 128x128 GEMM
// tiles); each records HW_REG_HW_ID, HW_REG_XCC_ID and its start time. Prints the number of distinct CUs, the workgroups per CU and the
// CU of the first workgroups in launch order.   hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, unsigned long long *t) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    smem[0] = 1;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 500) __builtin_amdgcn_s_sleep(16);   // 5 us: keep the first round resident
}

int main() {
  const int n = 1024;
  unsigned *d;
  unsigned long long *dt;
  hipMalloc(&d, n * 8);
  hipMalloc(&dt, n * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 65536, 0, d, dt);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(2 * n);
  std::vector<unsigned long long> ht(n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  unsigned long long t0 = ht[0];
  for (int i = 0; i < n; ++i) t0 = ht[i] < t0 ? ht[i] : t0;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15u;
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    per_cu[((xcc * 8u + se) * 2u + sh) * 16u + cu]++;
    if (i < 80 || (i >= 256 && i < 272) || (i >= 512 && i < 528))
      printf("wg %4d: hw %08x xcc_reg %08x -> xcc %u se %u sh %u cu %2u simd %u wave %u  start +%.2f us\n", i, hw, h[2 * i + 1], xcc, se, sh, cu, (hw >> 4) & 3u,
             hw & 15u, 0.01 * (double)(ht[i] - t0));
  }
  std::map<int, int> hist;
  for (auto &kv : per_cu) hist[kv.second]++;
  printf("distinct (xcc,se,sh,cu): %zu\n", per_cu.size());
  for (auto &kv : hist) printf("  %d CUs hold %d workgroups of the %d\n", kv.second, kv.first, n);
  return 0;
}
