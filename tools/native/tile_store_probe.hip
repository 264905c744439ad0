// What does the store pattern of a GEMM epilogue cost by itself? R rounds of 512 workgroups (256 threads, 64 KB of LDS each: two per
// CU) each write one 128 x 128 tile of a row-major [M, N] matrix, rows N elements apart, the way gemm.hip's epilogue_lds_rows walks a
// tile (32 lanes per row with 8-byte pieces, or 16 lanes per row with 16-byte pieces), optionally after reading a second tensor with
// the same pattern. No arithmetic, no LDS traffic: time per round = the memory system's share of the epilogue.
//   hipcc --offload-arch=gfx950 -O2 -o tile_store_probe tile_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// MODE 0: bf16 tile, 8 B per lane (2 rows per wave instruction)      1: bf16 tile, 16 B per lane (4 rows per wave instruction)
// MODE 2: two bf16 tensors, 8 B per lane each (GELU forward)         3: two bf16 tensors, 16 B per lane
// MODE 4: read bf16 8 B + write bf16 8 B (GELU')                      5: read 16 B + write 16 B
// MODE 6: f32 tile 16 B per lane (32 lanes per row)                   7: f32 read + f32 write (residual form)
template <int MODE>
__global__ __launch_bounds__(256) void tile_io(unsigned short *c, unsigned short *aux, float *cf, float *rf, int N, int tiles_n, int spin) {
  extern __shared__ char smem[];
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
  const int tid = threadIdx.x;
  if (spin) {                                       // stand-in for the K loop: keeps rounds apart like the GEMM does
    // spin < 0: de-phased -- every second workgroup of the FIRST round (two land on a CU back to back: hwid_probe) computes 1.5 periods
    int ticks = spin < 0 ? -spin : spin;
    if (spin < 0 && blockIdx.x < 512 && ((blockIdx.x >> 3) & 1)) ticks += ticks / 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (tid == 0) smem[0] = 1;
  if constexpr (MODE == 0 || MODE == 2 || MODE == 4) {
    const int ch = tid % 32, r0 = tid / 32;
    uint2 v = make_uint2(tid, tile), acc = make_uint2(0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const size_t o = (size_t)(tm * 128 + r0 + s * 8) * N + tn * 128 + ch * 4;
      if constexpr (MODE == 4) { const uint2 h = *reinterpret_cast<const uint2 *>(aux + o); acc.x ^= h.x; acc.y ^= h.y; v.x += h.x; }
      if constexpr (MODE == 2) *reinterpret_cast<uint2 *>(aux + o) = v;
      *reinterpret_cast<uint2 *>(c + o) = v;
    }
    if (acc.x == 0x12345 && acc.y == 0x777) c[0] = 1;
  } else if constexpr (MODE == 1 || MODE == 3 || MODE == 5) {
    const int ch = tid % 16, r0 = tid / 16;
    uint4 v = make_uint4(tid, tile, 1, 2), acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const size_t o = (size_t)(tm * 128 + r0 + s * 16) * N + tn * 128 + ch * 8;
      if constexpr (MODE == 5) { const uint4 h = *reinterpret_cast<const uint4 *>(aux + o); acc.x ^= h.x; acc.y ^= h.w; v.x += h.y; }
      if constexpr (MODE == 3) *reinterpret_cast<uint4 *>(aux + o) = v;
      *reinterpret_cast<uint4 *>(c + o) = v;
    }
    if (acc.x == 0x12345 && acc.y == 0x777) c[0] = 1;
  } else {
    const int ch = tid % 32, r0 = tid / 32;
    float4 v = make_float4(tid, tile, 1, 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const size_t o = (size_t)(tm * 128 + r0 + s * 8) * N + tn * 128 + ch * 4;
      if constexpr (MODE == 7) { const float4 h = *reinterpret_cast<const float4 *>(rf + o); v.x += h.x; v.y += h.w; }
      *reinterpret_cast<float4 *>(cf + o) = v;
    }
  }
}

template <int MODE> float run(unsigned short *c, unsigned short *aux, float *cf, float *rf, int N, int tiles, int spin) {
  auto k = tile_io<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 65536, 0, c, aux, cf, rf, N, N / 128, spin);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 65536, 0, c, aux, cf, rf, N, N / 128, spin);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

int main() {
  const int N = 2048, M = 4 * 512 / (N / 128) * 128;      // 2048 tiles = 4 rounds
  unsigned short *c, *aux;
  float *cf, *rf;
  hipMalloc(&c, (size_t)M * N * 2);
  hipMalloc(&aux, (size_t)M * N * 2);
  hipMalloc(&cf, (size_t)M * N * 4);
  hipMalloc(&rf, (size_t)M * N * 4);
  hipMemset(aux, 1, (size_t)M * N * 2);
  hipMemset(rf, 0, (size_t)M * N * 4);
  const char *names[8] = {"bf16 store 8 B/lane", "bf16 store 16 B/lane", "2 x bf16 store 8 B", "2 x bf16 store 16 B", "bf16 load + store 8 B", "bf16 load + store 16 B",
                          "f32 store 16 B/lane", "f32 load + store 16 B"};
  for (int tiles : {256, 512, 1024, 2048})
    for (int spin : {0, 800, -800}) {
      float t[8];
      t[0] = run<0>(c, aux, cf, rf, N, tiles, spin); t[1] = run<1>(c, aux, cf, rf, N, tiles, spin);
      t[2] = run<2>(c, aux, cf, rf, N, tiles, spin); t[3] = run<3>(c, aux, cf, rf, N, tiles, spin);
      t[4] = run<4>(c, aux, cf, rf, N, tiles, spin); t[5] = run<5>(c, aux, cf, rf, N, tiles, spin);
      t[6] = run<6>(c, aux, cf, rf, N, tiles, spin); t[7] = run<7>(c, aux, cf, rf, N, tiles, spin);
      printf("%4d tiles (%.1f rounds), stand-in K loop %4.1f us%s:", tiles, tiles / 512.0, (spin < 0 ? -spin : spin) * 0.01, spin < 0 ? " de-phased" : "");
      for (int m = 0; m < 8; ++m) printf("  [%s] %.1f", names[m], t[m]);
      printf("  us per launch\n");
    }
  return 0;
}
