// Development harness (not part of the product): the "G4" body generalised over operand storage and tile shape.
//   C[M][N] (f32) = A . B^T-ish, where each operand is either k-contiguous (A stored [M][K], B stored [N][K]) or k-strided
//   (A stored [K][M], B stored [K][N]). NT = forward (both k-contiguous), NN = data gradient (A k-contiguous, B k-strided),
//   TN = weight gradient (both k-strided; the shipped gemm_g4_tn_body).
// Structure as in g4_gemm.hip's TN body: 4 waves, one per SIMD, v_mfma_f32_32x32x16_bf16, a wave owns (32 FI) x (32 FJ) of a
// (64 FI) x (64 FJ) tile, FOUR-stage LDS ring of 32-k stages filled by LDS-DMA, one barrier per stage, fragment reads of K step
// s+1 issued in front of the MFMAs of step s (inline asm, counted lgkmcnt).
// The question this file answers: does a k-contiguous operand survive 32-k stages (64-byte row segments = half cache lines)?
//
//   hipcc --offload-arch=gfx950 -O3 -o g4x_gemm g4x_gemm.hip && ./g4x_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef uint16_t bf16_t;

template <int OFF> __device__ __forceinline__ u32x4 lds_read128i(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void tie4(u32x4 &x) { asm volatile("" : "+v"(x)); }

template <int N> __device__ __forceinline__ void wait_lgkm() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt range");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int xcd_renumber(int nblk, int bid) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

struct GxParams {
  int M, N, K;
  const bf16_t *A; int lda;
  const bf16_t *B; int ldb;
  float *C; int ldc;
  int tiles_m, tiles_n;
};

// One operand's side of the tile: EXT rows (m or n) = 64 F; KC = k-contiguous storage.
//   KC image  [EXT rows][32 k], 64-byte rows, 16-byte chunk index XORed with (row >> 3) & 3 (conflict-free ds_read_b128);
//             DMA piece = 16 rows x 64 B.
//   !KC image [32 k][EXT], rows of 2 EXT bytes, 64-byte block index XORed with (k & 3); DMA piece = 1024 / (2 EXT) k-rows.
template <bool KC, int F> struct Operand {
  static constexpr int EXT = 64 * F, IMG = EXT * 64, NREAD = KC ? F : 2 * F;
  static_assert(KC || EXT == 256 || EXT == 128, "k-strided operand: tile extent 128 or 256");
  int voff[F];
  unsigned addr[KC ? 2 : F];
  int kstep;
  __device__ __forceinline__ void init(int lane, int wave, int wsel, int origin, int limit, int ld, unsigned img_base) {
    if constexpr (KC) {
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const int row = (wave * F + i) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 3) & 3);
        voff[i] = (origin + row < limit) ? ((origin + row) * ld + chunk * 8) * 2 : (int)0x80000000;
      }
      kstep = 64;
      const int row = wsel * 32 * F + (lane & 31), g = (row >> 3) & 3, h = lane >> 5;
      addr[0] = img_base + row * 64 + (((0 + h) ^ g) << 4);
      addr[1] = img_base + row * 64 + (((2 + h) ^ g) << 4);
    } else {
      constexpr int RPP = 1024 / (2 * EXT), LPR = 64 / RPP;   // k-rows per piece, lanes per k-row
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const int piece = wave * F + i;
        const int kr = piece * RPP + lane / LPR, pos = lane % LPR;
        const int m = (((pos >> 2) ^ (kr & 3)) << 5) + ((pos & 3) << 3);
        voff[i] = (origin + m < limit) ? (kr * ld + origin + m) * 2 : (int)0x80000000;
      }
      kstep = 32 * ld * 2;
      const int h = lane >> 5, sub = (lane >> 4) & 1, q = (lane >> 2) & 3, pq = lane & 3;
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const int blk = wsel * F + i;
        addr[i] = img_base + (8 * h + q) * (2 * EXT) + (16 * sub + 4 * pq) * 2 + ((blk ^ q) << 6);
      }
    }
  }
  // fragment registers: one ds_read_b128 per fragment (k-contiguous) or two transposed 8-byte reads (k-strided); the halves
  // are only put together at the MFMA, after the counted wait (any earlier use would make hipcc wait for the read)
  struct Frags {
    u32x4 v[KC ? F : 1];
    u32x2 h[KC ? 1 : F][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
      for (int i = 0; i < (KC ? F : 1); ++i) v[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < (KC ? 1 : F); ++i) h[i][0] = h[i][1] = u32x2{0u, 0u};
    }
    __device__ __forceinline__ void tie() {
      if constexpr (KC) {
#pragma unroll
        for (int i = 0; i < F; ++i) asm volatile("" : "+v"(v[i]));
      } else {
#pragma unroll
        for (int i = 0; i < F; ++i) {
          asm volatile("" : "+v"(h[i][0]));
          asm volatile("" : "+v"(h[i][1]));
        }
      }
    }
    __device__ __forceinline__ bf16x8 get(int i) const {
      if constexpr (KC) return __builtin_bit_cast(bf16x8, v[i]);
      else return __builtin_bit_cast(bf16x8, (u32x4{h[i][0][0], h[i][0][1], h[i][1][0], h[i][1][1]}));
    }
  };
  template <int KS, int I = 0> __device__ __forceinline__ void read(Frags &f, unsigned soff) const {
    if constexpr (I < F) {
      if constexpr (KC) {
        f.v[I] = lds_read128i<I * 2048>(addr[KS] + soff);
      } else {
        f.h[I][0] = lds_read_tr<KS * 16 * 2 * EXT>(addr[I] + soff);
        f.h[I][1] = lds_read_tr<KS * 16 * 2 * EXT + 4 * 2 * EXT>(addr[I] + soff);
      }
      read<KS, I + 1>(f, soff);
    }
  }
};

// MODE bits: 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no barrier, 16 no vmcnt wait, 32 no lgkmcnt wait, 64 no C store,
// 128 accumulator with n on the lane (dword stores of two full 128-byte lines) instead of m on the lane (16-byte pieces of 64 rows),
// 256 bf16 C (8-byte stores, m on the lane), 512 THREE-stage ring (a 256x128 tile then needs 72 KiB: two workgroups per CU)
template <bool AKC, bool BKC, int FI, int FJ, int MODE>
__device__ __forceinline__ void g4x_body(const GxParams &p, const int tile_m, const int tile_n) {
  using OA = Operand<AKC, FI>;
  using OB = Operand<BKC, FJ>;
  constexpr int STAGE = OA::IMG + OB::IMG, NP = FI + FJ, NMF = FI * FJ, NST = (MODE & 512) ? 3 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * OA::EXT, n0 = tile_n * OB::EXT;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.B), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned smem_base = (unsigned)(uintptr_t)(lds_void *)smem;
  OA oa;
  OB ob;
  oa.init(lane, wave, wm, m0, p.M, p.lda, smem_base);
  ob.init(lane, wave, wn, n0, p.N, p.ldb, smem_base + OA::IMG);

  auto dma_piece = [&](int idx, int t) {          // idx 0..NP-1 (compile time after unrolling)
    char *stage = smem + ((unsigned)t % (unsigned)NST) * STAGE;
    if (idx < FI) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void *)(stage + (wave * FI + idx) * 1024), 16, oa.voff[idx < FI ? idx : 0], t * oa.kstep, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void *)(stage + OA::IMG + (wave * FJ + (idx - FI)) * 1024), 16, ob.voff[idx >= FI ? idx - FI : 0], t * ob.kstep, 0, 0);
  };

  f32x16 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typename OA::Frags a0, a1;
  typename OB::Frags b0, b1;
  a1.zero();
  b1.zero();

  auto tie_all = [&](typename OA::Frags &fa, typename OB::Frags &fb) {
    fa.tie();
    fb.tie();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_block = [&](typename OA::Frags &fa, typename OB::Frags &fb, auto dmac, int tn) {
    constexpr bool DMA = decltype(dmac)::value && !(MODE & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        if constexpr (!(MODE & 2))
          acc[i][j] = (MODE & 128) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.get(i), fb.get(j), acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb.get(j), fa.get(i), acc[i][j], 0, 0, 0);
        const int qn = j * FI + i;
        if constexpr (DMA) {
          const int before = (qn * NP) / NMF, after = ((qn + 1) * NP) / NMF;
#pragma unroll
          for (int pc = before; pc < after; ++pc) dma_piece(pc, tn);
          if (after > before) __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = p.K / 32;
#pragma unroll
  for (int i = 0; i < NP; ++i) dma_piece(i, 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) dma_piece(i, 1);
  if constexpr (NST == 4) {
#pragma unroll
    for (int i = 0; i < NP; ++i) dma_piece(i, 2);
  }
  wait_vm<(NST - 2) * NP>();
  __builtin_amdgcn_s_barrier();

  auto iteration = [&](auto dmac, auto vmc, int t) {
    constexpr int VM = decltype(vmc)::value;
    const unsigned soff = ((unsigned)t % (unsigned)NST) * (unsigned)STAGE;
    if constexpr (!(MODE & 4)) {
      oa.template read<0>(a0, soff);
      ob.template read<0>(b0, soff);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a1, b1, std::false_type{}, 0);              // (t-1, K step 1); zeros at t = 0
    if constexpr (!(MODE & 4)) {
      oa.template read<1>(a1, soff);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 32)) wait_lgkm<OA::NREAD>();                              // the reads of K step 0 have landed
      ob.template read<1>(b1, soff);
    }
    __builtin_amdgcn_sched_barrier(0);
    tie_all(a0, b0);
    mfma_block(a0, b0, dmac, t + NST - 1);
    if constexpr (!(MODE & 32)) wait_lgkm<0>();
    tie_all(a1, b1);
    if constexpr (!(MODE & 16)) {
      if constexpr (VM == 2) wait_vm<2 * NP>();
      else if constexpr (VM == 1) wait_vm<NP>();
      else if constexpr (VM == 0) wait_vm<0>();
    }
    if constexpr (VM >= 0 && !(MODE & 8)) __builtin_amdgcn_s_barrier();
  };
  // nk >= 3 (precondition): the main loop keeps three stages in flight, the last three stages drain
  int t = 0;
  for (; t + NST - 1 < nk; ++t) iteration(std::true_type{}, std::integral_constant<int, NST - 2>{}, t);
  if constexpr (NST == 4) iteration(std::false_type{}, std::integral_constant<int, 1>{}, t++);
  iteration(std::false_type{}, std::integral_constant<int, 0>{}, t);
  iteration(std::false_type{}, std::integral_constant<int, -1>{}, t + 1);
  mfma_block(a1, b1, std::false_type{}, 0);

  if constexpr (MODE & 64) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 1234.5f) p.C[tid] = sum;
  } else if constexpr (MODE & 128) {
    // lane = n, register r: m = .. + 8 (r >> 2) + 4 (lane >> 5) + (r & 3): one store instruction = two rows x 128 contiguous bytes
    const int mrow = m0 + wm * 32 * FI + 4 * (lane >> 5), ncol = n0 + wn * 32 * FJ + (lane & 31);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow + 32 * i + 8 * (r >> 2) + (r & 3);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          const int n = ncol + 32 * j;
          if (n < p.N) p.C[(int64_t)m * p.ldc + n] = acc[i][j][r];
        }
      }
  } else if constexpr (MODE & 256) {
    const int mrow = m0 + wm * 32 * FI + (lane & 31), ncol = n0 + wn * 32 * FJ + 4 * (lane >> 5);
    bf16_t *Cb = reinterpret_cast<bf16_t *>(p.C);
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int m = mrow + 32 * i;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol + 32 * j + 8 * g;
          if (n + 3 >= p.N) continue;
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[i][j][4 * g + e];
          *reinterpret_cast<bf16x4 *>(Cb + (int64_t)m * p.ldc + n) = o;
        }
    }
  } else {
    // C[m][n..n+3]: lane m = .. + (lane & 31); reg r: n = .. + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    const int mrow = m0 + wm * 32 * FI + (lane & 31), ncol = n0 + wn * 32 * FJ + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int m = mrow + 32 * i;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol + 32 * j + 8 * g;
          if (n + 3 >= p.N) continue;
          *reinterpret_cast<float4 *>(p.C + (int64_t)m * p.ldc + n) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
    }
  }
}

template <bool AKC, bool BKC, int FI, int FJ, int MODE>
__global__ __launch_bounds__(256) void g4x_kernel(const GxParams p) {
  const int t_lin = xcd_renumber(gridDim.x, blockIdx.x);
  g4x_body<AKC, BKC, FI, FJ, MODE>(p, t_lin / p.tiles_n, t_lin % p.tiles_n);
}

// ------------------------------------------------------------------------------------------------------------ host
static inline bf16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (bf16_t)(u >> 16);
}
static inline float bf2f(bf16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <bool AKC, bool BKC, int FI, int FJ, int MODE> static float run(const GxParams &p, int reps, hipEvent_t e0, hipEvent_t e1) {
  constexpr int smem = ((MODE & 512) ? 3 : 4) * (64 * FI + 64 * FJ) * 64;
  void (*k)(const GxParams) = g4x_kernel<AKC, BKC, FI, FJ, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  GxParams q = p;
  q.tiles_m = (p.M + 64 * FI - 1) / (64 * FI);
  q.tiles_n = (p.N + 64 * FJ - 1) / (64 * FJ);
  const dim3 grid(q.tiles_m * q.tiles_n);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, grid, dim3(256), smem, 0, q);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(256), smem, 0, q);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); exit(2); }
  return ms / reps * 1e3f;
}

struct Shape { const char *name; int M, N, K; };

template <bool AKC, bool BKC, int FI, int FJ> static int sweep(const char *tag, const std::vector<Shape> &shapes, hipEvent_t e0, hipEvent_t e1) {
  int bad_total = 0;
  for (const Shape &s : shapes) {
    const size_t na = (size_t)s.K * s.M, nb = (size_t)s.K * s.N, nc = (size_t)s.M * s.N;
    std::vector<bf16_t> ha(na), hb(nb);
    std::vector<float> hc(nc);
    for (auto &v : ha) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.f - 1.f);
    for (auto &v : hb) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.f - 1.f);
    bf16_t *da, *db;
    float *dc;
    hipMalloc(&da, na * 2); hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 4);
    hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice);
    hipMemset(dc, 0xFF, nc * 4);
    GxParams p{s.M, s.N, s.K, da, AKC ? s.K : s.M, db, BKC ? s.K : s.N, dc, s.N, 0, 0};
    const int reps = (double)s.M * s.N * s.K > 3e10 ? 5 : 20;
    const float us = run<AKC, BKC, FI, FJ, 128>(p, reps, e0, e1);
    hipMemcpy(hc.data(), dc, nc * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    const size_t nchk = nc <= 400000 ? nc : 3000;
    for (size_t c = 0; c < nchk; ++c) {
      const size_t e = nc <= 400000 ? c : ((size_t)rand() * 2654435761u + c * 7919) % nc;
      const int m = (int)(e / s.N), n = (int)(e % s.N);
      double ref = 0;
      for (int k = 0; k < s.K; ++k) {
        const float av = bf2f(AKC ? ha[(size_t)m * s.K + k] : ha[(size_t)k * s.M + m]);
        const float bv = bf2f(BKC ? hb[(size_t)n * s.K + k] : hb[(size_t)k * s.N + n]);
        ref += (double)av * bv;
      }
      const double got = hc[e];
      if (!(fabs(got - ref) <= 1e-3 * fabs(ref) + 2e-3 * sqrt((double)s.K))) {
        if (bad < 3) printf("\n  MISMATCH m=%d n=%d got %g ref %g", m, n, got, ref);
        ++bad;
      }
    }
    bad_total += bad;
    const int tiles = ((s.M + 64 * FI - 1) / (64 * FI)) * ((s.N + 64 * FJ - 1) / (64 * FJ));
    printf("%s %dx%d %-9s %5dx%5dx%5d %4dt %8.1fus %6.0fTF %6.2fus/K64/round%s", tag, 64 * FI, 64 * FJ, s.name, s.M, s.N, s.K, tiles, us,
           2.0 * s.M * s.N * s.K / us * 1e-6, us / (s.K / 64.0) / ((tiles + 255) / 256), bad ? " BAD" : "");
    printf("  [f32 m-on-lane %.1f, no store %.1f, bf16 C %.1f", run<AKC, BKC, FI, FJ, 0>(p, reps, e0, e1), run<AKC, BKC, FI, FJ, 64>(p, reps, e0, e1),
           run<AKC, BKC, FI, FJ, 256>(p, reps, e0, e1));
    if constexpr (FI * FJ <= 8) printf(", 3-stage ring 2 WG/CU: bf16 C %.1f, no store %.1f", run<AKC, BKC, FI, FJ, 256 + 512>(p, reps, e0, e1), run<AKC, BKC, FI, FJ, 64 + 512>(p, reps, e0, e1));
    printf("]");
    if (s.K >= 4096)
      printf("\n   ablation (results invalid by construction): noDMA %.1f noMFMA %.1f noREAD %.1f noBARRIER %.1f noVMWAIT %.1f noLGKMWAIT %.1f noBAR+noVM+noLGKM %.1f "
             "MFMAonly %.1f MFMA+BARRIER %.1f",
             run<AKC, BKC, FI, FJ, 1>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 2>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 4>(p, 5, e0, e1),
             run<AKC, BKC, FI, FJ, 8>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 16>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 32>(p, 5, e0, e1),
             run<AKC, BKC, FI, FJ, 56>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 1 + 4 + 8 + 16 + 32>(p, 5, e0, e1), run<AKC, BKC, FI, FJ, 1 + 4 + 16 + 32>(p, 5, e0, e1));
    printf("\n");
    fflush(stdout);
    hipFree(da); hipFree(db); hipFree(dc);
  }
  return bad_total;
}

int main(int argc, char **argv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  srand(1);
  // forward shapes (M = tokens, N = out features, K = in features) and data-gradient shapes (N = in features, K = out features)
  const std::vector<Shape> fwd = {{"check", 520, 264, 160},        {"enc.qkv", 6272, 2304, 768},  {"enc.proj", 6272, 768, 768},
                                  {"enc.fc1", 6272, 3072, 768},    {"enc.fc2", 6272, 768, 3072},  {"dec.qkv", 12544, 1536, 512},
                                  {"dec.proj", 12544, 512, 512},   {"dec.fc1", 12544, 2048, 512}, {"dec.fc2", 12544, 512, 2048},
                                  {"sq4096", 4096, 4096, 4096}};
  const std::vector<Shape> dgr = {{"check", 520, 264, 160},        {"enc.dqkv", 6272, 768, 2304}, {"enc.dproj", 6272, 768, 768},
                                  {"enc.dfc1", 6272, 768, 3072},   {"enc.dfc2", 6272, 3072, 768}, {"dec.dqkv", 12544, 512, 1536},
                                  {"dec.dfc1", 12544, 512, 2048},  {"dec.dfc2", 12544, 2048, 512}, {"sq4096", 4096, 4096, 4096}};
  int bad = 0;
  const char *which = argc > 1 ? argv[1] : "all";
  if (argc > 2) {                      // one named shape only (for counter runs)
    for (auto *v : {const_cast<std::vector<Shape> *>(&fwd), const_cast<std::vector<Shape> *>(&dgr)}) {
      std::vector<Shape> keep;
      for (const Shape &s : *v)
        if (!strcmp(s.name, argv[2])) keep.push_back(s);
      *v = keep;
    }
  }
  if (!strcmp(which, "all") || !strcmp(which, "nt")) {
    bad += sweep<true, true, 4, 4>("NT", fwd, e0, e1);
    bad += sweep<true, true, 4, 2>("NT", fwd, e0, e1);
    bad += sweep<true, true, 2, 4>("NT", fwd, e0, e1);
    bad += sweep<true, true, 3, 4>("NT", fwd, e0, e1);
  }
  if (!strcmp(which, "all") || !strcmp(which, "nn")) {
    bad += sweep<true, false, 4, 4>("NN", dgr, e0, e1);
    bad += sweep<true, false, 4, 2>("NN", dgr, e0, e1);
    bad += sweep<true, false, 3, 4>("NN", dgr, e0, e1);
    bad += sweep<true, false, 2, 4>("NN", dgr, e0, e1);
  }
  if (!strcmp(which, "all") || !strcmp(which, "tn")) bad += sweep<false, false, 4, 4>("TN", fwd, e0, e1);
  printf(bad ? "FAILED: %d mismatches\n" : "checks OK\n", bad);
  return bad ? 1 : 0;
}
