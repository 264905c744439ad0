// The "G4" GEMM bodies of libevtpretrain.so: 4 waves, ONE wave per SIMD, v_mfma_f32_32x32x16_bf16, LDS-DMA ring of 32-deep K stages,
// one barrier per stage, fragment reads of K step s+1 issued in front of the MFMAs of step s (inline asm, counted lgkmcnt).
//   gemm_g4_tn_body : weight gradients (both operands k-strided), f32 C (+)=, fused bias column sums
//   g4x_body        : forward (NT) / data gradient (NN) with the epilogues of evp_gemm, bf16 C through an LDS-staged row store
// Device code only; included by csrc/gemm_g4.hip (the product kernels) and by tools/native/g4x_gemm.hip (the timing harness), so
// there is ONE copy of each body in the tree.
#pragma once
#include "gemm_common.h"

namespace {

// ---- "G4" weight-gradient body: 256x256 tile, 4 waves, ONE wave per SIMD, v_mfma_f32_32x32x16_bf16 ---------------------
// C[M][N] (f32) (+)= A^T . B, A stored [K][M], B stored [K][N] (TN), K % 32 == 0, K >= 96.
// Why one wave per SIMD: an 8-wave half-tile ring (rounds 1-2, removed) ran two waves per SIMD through barrier-separated read /
// MFMA phases and needed ~2.0-2.1 us per 64-deep K tile on the weight-gradient shapes. Here a wave owns 128x128 of the tile in 256
// accumulator registers (the whole 512-register budget belongs to it), which (a) needs 0.25 fragment reads per MFMA,
// (b) lets the wave hide its own LDS latency: the fragment reads of K step s+1 are issued in front of the 16 MFMAs of
// step s, no phase barriers -- ONE barrier per 32-deep stage. Both operands are k-strided, so stages can be 32 k-rows
// thin without splitting cache lines: A 16 KiB + B 16 KiB per stage, FOUR stages in a ring = three tiles of LDS-DMA in
// flight (measured: the DMA is hidden completely, tools/native/g4x_gemm.hip `tn`: 1.25-1.3 us per 64-deep K tile on the
// step's weight-gradient shapes, 1186 TFLOP/s at 4096^3 against 849 for the ring). One wave per SIMD only issues the
// 32x32x16 shape at full rate (16x16x32 needs two waves per SIMD).
// LDS image [32 k][256 m] bf16 (512-byte rows); ds_read_b64_tr_b16 serves 32 lanes as 4 k-rows x 64 B, so the 64-byte
// block index is XORed with (k & 3) -- on the LDS-DMA source address and on the read address (same involution).
// RAW / WAR: a stage is read one barrier after every wave's counted vmcnt proved its own pieces landed; it is re-filled
// (tile t+3 into the stage of tile t-1) after the barrier that follows every wave's lgkmcnt(0) on its last reads of it.
template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr_imm(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void tie2(u32x2 &x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ void gemm_g4_tn_body(const GemmParams &p, const int tile_m, const int tile_n) {
  constexpr int IMG = 32 * 512, STAGE = 2 * IMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.B), 0, 0x7FFFFFFF, 0x00020000);
  const int lda = (int)p.lda, ldb = (int)p.ldb;

  // LDS-DMA: piece = 2 k-rows x 512 B, lane-linear in LDS; this wave's 4 pieces of each operand image
  int voffA[4], voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int kr = piece * 2 + (lane >> 5), pos = lane & 31;
    const int m = (((pos >> 2) ^ (kr & 3)) << 5) + ((pos & 3) << 3);
    voffA[i] = (m0 + m < p.M) ? (kr * lda + m0 + m) * 2 : (int)0x80000000;
    voffB[i] = (n0 + m < p.N) ? (kr * ldb + n0 + m) * 2 : (int)0x80000000;
  }
  const int kstepA = 32 * lda * 2, kstepB = 32 * ldb * 2;
  auto dma_piece = [&](int idx, int t) {          // idx 0..7: compile time after unrolling
    char *stage = smem + (t & 3) * STAGE;
    if (idx < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void *)(stage + (wave * 4 + idx) * 1024), 16, voffA[idx & 3], t * kstepA, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void *)(stage + IMG + (wave * 4 + (idx & 3)) * 1024), 16, voffB[idx & 3], t * kstepB, 0, 0);
  };

  // fragment addresses (stage 0, K step 0): lane -> k-row 8h + q (+4 for the second read), 16-lane group `sub`, 4 m at 4 pq
  const int h = lane >> 5, sub = (lane >> 4) & 1, q = (lane >> 2) & 3, pq = lane & 3;
  const unsigned smem_base = (unsigned)(uintptr_t)(lds_void *)smem;
  unsigned aaddr[4], baddr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lanepart = (unsigned)((8 * h + q) * 512 + (16 * sub + 4 * pq) * 2);
    aaddr[i] = smem_base + lanepart + (unsigned)((((wm * 4 + i) ^ q) << 6));
    baddr[i] = smem_base + IMG + lanepart + (unsigned)((((wn * 4 + i) ^ q) << 6));
  }

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x2 a0[4][2], b0[4][2], a1[4][2], b1[4][2];     // [fragment][k 0..3 / 4..7 of the lane's 8]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) a1[i][e] = b1[i][e] = u32x2{0u, 0u};
  // bias gradient riding on the weight gradient: colsum[m] = sum_k A[k][m], taken from the A fragments of the wn == 0 waves of
  // the tile_n == 0 workgroups (every A value is in exactly one of them once); v_dot2 against (1, 1), in the MFMAs' shadow
  const bool do_colsum = p.colsum != nullptr && tile_n == 0 && wn == 0;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  auto readsA = [&](u32x2 (&fa)[4][2], unsigned soff, auto ksc) {
    constexpr int KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i][0] = lds_read_tr_imm<KS * 8192>(aaddr[i] + soff);
      fa[i][1] = lds_read_tr_imm<KS * 8192 + 2048>(aaddr[i] + soff);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto readsB = [&](u32x2 (&fb)[4][2], unsigned soff, auto ksc) {
    constexpr int KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fb[i][0] = lds_read_tr_imm<KS * 8192>(baddr[i] + soff);
      fb[i][1] = lds_read_tr_imm<KS * 8192 + 2048>(baddr[i] + soff);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tie_all = [&](u32x2 (&fa)[4][2], u32x2 (&fb)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { tie2(fa[i][0]); tie2(fa[i][1]); tie2(fb[i][0]); tie2(fb[i][1]); }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_block = [&](u32x2 (&fa)[4][2], u32x2 (&fb)[4][2], auto dmac, int tn) {
    constexpr bool DMA = decltype(dmac)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 av = u32x4{fa[i][0][0], fa[i][0][1], fa[i][1][0], fa[i][1][1]};
        const u32x4 bv = u32x4{fb[j][0][0], fb[j][0][1], fb[j][1][0], fb[j][1][1]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, av), acc[i][j], 0, 0, 0);
        const int qn = j * 4 + i;
        if constexpr (DMA) {
          if ((qn & 1) == 1) {
            dma_piece(qn >> 1, tn);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    if (do_colsum) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (pairs taken with shufflevector from the 8-element fragment: bit-casting single dwords made hipcc 7.2 feed the
        // same dword to several dot instructions -- the ring body above hit the same miscompile)
        const bf16x8 f = __builtin_bit_cast(bf16x8, u32x4{fa[i][0][0], fa[i][0][1], fa[i][1][0], fa[i][1][1]});
        float c = csum[i];
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones, c, false);
        csum[i] = c;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = p.K / 32;                        // >= 3 (launcher)
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_piece(i, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_piece(i, 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_piece(i, 2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // one stage; VM = pieces that may still be in flight at its end (16: two newer tiles, 8: one, 0: none, -1: last stage)
  auto iteration = [&](auto dmac, auto vmc, int t) {
    constexpr int VM = decltype(vmc)::value;
    const unsigned soff = (unsigned)((t & 3) * STAGE);
    readsA(a0, soff, std::integral_constant<int, 0>{});
    readsB(b0, soff, std::integral_constant<int, 0>{});
    mfma_block(a1, b1, std::false_type{}, 0);              // (t-1, K step 1); zeros at t = 0
    readsA(a1, soff, std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the 16 reads of K step 0 have landed (reads return in order)
    readsB(b1, soff, std::integral_constant<int, 1>{});
    tie_all(a0, b0);
    mfma_block(a0, b0, dmac, t + 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_all(a1, b1);
    if constexpr (VM == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (VM == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (VM >= 0) __builtin_amdgcn_s_barrier();
  };
  int t = 0;
  for (; t + 3 < nk; ++t) iteration(std::true_type{}, std::integral_constant<int, 16>{}, t);
  iteration(std::false_type{}, std::integral_constant<int, 8>{}, t);
  iteration(std::false_type{}, std::integral_constant<int, 0>{}, t + 1);
  iteration(std::false_type{}, std::integral_constant<int, -1>{}, t + 2);
  mfma_block(a1, b1, std::false_type{}, 0);

  // C[m][n..n+3]: lane m = .. + (lane & 31); register r: n = .. + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
  float *C = reinterpret_cast<float *>(p.C);
  const int mrow = m0 + wm * 128 + (lane & 31), ncol = n0 + wn * 128 + 4 * (lane >> 5);
  // The tile leaves through the (now idle) 128 KiB ring, 128 rows at a time: stored straight from the accumulators a wave
  // instruction wrote 32 rows x 32 bytes -- 64 cache lines touched per instruction, 64 instructions per lane -- and a round of 256
  // tiles paid 34 us outside its K loop (tools/g4_intercept_probe.py: time against K at 4096 x 4096), most of it this store. Parked
  // row-major with the 16-byte chunk index XORed with (row & 63), every wave instruction then writes ONE whole 1 KiB row.
  {
    (void)mrow; (void)ncol;
    float4 *tile = reinterpret_cast<float4 *>(smem);          // [128 rows][64 chunks]
    const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    const int ncol4 = n0 + 4 * lane;                            // this lane's 4 columns in the row-wise pass
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      __syncthreads();                                         // every wave is done with the ring (or with the previous half)
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 32 * i + (lane & 31);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int chunk = (wn * 128 + 32 * j + 8 * g + 4 * (lane >> 5)) >> 2;
              tile[row * 64 + (chunk ^ (row & 63))] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
        }
      }
      __syncthreads();
      const int mbase = m0 + half * 128;
      float *crow = C + (int64_t)mbase * p.ldc + ncol4;
#pragma unroll 1
      for (int r0 = wave; r0 < 128; r0 += 16) {                // 4 rows per trip and wave
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + 4 * u;
          if (mbase + r >= p.M || ncol4 >= p.N) continue;
          float4 v = tile[r * 64 + (lane ^ (r & 63))];
          float *c = crow + (int64_t)r * p.ldc;
          if (vec_ok && ncol4 + 3 < p.N) {
            if (p.accumulate) {
              const float4 o = *reinterpret_cast<const float4 *>(c);
              v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
            }
            *reinterpret_cast<float4 *>(c) = v;
          } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
            for (int w_ = 0; w_ < 4 && ncol4 + w_ < p.N; ++w_) c[w_] = p.accumulate ? c[w_] + e[w_] : e[w_];
          }
        }
      }
    }
  }
  if (do_colsum) {                               // lanes l and l + 32 hold the two k halves of row (lane & 31)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = csum[i];
      v += __shfl_xor(v, 32, 64);
      const int m = mrow + 32 * i;
      if (lane < 32 && m < p.M) p.colsum[m] = p.colsum_acc ? p.colsum[m] + v : v;
    }
  }
}


// ---- g4x: the same structure generalised over operand storage and tile shape (forward / data gradient) -----------------------
// C[M][N] = epilogue(alpha * sum_k A[m][k] B[n][k]); each operand is k-contiguous (KC: A stored [M][K], B stored [N][K]) or
// k-strided (A stored [K][M], B stored [K][N]). NT = forward (both KC), NN = data gradient (A KC, B strided).
// Tile = (64 FI) x (64 FJ), a wave owns (32 FI) x (32 FJ) = FI x FJ accumulators of 32x32. NST-stage ring of 32-k stages:
//   NST = 4, FI = FJ = 4 : 256x256, 128 KiB of LDS, one workgroup per CU (436 registers)
//   NST = 3, FI*FJ = 8   : 256x128 / 128x256, 72 KiB and <= 256 registers: TWO workgroups per CU, so one's epilogue (a 64 KiB
//                          write burst per tile) runs under the other's K loop
// k-contiguous image [rows][32 k]: 64-byte rows (half cache lines -- measured harmless), 16-byte chunk
// index XORed with (row >> 3) & 3 (conflict-free ds_read_b128), LDS-DMA piece = 16 rows x 64 B.
// k-strided image [32 k][EXT]: 64-byte block index XORed with (k & 3), piece = 1024 / (2 EXT) k-rows (as gemm_g4_tn_body).
template <int OFF> __device__ __forceinline__ u32x4 lds_read128_imm(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt range");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool KC, int F> struct G4Operand {
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
        f.v[I] = lds_read128_imm<I * 2048>(addr[KS] + soff);
      } else {
        f.h[I][0] = lds_read_tr_imm<KS * 16 * 2 * EXT>(addr[I] + soff);
        f.h[I][1] = lds_read_tr_imm<KS * 16 * 2 * EXT + 4 * 2 * EXT>(addr[I] + soff);
      }
      read<KS, I + 1>(f, soff);
    }
  }
};

// ---- g4x epilogues --------------------------------------------------------------------------------------------------------
// Accumulator layout (operand roles swapped in the MFMA): acc[i][j][4 g + e] = C[wm*32FI + 32 i + (lane & 31)]
//                                                                           [wn*32FJ + 32 j + 8 g + 4 (lane >> 5) + e].
// bf16 C: a lane's 8-byte pieces are 64 B apart per row and 32 rows apart per instruction -- written straight to memory that
// pattern ran at 2.5 TB/s and was 10-20 us of a 30-50 us launch (g4x harness, round 2). So the epilogue math (bias, GELU, GELU')
// runs on the f32 accumulators in registers, the rounded bf16 tile is parked in the (now idle) ring -- row-major, 16-byte chunk
// index XORed with (row & 15) -- and every thread then walks rows with a FIXED chunk column: each wave store instruction covers
// 2-4 whole contiguous rows of the tile. The GELU' operand (pre-activation) comes in the same way: rows -> LDS -> lane pieces.
struct G4Piece { float x, y, z, w; };
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  uint2 u;
  u.x = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
  u.y = (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16);
  return u;
}

template <int FI, int FJ, int EPI>
__device__ __forceinline__ void g4x_epilogue_bf16(const f32x16 (&acc)[FI][FJ], const GemmParams &p, char *smem, int m0, int n0, int wm, int wn,
                                                  int tid, int lane) {
  constexpr int BM = 64 * FI, BN = 64 * FJ, ROWB = BN * 2, CPR = BN / 8, RPP = 256 / CPR, STEPS = BM / RPP;
  static_assert(STEPS % 8 == 0, "row passes are unrolled by four / eight");
  bf16_t *C = reinterpret_cast<bf16_t *>(p.C);
  bf16_t *aux = reinterpret_cast<bf16_t *>(p.aux);
  const int lm = lane & 31, lh = lane >> 5;
  const int rc = tid % CPR, rr0 = tid / CPR;          // row pass: this thread's 16-byte chunk column and first row
  const int rn = n0 + rc * 8;                         // N % 8 == 0 (launcher): a chunk is in range as a whole
  const bool interior_m = m0 + BM <= p.M;             // no row of the tile is out of range: no per-row test (branch-free stores)
  auto rows_out = [&](bf16_t *dst, int64_t ld) {
    if (rn >= p.N) return;
    bf16_t *base = dst + (int64_t)m0 * ld + rn;
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 4) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = rr0 + (s + u) * RPP;
        v[u] = *reinterpret_cast<const uint4 *>(smem + row * ROWB + ((rc ^ (row & 15)) << 4));
      }
      if (interior_m) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4 *>(base + (int64_t)(rr0 + (s + u) * RPP) * ld) = v[u];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = rr0 + (s + u) * RPP;
          if (m0 + row < p.M) *reinterpret_cast<uint4 *>(base + (int64_t)row * ld) = v[u];
        }
      }
    }
  };
  auto rows_in = [&](const bf16_t *src, int64_t ld) {
    if (rn >= p.N) return;
    const bf16_t *base = src + (int64_t)m0 * ld + rn;
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = rr0 + (s + u) * RPP;
        const int rowc = (interior_m || m0 + row < p.M) ? row : 0;      // out-of-range rows re-read row 0 (their products are never stored)
        v[u] = *reinterpret_cast<const uint4 *>(base + (int64_t)rowc * ld);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = rr0 + (s + u) * RPP;
        *reinterpret_cast<uint4 *>(smem + row * ROWB + ((rc ^ (row & 15)) << 4)) = v[u];
      }
    }
  };
  // the lane's piece (i, j, g) in the parked tile
  auto piece_ptr = [&](int i, int j, int g) -> uint2 * {
    const int row = wm * 32 * FI + 32 * i + lm, c8 = wn * 4 * FJ + 4 * j + g;
    return reinterpret_cast<uint2 *>(smem + row * ROWB + ((c8 ^ (row & 15)) << 4) + lh * 8);
  };
  const int ncol0 = n0 + wn * 32 * FJ + 4 * lh;
  // the lane's bias pieces, ALL requested before anything waits on one (a test around each load makes hipcc emit load -> s_waitcnt
  // vmcnt(0) per piece: 16 serial L2 round trips, ~12 us of a 35 us launch when first built that way); out-of-range columns read
  // column 0 (N % 8 == 0: a 4-wide piece is in range as a whole or not at all; such pieces are never stored).
  // (the 256x256 form holds 256 accumulator registers: it takes the bias one 32-column block (4 pieces) at a time, the others all
  // of it up front)
  constexpr int JB = (FI * FJ >= 16) ? 1 : FJ;
  // MODE 0: linear, 1: activation of the value, 2: value times act'(parked pre-activation)
  auto park = [&](auto modec) {
    constexpr int MODE = decltype(modec)::value;
#pragma unroll
    for (int jb = 0; jb < FJ; jb += JB) {
      float4 bias4[JB][4];
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[j][g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = ncol0 + 32 * (jb + j) + 8 * g;
            bias4[j][g] = *reinterpret_cast<const float4 *>(p.bias + (n < p.N ? n : 0));
          }
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = jb + jj;
          const float4 b4 = bias4[jj][g];
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            float4 v = make_float4(acc[i][j][4 * g] * p.alpha + b4.x, acc[i][j][4 * g + 1] * p.alpha + b4.y, acc[i][j][4 * g + 2] * p.alpha + b4.z,
                                   acc[i][j][4 * g + 3] * p.alpha + b4.w);
            uint2 *dst = piece_ptr(i, j, g);
            if constexpr (MODE == 1) {
              if (p.act == EVP_ACT_GELU) v = gelu4(v, true);
              else v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            } else if constexpr (MODE == 2) {
              const uint2 hu = *dst;
              const float4 h = make_float4(__uint_as_float(hu.x << 16), __uint_as_float(hu.x & 0xFFFF0000u), __uint_as_float(hu.y << 16),
                                           __uint_as_float(hu.y & 0xFFFF0000u));
              if (p.act == EVP_ACT_DGELU) v = dgelu_mul4(v, h, true);
              else v = make_float4(h.x > 0.f ? v.x : 0.f, h.y > 0.f ? v.y : 0.f, h.z > 0.f ? v.z : 0.f, h.w > 0.f ? v.w : 0.f);
            }
            *dst = pack_bf16x4(v.x, v.y, v.z, v.w);
          }
        }
    }
  };
  __syncthreads();                                   // every wave has finished its fragment reads of the ring
  if constexpr (EPI == 0) {
    park(std::integral_constant<int, 0>{});
  } else if constexpr (EPI == 1) {
    if (p.aux) {                                     // pre-activation first, then the activation over the same LDS tile
      park(std::integral_constant<int, 0>{});
      __syncthreads();
      rows_out(aux, p.ldaux);
      __syncthreads();
    }
    park(std::integral_constant<int, 1>{});
  } else {
    rows_in(aux, p.ldaux);
    __syncthreads();
    park(std::integral_constant<int, 2>{});          // in place: a piece is read and rewritten by the one lane that owns it
  }
  __syncthreads();
  rows_out(C, p.ldc);
}

// f32 C (+ f32 residual / accumulate), linear epilogue: 16-byte pieces straight from the registers (that pattern streams at
// ~5.5 TB/s: 16 B per lane, two lanes per 32-byte sector pair of a row)
template <int FI, int FJ>
__device__ __forceinline__ void g4x_epilogue_f32(const f32x16 (&acc)[FI][FJ], const GemmParams &p, int m0, int n0, int wm, int wn, int lane) {
  float *C = reinterpret_cast<float *>(p.C);
  const int mrow = m0 + wm * 32 * FI + (lane & 31), ncol0 = n0 + wn * 32 * FJ + 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < FJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ncol0 + 32 * j + 8 * g;
      if (n >= p.N) continue;                          // N % 8 == 0 (launcher)
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4 = *reinterpret_cast<const float4 *>(p.bias + n);
      float4 r[FI], c[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int m = mrow + 32 * i;
        r[i] = c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < p.M) {
          if (p.residual) r[i] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldres + n);
          if (p.accumulate) c[i] = *reinterpret_cast<const float4 *>(C + (int64_t)m * p.ldc + n);
        }
      }
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int m = mrow + 32 * i;
        if (m >= p.M) continue;
        const float4 v = make_float4(acc[i][j][4 * g] * p.alpha + b4.x + r[i].x + c[i].x, acc[i][j][4 * g + 1] * p.alpha + b4.y + r[i].y + c[i].y,
                                     acc[i][j][4 * g + 2] * p.alpha + b4.z + r[i].z + c[i].z, acc[i][j][4 * g + 3] * p.alpha + b4.w + r[i].w + c[i].w);
        *reinterpret_cast<float4 *>(C + (int64_t)m * p.ldc + n) = v;
      }
    }
}

template <bool AKC, bool BKC, int FI, int FJ, int NST, typename TC, int EPI>
__device__ __forceinline__ void g4x_body(const GemmParams &p, const int tile_m, const int tile_n) {
  using OA = G4Operand<AKC, FI>;
  using OB = G4Operand<BKC, FJ>;
  constexpr int STAGE = OA::IMG + OB::IMG, NP = FI + FJ, NMF = FI * FJ;
  static_assert(NST == 3 || NST == 4, "ring depth");
  static_assert(sizeof(TC) == 4 || NST * STAGE >= 64 * FI * 64 * FJ * 2, "the bf16 C tile is parked in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * OA::EXT, n0 = tile_n * OB::EXT;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.B), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned smem_base = (unsigned)(uintptr_t)(lds_void *)smem;
  OA oa;
  OB ob;
  oa.init(lane, wave, wm, m0, p.M, (int)p.lda, smem_base);
  ob.init(lane, wave, wn, n0, p.N, (int)p.ldb, smem_base + OA::IMG);

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
    constexpr bool DMA = decltype(dmac)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb.get(j), fa.get(i), acc[i][j], 0, 0, 0);
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

  const int nk = p.K / 32;                        // >= NST - 1 (launcher: K >= 96)
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

  // one stage; VM = tiles that may still be in flight at its end (-1: last stage, no barrier)
  auto iteration = [&](auto dmac, auto vmc, int t) {
    constexpr int VM = decltype(vmc)::value;
    const unsigned soff = ((unsigned)t % (unsigned)NST) * (unsigned)STAGE;
    oa.template read<0>(a0, soff);
    ob.template read<0>(b0, soff);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a1, b1, std::false_type{}, 0);              // (t-1, K step 1); zeros at t = 0
    oa.template read<1>(a1, soff);
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm<OA::NREAD>();                                // the reads of K step 0 have landed (reads return in order)
    ob.template read<1>(b1, soff);
    __builtin_amdgcn_sched_barrier(0);
    tie_all(a0, b0);
    mfma_block(a0, b0, dmac, t + NST - 1);
    wait_lgkm<0>();
    tie_all(a1, b1);
    if constexpr (VM == 2) wait_vm<2 * NP>();
    else if constexpr (VM == 1) wait_vm<NP>();
    else if constexpr (VM == 0) wait_vm<0>();
    if constexpr (VM >= 0) __builtin_amdgcn_s_barrier();
  };
  int t = 0;
  for (; t + NST - 1 < nk; ++t) iteration(std::true_type{}, std::integral_constant<int, NST - 2>{}, t);
  if constexpr (NST == 4) iteration(std::false_type{}, std::integral_constant<int, 1>{}, t++);
  iteration(std::false_type{}, std::integral_constant<int, 0>{}, t);
  iteration(std::false_type{}, std::integral_constant<int, -1>{}, t + 1);
  mfma_block(a1, b1, std::false_type{}, 0);

  if (p.dbg == 1 && acc[0][0][0] != 12345.678f) return;      // measurement aid: no epilogue
  if constexpr (sizeof(TC) == 2) g4x_epilogue_bf16<FI, FJ, EPI>(acc, p, smem, m0, n0, wm, wn, tid, lane);
  else g4x_epilogue_f32<FI, FJ>(acc, p, m0, n0, wm, wn, lane);
}

}  // namespace
