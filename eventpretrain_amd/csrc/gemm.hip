// Batched strided GEMM for gfx950 (MI355X): C = epilogue(alpha * A . B^T), MFMA with f32 accumulate.
//
//   bf16 path : v_mfma_f32_16x16x32_bf16, LDS-staged 128x128x64 (or 64x64x64) tiles, double-buffered,
//               XOR-swizzled LDS images; k-strided operands (transA/transB) are read with ds_read_b64_tr_b16.
//   f32  path : v_mfma_f32_16x16x4_f32 (exact fmaf chain) -- the parity mode checked against the CPU oracle.
//
// Operand roles are swapped inside the MFMA (acc = mfma(Bfrag, Afrag)) so that each lane ends up holding FOUR
// CONSECUTIVE n of one output row m: the epilogue (bias, GELU, GELU', residual, accumulate) then runs on 16-byte
// (f32) / 8-byte (bf16) vectors.
//
// Replaces the reference's nn.Linear / matmul / einsum call sites listed in include/evtpretrain.h (evp_gemm).
#include "gemm_common.h"

static int g_gemm_variant = 1;
// A/B switch, bit 0: one-round 256x256 G4 tiles for wide forward GEMMs, bit 1: 128x256 G4 tiles for wide data gradients. OFF by
// default: re-launched back to back the G4 bodies win 5-20 % on those shapes (tools/gemm_g4_sweep.py), but in the replayed step
// they lose -- same-box A/B (tools/ab_step.py, ViT-Base rec, 3 x 15 steps): all off 11.17 ms, 256x256 forward only 11.20, 128x256
// data gradients only 11.31, both 11.31. The tiles stay selectable (evp_gemm_desc::tile 20-22).
static int g_gemm_g4_fwd = 0;
// (Round 3, measured and removed again -- tools/gemm_rounds_probe.py, DESIGN.md section 4 "rounds":
//  * the rows beyond the last whole round of 512 tiles as 64x64 tiles on a forked stream: bit-identical, but the fork / join pair costs
//    more inside a captured graph than the partial round it saves (enc.fc1 48.0 -> 56.4 us, dec.dfc2 57.5 -> 61.7; step 11.34 -> 11.68 ms);
//  * the second workgroup of a CU starting half a tile period late, so that one computes while the other writes: every launch got
//    slower by exactly the delay (dec.dfc2 3 rounds 49.8 -> 55.0 us, 4 rounds 65.1 -> 70.3): the rounds themselves do not speed up;
//  * a K-loop stagger between the two workgroups of a CU (the waves in the odd wave slot of their SIMDs wait 300 / 600 / 1200 / 2400
//    cycles before their first K tile; MI355X_MICROARCH.md "Two waves per SIMD", item 9): step 10.89 / 10.93 / 10.97 / 10.95 against
//    10.91 ms -- nothing.)
static int g_gemm_dbg = 0;
static int g_gemm_wt16 = 1;      // bf16 epilogue in 8-column pieces / 16-byte write-through stores (evp_gemm_set_variant(18 off / 19 on))
static unsigned long long *g_stamp_buf = nullptr;   // measurement aid, see gemm_common.h "in-kernel wall-clock stamps"
static long long g_stamp_slots = 0, g_stamp_next = 0;
unsigned long long *evp_gemm_next_stamp_slot() {
  if (!g_stamp_buf || g_stamp_slots <= 0) return nullptr;
  return g_stamp_buf + (size_t)((g_stamp_next++) % g_stamp_slots) * 2 * EVP_STAMP_WGS;
}

namespace {

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
  static constexpr int BK = 64;    // elements per K tile
  static constexpr int EPC = 8;    // elements per 16-byte chunk
  static constexpr int KSTEP = 32; // K per MFMA
};
template <> struct Cfg<float> {
  static constexpr int BK = 16;
  static constexpr int EPC = 4;
  static constexpr int KSTEP = 4;
};

// ---- LDS image geometry ------------------------------------------------------------------------------------
// Non-transposed operand (k contiguous): image [ROWS][BK].
//   bf16: 128-byte rows, 16-byte chunk index XORed with (row & 7)          -> conflict-free ds_read_b128
//   f32 : rows padded to BK+1 floats                                       -> conflict-free ds_read_b32
// Transposed operand (k strided): image [BK][ROWS].
//   bf16: ROWS*2-byte rows; chunk XOR per the transposed-read rule ("LDS images")
//   f32 : rows padded to ROWS+16 floats
template <typename T, bool TR, int ROWS, int BK> struct Img;

template <int ROWS, int BK> struct Img<bf16_t, false, ROWS, BK> {
  static constexpr int BYTES = ROWS * BK * 2;
  static __device__ __forceinline__ int chunk_off(int row, int ch) {
    if (BK == 64) return row * 128 + ((ch ^ (row & 7)) << 4);
    return row * 64 + (ch << 4);              // BK = 32: 64-byte rows, a 16-row fragment read is 1 KiB contiguous
  }
};
template <int ROWS, int BK> struct Img<bf16_t, true, ROWS, BK> {
  static constexpr int BYTES = BK * ROWS * 2;
  // k = LDS row, ch = 16-byte chunk along the ROWS (m or n) direction
  static __device__ __forceinline__ int chunk_off(int k, int ch) {
    if (ROWS == 256) return k * 512 + ((ch ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 4);
    if (ROWS == 128) return k * 256 + ((ch ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 4);
    else return k * 128 + ((ch ^ ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1)) << 4);
  }
};
template <int ROWS, int BK> struct Img<float, false, ROWS, BK> {
  static constexpr int LD = 16 + 1;
  static constexpr int BYTES = ((ROWS * LD * 4 + 15) / 16) * 16;
};
template <int ROWS, int BK> struct Img<float, true, ROWS, BK> {
  static constexpr int LD = ROWS + 16;
  static constexpr int BYTES = 16 * LD * 4;
};

// ---- global -> register staging of one operand tile --------------------------------------------------------
// Non-transposed: tile = ROWS rows x BK elements, chunks along k. Transposed: BK rows (k) x ROWS elements.
template <typename T, bool TR, int ROWS, int NT, int BK> struct Stage {
  static constexpr int EPC = Cfg<T>::EPC;
  static constexpr int CPR = TR ? ROWS / EPC : BK / EPC;           // chunks per LDS row
  static constexpr int NROW = TR ? BK : ROWS;
  static constexpr int NCHUNK = (NROW * CPR + NT - 1) / NT;         // per thread
  uint4 r[NCHUNK];

  // rs: buffer descriptor of this batch's operand (2 GiB window); row0: first m/n of the tile; k0: first k;
  // nrows: M or N; K: depth bound. Out-of-range chunks get an offset beyond the window: the hardware range check
  // returns zeros, so there is no branch, no select and nothing that forces a wait before the MFMAs.
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int k0, int nrows, int K, int tid) {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
      const int cid = tid + i * NT;
      const int lr = cid / CPR, c = cid % CPR;
      const int gr = TR ? row0 + c * EPC : row0 + lr;   // m / n index
      const int gk = TR ? k0 + lr : k0 + c * EPC;       // k index
      const bool ok = (NROW * CPR % NT == 0 || cid < NROW * CPR) && gr < nrows && gk < K;
      const int off = ok ? (TR ? gk * ld + gr : gr * ld + gk) * (int)sizeof(T) : (int)0x80000000;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      r[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
  __device__ __forceinline__ void store(char *img, int tid) const {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
      const int cid = tid + i * NT;
      const int lr = cid / CPR, c = cid % CPR;
      if (NROW * CPR % NT == 0 || cid < NROW * CPR) {
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<uint4 *>(img + Img<T, TR, ROWS, BK>::chunk_off(lr, c)) = r[i];
        } else {
          float *p = reinterpret_cast<float *>(img) + lr * Img<T, TR, ROWS, BK>::LD + c * 4;
          if (!TR) {  // padded rows are not 16-byte aligned: scalar stores
            p[0] = __uint_as_float(r[i].x); p[1] = __uint_as_float(r[i].y);
            p[2] = __uint_as_float(r[i].z); p[3] = __uint_as_float(r[i].w);
          } else {
            *reinterpret_cast<uint4 *>(p) = r[i];
          }
        }
      }
    }
  }
};
// ---- global -> LDS direct (LDS-DMA) staging of one bf16 operand tile -------------------------------------------
// One wave-instruction (buffer_load_dwordx4 ... lds) lands 64 x 16 B = 1 KiB contiguously in LDS, so the LDS image is
// written linearly and the XOR swizzle is applied to the per-lane SOURCE address instead (the same involution the
// fragment reads use). A 16 KiB tile image = 16 such pieces, 4 per wave (256 threads).
template <bool TR, int ROWS, int NT, int BK> struct GStage {
  static constexpr int PIECES = Img<bf16_t, TR, ROWS, BK>::BYTES / 1024;   // wave-instructions per tile
  static constexpr int PER_WAVE = PIECES / (NT / 64);
  static_assert(PIECES % (NT / 64) == 0, "tile image must split evenly over the waves");
  __device__ __forceinline__ static void issue(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int k0, int nrows, int K, char *img,
                                               int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int piece = wave * PER_WAVE + i;
      int gr, gk;
      if (!TR && BK == 64) {           // 128-byte rows: 8 rows per piece, 8 chunks per row, chunk ^ (row & 7)
        const int row = piece * 8 + (lane >> 3), pos = lane & 7;
        gr = row0 + row;
        gk = k0 + ((pos ^ (row & 7)) << 3);
      } else if (!TR) {                // BK = 32: 64-byte rows, 16 rows per piece, linear
        gr = row0 + piece * 16 + (lane >> 2);
        gk = k0 + ((lane & 3) << 3);
      } else if (ROWS == 128) {        // 256-byte k-rows: 4 per piece, 16 chunks per row
        const int k = piece * 4 + (lane >> 4), pos = lane & 15;
        gk = k0 + k;
        gr = row0 + ((pos ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 3);
      } else {                         // 128-byte k-rows (64-wide tile): 8 per piece, 8 chunks per row
        const int k = piece * 8 + (lane >> 3), pos = lane & 7;
        gk = k0 + k;
        gr = row0 + ((pos ^ ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1)) << 3);
      }
      const bool ok = gr < nrows && gk < K;
      const int off = ok ? (TR ? gk * ld + gr : gr * ld + gk) * 2 : (int)0x80000000;   // out of range -> zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(img + piece * 1024), 16, off, 0, 0, 0);
    }
  }
};

// ---- fragment loads ----------------------------------------------------------------------------------------
// bf16 fragment of the 16 rows starting at `rb` for k-step `ks` (32 deep): lane holds row rb+(l&15), k 8*(l>>4)..+7
template <bool TR, int ROWS, int BK>
__device__ __forceinline__ bf16x8 frag_bf16(const char *img, int rb, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if constexpr (!TR) {
    const int row = rb + i;
    const uint4 v = *reinterpret_cast<const uint4 *>(img + Img<bf16_t, false, ROWS, BK>::chunk_off(row, ks * 4 + g));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const int q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;
    const int ch = (rb >> 3) + (p >> 1);
    const char *a0 = img + Img<bf16_t, true, ROWS, BK>::chunk_off(k, ch) + 8 * (p & 1);
    const char *a1 = img + Img<bf16_t, true, ROWS, BK>::chunk_off(k + 4, ch) + 8 * (p & 1);
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(a0));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(a1));
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  }
}
// The same fragment through inline-asm LDS reads. Why: the compiler's waitcnt pass knows that an LDS-DMA
// (buffer_load ... lds) writes LDS and puts `s_waitcnt vmcnt(0)` in front of every ds_read it can see after one -- which
// silently turned every counted vmcnt(N) of the staging rings into "wait for everything", i.e. no K tile was ever in
// flight across the fragment reads. Reads it cannot see are not guarded; the loops guard them themselves (counted vmcnt
// + barrier before, s_waitcnt lgkmcnt + the `tie` below after).
template <bool TR, int ROWS, int BK>
__device__ __forceinline__ u32x4 frag_bf16_asm(const char *img, int rb, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if constexpr (!TR) {
    const unsigned a = (unsigned)(uintptr_t)(lds_void *)(img + Img<bf16_t, false, ROWS, BK>::chunk_off(rb + i, ks * 4 + g));
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
  } else {
    const int q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;
    const int ch = (rb >> 3) + (p >> 1);
    const unsigned a0 = (unsigned)(uintptr_t)(lds_void *)(img + Img<bf16_t, true, ROWS, BK>::chunk_off(k, ch) + 8 * (p & 1));
    const unsigned a1 = (unsigned)(uintptr_t)(lds_void *)(img + Img<bf16_t, true, ROWS, BK>::chunk_off(k + 4, ch) + 8 * (p & 1));
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  }
}
// orders the consumers of x after every earlier `asm volatile` (the s_waitcnt lgkmcnt in front of it); no instruction
__device__ __forceinline__ void tie(u32x4 &x) { asm volatile("" : "+v"(x)); }

template <bool TR, int ROWS>
__device__ __forceinline__ float frag_f32(const char *img, int rb, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const float *f = reinterpret_cast<const float *>(img);
  if constexpr (!TR) return f[(rb + i) * Img<float, false, ROWS, 16>::LD + ks * 4 + g];
  else return f[(ks * 4 + g) * Img<float, true, ROWS, 16>::LD + rb + i];
}

// ragged right edge (N not a multiple of 4, e.g. attention scores with N = 98): element-wise and out of line. All
// arguments by value: taking the address of the kernel-argument struct would push it (and every pointer derived from
// it) into scratch memory.
template <typename TC>
__device__ __noinline__ void epilogue_edge(TC *C, TC *aux, const float *bias, const float *residual, float alpha, int act,
                                           int accumulate, int splitk, int64_t o, int64_t ao, int64_t ro, int n, int nv,
                                           float a0, float a1, float a2, bool fast) {
  for (int e = 0; e < nv; ++e) {
    float x = (e == 0 ? a0 : e == 1 ? a1 : a2) * alpha;
    if (bias) x += bias[n + e];
    if (act == EVP_ACT_GELU || act == EVP_ACT_RELU) {
      if (aux) ElemIO<TC>::st(aux + ao + e, x);
      x = act == EVP_ACT_GELU ? gelu_sel(x, fast) : fmaxf(x, 0.f);
    } else if (act == EVP_ACT_DGELU || act == EVP_ACT_DRELU) {
      const float h = ElemIO<TC>::ld(aux + ao + e);
      x *= act == EVP_ACT_DGELU ? dgelu_sel(h, fast) : (h > 0.f ? 1.f : 0.f);
    }
    if (residual) x += residual[ro + e];
    if (splitk > 1) { atomicAdd(reinterpret_cast<float *>(C) + o + e, x); continue; }
    if (accumulate) x += ElemIO<TC>::ld(C + o + e);
    ElemIO<TC>::st(C + o + e, x);
  }
}

// EPI: 0 = linear (bias / residual / accumulate), 1 = activation forward (GELU / ReLU, optional pre-activation store),
//      2 = activation backward (multiply by act'(aux))
// One 4-wide piece C[m][n..n+3] of the epilogue; `a` = raw accumulators.
template <typename TC, int EPI>
__device__ __forceinline__ void epi_apply4(float4 a, const GemmParams &p, int64_t coff, int m, int n, float4 bias4, bool fast) {
  TC *C = reinterpret_cast<TC *>(p.C);
  if (n + 3 >= p.N) {
    epilogue_edge<TC>(C, reinterpret_cast<TC *>(p.aux), p.bias, p.residual, p.alpha, p.act, p.accumulate, p.splitk,
                      coff + (int64_t)m * p.ldc + n, coff + (int64_t)m * p.ldaux + n, coff + (int64_t)m * p.ldres + n, n,
                      p.N - n, a.x, a.y, a.z, fast);
    return;
  }
  float4 v = make_float4(a.x * p.alpha + bias4.x, a.y * p.alpha + bias4.y, a.z * p.alpha + bias4.z, a.w * p.alpha + bias4.w);
  const int64_t ao = coff + (int64_t)m * p.ldaux + n;
  if constexpr (EPI == 1) {
    if (p.aux) st4<TC>(reinterpret_cast<TC *>(p.aux) + ao, v);
    if (p.act == EVP_ACT_GELU) v = gelu4(v, fast);
    else v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  } else if constexpr (EPI == 2) {
    const float4 h = ld4<TC>(reinterpret_cast<const TC *>(p.aux) + ao);
    if (p.act == EVP_ACT_DGELU) v = dgelu_mul4(v, h, fast);
    else v = make_float4(h.x > 0.f ? v.x : 0.f, h.y > 0.f ? v.y : 0.f, h.z > 0.f ? v.z : 0.f, h.w > 0.f ? v.w : 0.f);
  }
  if (p.residual) {
    const float4 r = *reinterpret_cast<const float4 *>(p.residual + coff + (int64_t)m * p.ldres + n);
    v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
  }
  const int64_t o = coff + (int64_t)m * p.ldc + n;
  if (p.splitk > 1) {          // split-K partial sums meet in HBM (f32 C, zeroed by the launcher)
    float *c = reinterpret_cast<float *>(p.C) + o;
    atomicAdd(c + 0, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w);
    return;
  }
  if (p.accumulate) {
    const float4 c = ld4<TC>(C + o);
    v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
  }
  st4<TC>(C + o, v);
}

// Interior fast path of the epilogue for one row of NI 4-wide pieces (n = n0 + 16 j): the optional operands are
// template flags, so the code is branch-free and the compiler issues the NI aux / residual / C loads back to back and
// waits once (with run-time `if (p.residual)` tests around every piece it emitted load -> s_waitcnt vmcnt(0) -> store,
// 16 serial HBM round trips per tile: the epilogue then took 8.7k cycles per 128x128 tile against 28k for its K loop).
template <typename TC, int EPI, int NI, bool RES, bool ACC, bool AUXST>
__device__ __forceinline__ void epi_row_fast(const float4 (&a)[NI], const GemmParams &p, const float4 (&bias4)[NI], TC *crow, TC *auxrow,
                                             const float *resrow, bool fast) {
  float4 h[NI], r[NI], c[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    if constexpr (EPI == 2) h[j] = ld4<TC>(auxrow + j * 16);
    if constexpr (RES) r[j] = *reinterpret_cast<const float4 *>(resrow + j * 16);
    if constexpr (ACC) c[j] = ld4<TC>(crow + j * 16);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    float4 v = make_float4(a[j].x * p.alpha + bias4[j].x, a[j].y * p.alpha + bias4[j].y, a[j].z * p.alpha + bias4[j].z,
                           a[j].w * p.alpha + bias4[j].w);
    if constexpr (EPI == 1) {
      if constexpr (AUXST) st4<TC>(auxrow + j * 16, v);
      if (p.act == EVP_ACT_GELU) v = gelu4(v, fast);
      else v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else if constexpr (EPI == 2) {
      if (p.act == EVP_ACT_DGELU) v = dgelu_mul4(v, h[j], fast);
      else v = make_float4(h[j].x > 0.f ? v.x : 0.f, h[j].y > 0.f ? v.y : 0.f, h[j].z > 0.f ? v.z : 0.f, h[j].w > 0.f ? v.w : 0.f);
    }
    if constexpr (RES) v = make_float4(v.x + r[j].x, v.y + r[j].y, v.z + r[j].z, v.w + r[j].w);
    if constexpr (ACC) v = make_float4(v.x + c[j].x, v.y + c[j].y, v.z + c[j].z, v.w + c[j].w);
    st4<TC>(crow + j * 16, v);
  }
}

template <typename TC, int EPI, int MI, int NI, bool RES, bool ACC, bool AUXST>
__device__ __forceinline__ void epilogue_fast(const f32x4 (&acc)[MI][NI], const GemmParams &p, int64_t coff, int mbase, int nbase, bool fast) {
  float4 bias4[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
    bias4[j] = p.bias ? *reinterpret_cast<const float4 *>(p.bias + nbase + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  TC *C = reinterpret_cast<TC *>(p.C) + coff + (int64_t)mbase * p.ldc + nbase;
  TC *aux = reinterpret_cast<TC *>(p.aux) + coff + (int64_t)mbase * p.ldaux + nbase;
  const float *res = p.residual + coff + (int64_t)mbase * p.ldres + nbase;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float4 a[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) a[j] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    if (p.dbg == 2 && i > 0 && a[0].x != 12345.678f) continue;      // measurement aid: a quarter of the stores
    epi_row_fast<TC, EPI, NI, RES, ACC, AUXST>(a, p, bias4, C + (int64_t)i * 16 * p.ldc, aux + (int64_t)i * 16 * p.ldaux,
                                                res + (int64_t)i * 16 * p.ldres, fast);
  }
}

// Straight from the MFMA accumulators: lane holds C[m][n..n+3] with m = mbase + 16 i, n = nbase + 16 j. Interior lanes
// (every piece in range, no split-K) take the branch-free path specialised on which optional operands exist; the rest
// (ragged edges) the generic piece-by-piece path.
template <typename TC, int EPI, int MI, int NI>
__device__ __forceinline__ void epilogue(const f32x4 (&acc)[MI][NI], const GemmParams &p, int64_t coff, int mbase, int nbase, bool fast) {
  if (p.dbg == 1 && acc[0][0][0] != 12345.678f) return;
  const bool interior = mbase + (MI - 1) * 16 < p.M && nbase + (NI - 1) * 16 + 3 < p.N && p.splitk <= 1;
  if (interior) {
    const bool res = p.residual != nullptr, accu = p.accumulate != 0, auxst = p.aux != nullptr;
    if (res) {
      if (accu) epilogue_fast<TC, EPI, MI, NI, true, true, true>(acc, p, coff, mbase, nbase, fast);      // rare: keep one generic-ish form
      else if (auxst || EPI != 1) epilogue_fast<TC, EPI, MI, NI, true, false, true>(acc, p, coff, mbase, nbase, fast);
      else epilogue_fast<TC, EPI, MI, NI, true, false, false>(acc, p, coff, mbase, nbase, fast);
    } else {
      if (accu) epilogue_fast<TC, EPI, MI, NI, false, true, true>(acc, p, coff, mbase, nbase, fast);
      else if (auxst || EPI != 1) epilogue_fast<TC, EPI, MI, NI, false, false, true>(acc, p, coff, mbase, nbase, fast);
      else epilogue_fast<TC, EPI, MI, NI, false, false, false>(acc, p, coff, mbase, nbase, fast);
    }
    return;
  }
  float4 bias4[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = nbase + j * 16;
    bias4[j] = (p.bias && n + 3 < p.N) ? *reinterpret_cast<const float4 *>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mbase + i * 16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nbase + j * 16;
      if (n >= p.N) continue;
      epi_apply4<TC, EPI>(make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]), p, coff, m, n, bias4[j], fast);
    }
  }
}

// Through an f32 tile in LDS (the operand stages, free once the K loop has ended): the accumulators are parked with a
// 16-byte-chunk XOR swizzle (chunk ^ (row & 15): conflict-free ds_write_b128 and ds_read_b128), then every thread
// walks rows with a FIXED 4-column group, so each wave instruction of the epilogue -- the C store, and the aux /
// residual / accumulate loads and the pre-activation store -- covers whole contiguous rows (BN*4 B of f32, BN*2 B of
// bf16) instead of 16 rows x 32-64 B. The direct form above ran the bf16 stores at ~2.6 TB/s and was a quarter of a
// K = 768 GEMM's time.
// (Round 3, tried and removed: requesting the epilogue's READ operands -- the f32 residual of a proj / fc2 forward, the stored
// pre-activation of a GELU' data gradient, cold in the replayed step -- before the K loop and holding them in registers through it,
// 64 VGPRs at two workgroups per CU. The launches that read such operands run 15-25 % slower in the step than re-launched on warm
// caches, the others 2-10 %; same-box A/B of the prefetch: 11.471 against 11.454 ms per ViT-Base step -- nothing. The loads are
// older than the first K tile's LDS-DMA, so the first counted vmcnt waits for them: the latency moves, it does not disappear.
// Also round 3: requesting the read operands of ALL sixteen rows of a thread at the top of the epilogue instead of four rows per trip
// (fully unrolled, 2 waves per SIMD pinned): per-workgroup epilogue 6.27 against 6.43 us (GELU'), 3.43 against 3.45 (f32 + residual),
// step 11.47-11.58 against 11.43-11.46 ms -- the epilogue is not a chain of dependent round trips either. What it is: a chip-wide
// burst. tools/native/tile_store_probe.hip replays just the memory pattern of a round of 512 tiles behind an 8 us stand-in for the K loop:
// +2.4 us per round for the bf16 tile store (16.8 MB at 7 TB/s), +3.6 for two tensors, +5.1..6.7 for load + store, +11.4 for the f32
// load + store -- the same figures the GEMM pays (2.0 / 5.0 / 6.4 / 3.5 us per workgroup, tools/gemm_epilogue_probe.py); 8- and 16-byte
// pieces per lane cost the same.)
template <typename TC, int EPI, int BM, int BN, int NT, bool RES, bool ACC, bool AUXST>
__device__ __forceinline__ void epilogue_lds_rows(const float4 *tile, const GemmParams &p, int64_t coff, int m0, int n, int ch, int r0, float4 bias4,
                                                  bool fast) {
  constexpr int CPR = BN / 4, RPP = NT / CPR, STEPS = BM / RPP;
  static_assert(STEPS % 4 == 0, "rows per thread must come in fours");
  TC *C = reinterpret_cast<TC *>(p.C) + coff + n;
  TC *aux = reinterpret_cast<TC *>(p.aux) + coff + n;
  const float *res = p.residual + coff + n;
#pragma unroll 1
  for (int s = 0; s < STEPS; s += 4) {
    float4 a[4], h[4], r[4], c[4];
    int m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = r0 + (s + u) * RPP;
      m[u] = m0 + row;
      a[u] = tile[row * CPR + (ch ^ (row & 15))];
      if constexpr (EPI == 2) h[u] = ld4<TC>(aux + (int64_t)m[u] * p.ldaux);
      if constexpr (RES) r[u] = *reinterpret_cast<const float4 *>(res + (int64_t)m[u] * p.ldres);
      if constexpr (ACC) c[u] = ld4<TC>(C + (int64_t)m[u] * p.ldc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 v = make_float4(a[u].x * p.alpha + bias4.x, a[u].y * p.alpha + bias4.y, a[u].z * p.alpha + bias4.z, a[u].w * p.alpha + bias4.w);
      if constexpr (EPI == 1) {
        if constexpr (AUXST) st4<TC>(aux + (int64_t)m[u] * p.ldaux, v);
        if (p.act == EVP_ACT_GELU) v = gelu4(v, fast);
        else v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      } else if constexpr (EPI == 2) {
        if (p.act == EVP_ACT_DGELU) v = dgelu_mul4(v, h[u], fast);
        else v = make_float4(h[u].x > 0.f ? v.x : 0.f, h[u].y > 0.f ? v.y : 0.f, h[u].z > 0.f ? v.z : 0.f, h[u].w > 0.f ? v.w : 0.f);
      }
      if constexpr (RES) v = make_float4(v.x + r[u].x, v.y + r[u].y, v.z + r[u].z, v.w + r[u].w);
      if constexpr (ACC) v = make_float4(v.x + c[u].x, v.y + c[u].y, v.z + c[u].z, v.w + c[u].w);
      st4<TC>(C + (int64_t)m[u] * p.ldc, v);
    }
  }
}

// bf16 C, interior tile, no residual / accumulate: every thread takes EIGHT columns (two adjacent 16-byte chunks of the parked f32 tile),
// so that C and the stored pre-activation leave as 16-byte write-through stores and the GELU' operand arrives as 16-byte loads.
template <int EPI, int BM, int BN, int NT, bool AUXST>
__device__ __forceinline__ void epilogue_lds_rows8(const float4 *tile, const GemmParams &p, int64_t coff, int m0, int n0, bool fast) {
  constexpr int CPR = BN / 4, CPR8 = BN / 8, RPP = NT / CPR8, STEPS = BM / RPP, G = (STEPS % 4 == 0) ? 4 : 2;
  static_assert(STEPS % G == 0, "rows per thread must come in pairs");
  const int tid = threadIdx.x, ch8 = tid % CPR8, r0 = tid / CPR8, n = n0 + ch8 * 8;
  bf16_t *Cb = reinterpret_cast<bf16_t *>(p.C), *Ab = reinterpret_cast<bf16_t *>(p.aux);
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (p.bias) {
    b0 = *reinterpret_cast<const float4 *>(p.bias + n);
    b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
  }
#pragma unroll 1
  for (int s = 0; s < STEPS; s += G) {
    float4 a0[G], a1[G];
    uint4 h[G];
    int m[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int row = r0 + (s + u) * RPP;
      m[u] = m0 + row;
      a0[u] = tile[row * CPR + ((2 * ch8) ^ (row & 15))];
      a1[u] = tile[row * CPR + ((2 * ch8 + 1) ^ (row & 15))];
      if constexpr (EPI == 2) h[u] = *reinterpret_cast<const uint4 *>(Ab + coff + (int64_t)m[u] * p.ldaux + n);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      float4 v0 = make_float4(a0[u].x * p.alpha + b0.x, a0[u].y * p.alpha + b0.y, a0[u].z * p.alpha + b0.z, a0[u].w * p.alpha + b0.w);
      float4 v1 = make_float4(a1[u].x * p.alpha + b1.x, a1[u].y * p.alpha + b1.y, a1[u].z * p.alpha + b1.z, a1[u].w * p.alpha + b1.w);
      if constexpr (EPI == 1) {
        if constexpr (AUXST) st8_bf16_wt(Ab, coff + (int64_t)m[u] * p.ldaux + n, v0, v1);
        if (p.act == EVP_ACT_GELU) { v0 = gelu4(v0, fast); v1 = gelu4(v1, fast); }
        else {
          v0 = make_float4(fmaxf(v0.x, 0.f), fmaxf(v0.y, 0.f), fmaxf(v0.z, 0.f), fmaxf(v0.w, 0.f));
          v1 = make_float4(fmaxf(v1.x, 0.f), fmaxf(v1.y, 0.f), fmaxf(v1.z, 0.f), fmaxf(v1.w, 0.f));
        }
      } else if constexpr (EPI == 2) {
        const float4 h0 = make_float4(__uint_as_float(h[u].x << 16), __uint_as_float(h[u].x & 0xFFFF0000u), __uint_as_float(h[u].y << 16),
                                      __uint_as_float(h[u].y & 0xFFFF0000u));
        const float4 h1 = make_float4(__uint_as_float(h[u].z << 16), __uint_as_float(h[u].z & 0xFFFF0000u), __uint_as_float(h[u].w << 16),
                                      __uint_as_float(h[u].w & 0xFFFF0000u));
        if (p.act == EVP_ACT_DGELU) { v0 = dgelu_mul4(v0, h0, fast); v1 = dgelu_mul4(v1, h1, fast); }
        else {
          v0 = make_float4(h0.x > 0.f ? v0.x : 0.f, h0.y > 0.f ? v0.y : 0.f, h0.z > 0.f ? v0.z : 0.f, h0.w > 0.f ? v0.w : 0.f);
          v1 = make_float4(h1.x > 0.f ? v1.x : 0.f, h1.y > 0.f ? v1.y : 0.f, h1.z > 0.f ? v1.z : 0.f, h1.w > 0.f ? v1.w : 0.f);
        }
      }
      st8_bf16_wt(Cb, coff + (int64_t)m[u] * p.ldc + n, v0, v1);
    }
  }
}

template <typename TC, int EPI, int BM, int BN, int NT, int MI, int NI>
__device__ __forceinline__ void epilogue_lds(const f32x4 (&acc)[MI][NI], const GemmParams &p, int64_t coff, char *smem, int m0, int n0,
                                             int wrow, int wcol, bool fast) {
  if (p.dbg == 1 && acc[0][0][0] != 12345.678f) return;
  constexpr int CPR = BN / 4;                      // 16-byte chunks per tile row
  static_assert(CPR >= 16 && NT % CPR == 0, "tile too narrow for the chunk swizzle");
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  float4 *tile = reinterpret_cast<float4 *>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = wrow + i * 16 + li, ch = (wcol >> 2) + j * 4 + lg;
      tile[row * CPR + (ch ^ (row & 15))] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  __syncthreads();
  if constexpr (sizeof(TC) == 2 && BN % 8 == 0 && (BM % (NT / (BN / 8))) == 0 && (BM / (NT / (BN / 8))) % 2 == 0) {
    // whole tile inside C, nothing read or added besides the GELU' operand, 16-byte aligned rows: the 8-column form
    if (p.c_wt16 && m0 + BM <= p.M && n0 + BN <= p.N && !p.residual && !p.accumulate && p.splitk <= 1 && (EPI != 2 || p.aux)) {
      if (EPI == 1 && p.aux) epilogue_lds_rows8<EPI, BM, BN, NT, true>(tile, p, coff, m0, n0, fast);
      else epilogue_lds_rows8<EPI, BM, BN, NT, false>(tile, p, coff, m0, n0, fast);
      return;
    }
  }
  const int ch = tid % CPR, r0 = tid / CPR;
  const int n = n0 + ch * 4;
  if (n >= p.N) return;
  const float4 bias4 = (p.bias && n + 3 < p.N) ? *reinterpret_cast<const float4 *>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int RPP = NT / CPR;                    // rows per pass
  if (m0 + BM <= p.M && n + 3 < p.N && p.splitk <= 1) {      // interior: branch-free, loads batched four rows at a time
    const bool res = p.residual != nullptr, accu = p.accumulate != 0, auxst = p.aux != nullptr;
    if (res) {
      if (accu) epilogue_lds_rows<TC, EPI, BM, BN, NT, true, true, true>(tile, p, coff, m0, n, ch, r0, bias4, fast);
      else if (auxst || EPI != 1) epilogue_lds_rows<TC, EPI, BM, BN, NT, true, false, true>(tile, p, coff, m0, n, ch, r0, bias4, fast);
      else epilogue_lds_rows<TC, EPI, BM, BN, NT, true, false, false>(tile, p, coff, m0, n, ch, r0, bias4, fast);
    } else {
      if (accu) epilogue_lds_rows<TC, EPI, BM, BN, NT, false, true, true>(tile, p, coff, m0, n, ch, r0, bias4, fast);
      else if (auxst || EPI != 1) epilogue_lds_rows<TC, EPI, BM, BN, NT, false, false, true>(tile, p, coff, m0, n, ch, r0, bias4, fast);
      else epilogue_lds_rows<TC, EPI, BM, BN, NT, false, false, false>(tile, p, coff, m0, n, ch, r0, bias4, fast);
    }
    return;
  }
  for (int r = r0; r < BM; r += RPP) {
    const int m = m0 + r;
    if (m >= p.M) break;
    epi_apply4<TC, EPI>(tile[r * CPR + (ch ^ (r & 15))], p, coff, m, n, bias4, fast);
  }
}

// ---- the tile body (shared by the plain and the grouped kernel) ------------------------------------------------
template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, bool GLDS, int STAGES, int BK>
__device__ __forceinline__ void gemm_body(const GemmParams &p, const int tile_m, const int tile_n, const int bz, const int kslice) {
  constexpr int NT = WM * WN * 64;
  constexpr int KSTEP = Cfg<T>::KSTEP;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int A_BYTES = Img<T, TA, BM, BK>::BYTES, B_BYTES = Img<T, TB, BN, BK>::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b0 = bz / p.batch1, b1 = bz % p.batch1;
  const T *A = reinterpret_cast<const T *>(p.A) + b0 * p.sA0 + b1 * p.sA1;
  const T *B = reinterpret_cast<const T *>(p.B) + b0 * p.sB0 + b1 * p.sB1;
  // wave-uniform buffer descriptors (kernel arguments and blockIdx only): hardware bounds check, 32-bit offsets
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(B), 0, 0x7FFFFFFF, 0x00020000);
  const int lda = (int)p.lda, ldb = (int)p.ldb;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kbeg = kslice * p.k_per_split;
  const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int ntiles = (kend - kbeg + BK - 1) / BK;

  auto compute_tile = [&](const char *ia, const char *ib) {
#pragma unroll
    for (int ks = 0; ks < BK / KSTEP; ++ks) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_bf16<TA, BM, BK>(ia, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = frag_bf16<TB, BN, BK>(ib, wn * WTN + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
      } else {
        float af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_f32<TA, BM>(ia, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = frag_f32<TB, BN>(ib, wn * WTN + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  };

  // bf16 + LDS-DMA staging: the fragment reads go through frag_bf16_asm (see there); all reads of the K tile are issued
  // first, the MFMAs of K step ks start when its own 8 reads are back (ds_reads return in order)
  auto compute_tile_asm = [&](const char *ia, const char *ib) {
    if constexpr (sizeof(T) == 2) {
      constexpr int KS = BK / KSTEP;
      static_assert(KS == 1 || KS == 2, "lgkmcnt literals below");
      u32x4 af[KS][MI], bf[KS][NI];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[ks][i] = frag_bf16_asm<TA, BM, BK>(ia, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[ks][j] = frag_bf16_asm<TB, BN, BK>(ib, wn * WTN + j * 16, ks, lane);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        constexpr int PER_KS = (TA ? 2 : 1) * MI + (TB ? 2 : 1) * NI;      // ds_read instructions per K step
        if (ks + 1 < KS && PER_KS <= 15) {
          if constexpr (PER_KS == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          else if constexpr (PER_KS == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) tie(af[ks][i]);
#pragma unroll
        for (int j = 0; j < NI; ++j) tie(bf[ks][j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[ks][j]), __builtin_bit_cast(bf16x8, af[ks][i]),
                                                                acc[i][j], 0, 0, 0);
      }
    }
  };

  if constexpr (GLDS) {
    // LDS-DMA pipeline: tile t+1 streams into the other stage while tile t is multiplied; a counted vmcnt leaves
    // the newest tile's pieces in flight across the raw barrier (a __syncthreads() would drain them).
    using GA = GStage<TA, BM, NT, BK>;
    using GB = GStage<TB, BN, NT, BK>;
    constexpr int INFLIGHT = GA::PER_WAVE + GB::PER_WAVE;   // LDS-DMA pieces per wave and tile
    static_assert(INFLIGHT == 8 || INFLIGHT == 7 || INFLIGHT == 6 || INFLIGHT == 4, "add the vmcnt literal for this tile shape");
    auto wait_all_but_newest_tile = [] {
      if constexpr (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (INFLIGHT == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    if constexpr (STAGES == 2) {
      GA::issue(rsA, lda, m0, kbeg, p.M, kend, smem, wave, lane);
      GB::issue(rsB, ldb, n0, kbeg, p.N, kend, smem + A_BYTES, wave, lane);
      for (int t = 0; t < ntiles; ++t) {
        char *cur = smem + (t & 1) * STAGE_BYTES;
        char *nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
        if (t + 1 < ntiles) {
          GA::issue(rsA, lda, m0, kbeg + (t + 1) * BK, p.M, kend, nxt, wave, lane);
          GB::issue(rsB, ldb, n0, kbeg + (t + 1) * BK, p.N, kend, nxt + A_BYTES, wave, lane);
          wait_all_but_newest_tile();
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        compute_tile_asm(cur, cur + A_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // 3-stage ring, ONE barrier per K tile, two tiles of LDS-DMA in flight: at iteration t every wave first waits
      // for its own pieces of tile t (leaving tile t+1 in flight), the barrier then proves (a) tile t is complete and
      // (b) every wave has finished reading tile t-1, whose stage is the one tile t+2 is issued into right after.
      GA::issue(rsA, lda, m0, kbeg, p.M, kend, smem, wave, lane);
      GB::issue(rsB, ldb, n0, kbeg, p.N, kend, smem + A_BYTES, wave, lane);
      if (ntiles > 1) {
        GA::issue(rsA, lda, m0, kbeg + BK, p.M, kend, smem + STAGE_BYTES, wave, lane);
        GB::issue(rsB, ldb, n0, kbeg + BK, p.N, kend, smem + STAGE_BYTES + A_BYTES, wave, lane);
      }
      int st = 0;                     // stage of tile t
      for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) wait_all_but_newest_tile();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < ntiles) {
          const int s2 = st == 0 ? 2 : st - 1;        // (t + 2) % 3
          GA::issue(rsA, lda, m0, kbeg + (t + 2) * BK, p.M, kend, smem + s2 * STAGE_BYTES, wave, lane);
          GB::issue(rsB, ldb, n0, kbeg + (t + 2) * BK, p.N, kend, smem + s2 * STAGE_BYTES + A_BYTES, wave, lane);
        }
        const char *cur = smem + st * STAGE_BYTES;
        compute_tile_asm(cur, cur + A_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st = st == 2 ? 0 : st + 1;
      }
      __builtin_amdgcn_s_barrier();   // the split-K epilogue reuses the ring as scratch
    }
  } else {
    Stage<T, TA, BM, NT, BK> sa;
    Stage<T, TB, BN, NT, BK> sb;
    sa.load(rsA, lda, m0, kbeg, p.M, kend, tid);
    sb.load(rsB, ldb, n0, kbeg, p.N, kend, tid);
    sa.store(smem, tid);
    sb.store(smem + A_BYTES, tid);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      char *cur = smem + (t & 1) * (A_BYTES + B_BYTES);
      char *nxt = smem + ((t + 1) & 1) * (A_BYTES + B_BYTES);
      const bool more = (t + 1) < ntiles;
      if (more) {
        sa.load(rsA, lda, m0, kbeg + (t + 1) * BK, p.M, kend, tid);
        sb.load(rsB, ldb, n0, kbeg + (t + 1) * BK, p.N, kend, tid);
      }
      compute_tile(cur, cur + A_BYTES);
      if (more) {
        sa.store(nxt, tid);
        sb.store(nxt + A_BYTES, tid);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
  if (p.dbg == 3 && p.stamp && tid == 0 && gridDim.x * gridDim.y * gridDim.z <= (unsigned)EVP_STAMP_WGS)   // measurement aid: the start
    p.stamp[2 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))] =                           // stamp moves to the end of the K loop
        ~(unsigned long long)__builtin_amdgcn_s_memrealtime();
  const int64_t coff = b0 * p.sC0 + b1 * p.sC1;
  const int li = lane & 15, lg = lane >> 4;
  if constexpr (STAGES * (A_BYTES + B_BYTES) >= BM * BN * 4 && EPI == 0 && sizeof(TC) == 4) {
    if (p.splitk > 1) {
      // split-K: stage the f32 tile through LDS so that every atomic wave-instruction adds 256 contiguous bytes of
      // one output row (the full-rate shape for global_atomic_add_f32, MI355X_MICROARCH.md "Global float atomics")
      float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int r = wm * WTM + i * 16 + li, c = wn * WTN + j * 16 + lg * 4;
          *reinterpret_cast<float4 *>(tile + r * BN + c) =
              make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
        }
      __syncthreads();
      float *C = reinterpret_cast<float *>(p.C) + coff;
      for (int e = tid; e < BM * BN; e += NT) {
        const int r = e / BN, c = e % BN;
        if (m0 + r < p.M && n0 + c < p.N) atomicAdd(C + (int64_t)(m0 + r) * p.ldc + n0 + c, tile[e]);
      }
      return;
    }
  }
  const bool fast = sizeof(T) == 2;   // bf16 mode may use the fast erf; f32 parity mode uses erff
  if constexpr (sizeof(T) == 2 && STAGES * (A_BYTES + B_BYTES) >= BM * BN * 4 && BN >= 64) {
    epilogue_lds<TC, EPI, BM, BN, NT, MI, NI>(acc, p, coff, smem, m0, n0, wm * WTM, wn * WTN, fast);   // every K loop ends on a barrier
  } else {
    epilogue<TC, EPI, MI, NI>(acc, p, coff, m0 + wm * WTM + li, n0 + wn * WTN + lg * 4, fast);
  }
}

template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, bool GLDS, int STAGES, int BK, int MINW>
__global__ __launch_bounds__(WM *WN * 64, MINW) void gemm_kernel(const GemmParams p) {
  int tile_m, tile_n;
  const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  stamp_begin(p.stamp, wg, gridDim.x * gridDim.y * gridDim.z);
  map_tile(gridDim.x, blockIdx.x, p.tiles_m, tile_m, tile_n);
  gemm_body<T, TC, EPI, TA, TB, BM, BN, WM, WN, GLDS, STAGES, BK>(p, tile_m, tile_n, blockIdx.z, blockIdx.y);
  stamp_end(p.stamp, wg, gridDim.x * gridDim.y * gridDim.z);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_grouped_tn_kernel(const GroupedProblem *__restrict__ probs, const GroupedItem *__restrict__ items,
                                                              unsigned long long *stamp) {
  const GroupedItem it = items[blockIdx.x];
  if (it.prob < 0) return;                       // padding of the per-XCD item lists
  stamp_begin(stamp, blockIdx.x, gridDim.x);
  const GroupedProblem g = probs[it.prob];
  GemmParams p;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.A = g.A; p.lda = g.lda; p.sA0 = 0; p.sA1 = 0;
  p.B = g.B; p.ldb = g.ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = g.C; p.c_dtype = EVP_F32; p.ldc = g.ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = g.accumulate; p.tiles_m = 0; p.splitk = 1; p.dbg = 0; p.colsum = nullptr; p.colsum_acc = 0; p.stamp = nullptr; p.c_wt16 = 0;
  p.k_per_split = (g.K + 63) / 64 * 64;
  gemm_body<bf16_t, float, 0, true, true, BM, BN, 2, 2, true, 2, 64>(p, it.tile_m, it.tile_n, 0, 0);
  stamp_end(stamp, blockIdx.x, gridDim.x);
}

template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, bool GLDS, int STAGES, int BK = Cfg<T>::BK,
          int MINW = 1>
int launch(const evp_gemm_desc *d, hipStream_t s) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = d->strideA0; p.sA1 = d->strideA1;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = d->strideB0; p.sB1 = d->strideB1;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = d->strideC0; p.sC1 = d->strideC1;
  p.batch1 = d->batch1 > 0 ? d->batch1 : 1;
  p.alpha = d->alpha; p.bias = d->bias; p.act = d->act; p.aux = d->aux; p.ldaux = d->ldaux;
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate; p.dbg = g_gemm_dbg; p.colsum = nullptr; p.colsum_acc = 0; p.stamp = evp_gemm_next_stamp_slot();
  p.tiles_m = (d->M + BM - 1) / BM;
  const int tiles_n = (d->N + BN - 1) / BN;
  {
    // the 8-column epilogue form: bf16 C (and aux) in one piece below 2 GiB, rows and bases 16-byte aligned
    const int nbz = (d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
    const int64_t span = (int64_t)d->M * (d->ldc > d->ldaux ? d->ldc : d->ldaux) * 2;
    p.c_wt16 = (g_gemm_wt16 && d->c_dtype == EVP_BF16 && nbz == 1 && span < 0x7FFFFFFFLL && d->ldc % 8 == 0 && (!d->aux || d->ldaux % 8 == 0) &&
                ((uintptr_t)d->C & 15) == 0 && ((uintptr_t)d->aux & 15) == 0 && (!d->bias || ((uintptr_t)d->bias & 15) == 0)) ? 1 : 0;
  }
  const int nb = (d->batch0 > 0 ? d->batch0 : 1) * p.batch1;
  // split-K: only for plain f32 outputs without epilogue extras (the weight-gradient GEMMs: few output tiles, long K)
  constexpr int BKc = BK;
  int splitk = d->splitk;
  const bool can_split = d->c_dtype == EVP_F32 && !d->bias && d->act == EVP_ACT_NONE && !d->residual && !d->aux && nb == 1;
  if (splitk == 0) {
    splitk = 1;
    const int tiles = p.tiles_m * tiles_n;
    const int ktiles = (d->K + BKc - 1) / BKc;
    // automatic split-K only for the weight-gradient layout: f32 atomics are order-dependent in the last bits, and
    // activations must stay bit-reproducible run to run
    if (can_split && d->transA && tiles < 384 && ktiles >= 16) {
      splitk = (512 + tiles - 1) / tiles;
      if (splitk > ktiles / 8) splitk = ktiles / 8;
      if (splitk < 1) splitk = 1;
    }
  }
  if (!can_split) splitk = 1;
  {
    const int ktiles = (d->K + BKc - 1) / BKc;
    const int per = (ktiles + splitk - 1) / splitk;
    p.k_per_split = per * BKc;
    splitk = (ktiles + per - 1) / per;
    p.splitk = splitk;
  }
  if (splitk > 1 && !d->accumulate) {
    hipError_t e = evp_zero2d_async(d->C, (size_t)d->ldc * 4, (size_t)d->N * 4, (size_t)d->M, s);
    if (e != hipSuccess) { evp_set_error("evp_gemm: memset for split-K failed: %s", hipGetErrorString(e)); return EVP_ELAUNCH; }
  }
  constexpr int smem = STAGES * (Img<T, TA, BM, BK>::BYTES + Img<T, TB, BN, BK>::BYTES);
  auto k = gemm_kernel<T, TC, EPI, TA, TB, BM, BN, WM, WN, GLDS, STAGES, BK, MINW>;
  static bool attr_done = false;  // one flag per instantiation
  if (!attr_done) {
    if (smem > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) {
        evp_set_error("evp_gemm: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
        return EVP_ELAUNCH;
      }
    }
    attr_done = true;
  }
  dim3 grid((unsigned)(p.tiles_m * tiles_n), (unsigned)splitk, (unsigned)nb);
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm");
  return EVP_OK;
}

// (Two persistent variants of the 128x128 body lived here in round 1 -- one workgroup per CU slot looping over tiles, the second
// with the C stores of tile i folded into the K loop of tile i+1. Measured: +3..8 % on the largest shapes, slower on the small
// ones; never the default, removed in round 2. That measurement predated the finding that hipcc guards every visible ds_read behind
// an LDS-DMA with vmcnt(0), so round 4 re-built the idea with every LDS access in inline asm and the vector-memory counter accounted
// by hand (`gemm_p3_kernel`, commit "gemm_p3_kernel: persistent 128x128 kernel with the epilogue drained ..."): 512 resident
// workgroups, a finished tile's accumulators parked in a second register set and drained in eight 16-row pieces through two 8 KiB
// LDS buffers during the next tile's first eight K iterations, one 16-byte store per thread and iteration issued right behind the
// top-of-iteration wait. Bit-identical to this kernel on every epilogue it took -- and slower everywhere: re-launched, enc qkv
// forward 37.0 against 32.8 us, enc fc1 + GELU 55.0 / 44.8, dec fc1 + GELU 58.4 / 48.9, GELU' data gradients 89-97 / 50-54 (the NN
// layout spills: acc + parked + fragments need > 256 registers at two workgroups per CU), 4096^3 140 / 134; in the replayed ViT-Base
// step 12.21 against 10.80 ms (profiles/r04_gemm_p3_probe.txt, r04_ab_gemm_p3.txt). Why it cannot win here even without the spills:
// the step's launches hold 1.7-3 tiles per resident workgroup, so at most one or two epilogues per workgroup can hide under a
// following K loop while the last one -- every workgroup has one -- pays a longer, serial tail. Removed again.)

template <typename T, typename TC, int EPI, bool TA, bool TB> int pick_tile(const evp_gemm_desc *d, hipStream_t s) {
  int tile = d->tile;
  const int64_t nb = (int64_t)(d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
  if (tile == 0) {
    // wide outputs (qkv / fc1 forward, fc2 data gradient): the G4 bodies (gemm_g4.hip), unless switched off for A/B runs
    if constexpr (sizeof(T) == 2 && !TA) {
      if (g_gemm_g4_fwd && g_gemm_variant != 2) {
        const int shape = evp_g4_gemm_pick(d);
        if ((shape == 20 && (g_gemm_g4_fwd & 1)) || (shape == 22 && (g_gemm_g4_fwd & 2))) return evp_g4_gemm(d, s, shape, g_gemm_dbg);
      }
    }
    const int64_t t128 = (int64_t)((d->M + 127) / 128) * ((d->N + 127) / 128) * nb;
    const bool splittable = d->transA && d->c_dtype == EVP_F32 && !d->bias && d->act == EVP_ACT_NONE && !d->residual &&
                            !d->aux && nb == 1 && d->K >= 2048;
    tile = (d->M >= 128 && d->N >= 128 && (t128 >= 192 || splittable)) ? 1 : 2;
    // One round of 128 x 128 tiles that leaves some CUs with two workgroups and the rest with one (257..384 tiles) ends when the
    // doubly loaded CUs do; 96 x 128 tiles still fit one round (<= 512) and are 25 % smaller: -13..16 % on the 6272 x 768 outputs
    // of the encoder (294 -> 396 tiles; tools/gemm_tiles.py), bit-identical results (same k order per element).
    if constexpr (sizeof(T) == 2 && !TA) {
      const int64_t t96 = (int64_t)((d->M + 95) / 96) * ((d->N + 127) / 128) * nb;
      if (tile == 1 && g_gemm_variant != 2 && nb == 1 && t128 > 256 && t96 <= 512) tile = 4;
    }
  }
  if (tile == 6 || tile == 7 || tile == 8 || tile == 12 || tile == 30 || tile == 31) {
    evp_set_error("evp_gemm: tile %d (256x256 ring / persistent / stream-K variants of rounds 1-4) was removed; see DESIGN.md section 4", tile);
    return EVP_EUNSUPPORTED;
  }
  if constexpr (sizeof(T) == 2) {
    if (tile == 9) {        // G4 body (weight-gradient layout only): 256x256, one wave per SIMD, 32x32x16
      if constexpr (TA && TB && EPI == 0 && std::is_same<TC, float>::value) {
        if (nb != 1 || d->K % 32 != 0 || d->K < 96 || d->bias || d->residual || d->aux || d->alpha != 1.0f || d->splitk > 1 ||
            d->lda % 8 != 0 || d->ldb % 8 != 0) {
          evp_set_error("evp_gemm: tile 9 (G4) needs transA, transB, f32 C, K %% 32 == 0, K >= 96, no batch / epilogue extras");
          return EVP_ESHAPE;
        }
        return evp_g4_gemm_tn(d, s);
      } else {
        evp_set_error("evp_gemm: tile 9 (G4) is built for the weight-gradient layout (transA = transB = 1, f32 C) only");
        return EVP_EUNSUPPORTED;
      }
    }
    if (tile >= 20 && tile <= 22) {     // G4 forward / data-gradient bodies, explicit shape
      if constexpr (!TA) return evp_g4_gemm(d, s, tile, g_gemm_dbg);
      evp_set_error("evp_gemm: tiles 20-22 need A row-major");
      return EVP_EUNSUPPORTED;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_gemm_variant != 2) {
      // 96 x 128: for launches whose 128 x 128 tiling leaves most CUs with one workgroup and a few with two (294 tiles of a
      // 6272 x 768 output -> 396 tiles): the launch ends when the doubly loaded CUs do, and their tiles are 25 % smaller
      if (tile == 4) return launch<T, TC, EPI, TA, TB, 96, 128, 2, 2, true, 2>(d, s);
      // (round 3: 64 x 128 and 128 x 64 half tiles at three workgroups per CU were instantiated and swept on every narrow-output shape
      //  of the step -- N = 768 / 512, tools/gemm_narrow_sweep.py -- and lost to the choice below by 5-30 %: removed again)
      if (tile == 1) return launch<T, TC, EPI, TA, TB, 128, 128, 2, 2, true, 2>(d, s);
      return launch<T, TC, EPI, TA, TB, 64, 64, 2, 2, true, 2>(d, s);
    }
  }
  if (tile == 1 || tile >= 3) return launch<T, TC, EPI, TA, TB, 128, 128, 2, 2, false, 2>(d, s);
  return launch<T, TC, EPI, TA, TB, 64, 64, 2, 2, false, 2>(d, s);
}

// Layout x epilogue combinations that exist on this path (everything else is refused, not silently emulated):
//   linear epilogue: NT (forward), NN (dgrad), TN (wgrad);  activation forward: NT only;  activation backward: NN only.
template <typename T, typename TC> int pick_layout(const evp_gemm_desc *d, hipStream_t s) {
  const int epi = (d->act == EVP_ACT_GELU || d->act == EVP_ACT_RELU) ? 1 : (d->act == EVP_ACT_DGELU || d->act == EVP_ACT_DRELU) ? 2 : 0;
  if (epi == 0) {
    if (!d->transA && !d->transB) return pick_tile<T, TC, 0, false, false>(d, s);
    if (!d->transA && d->transB) return pick_tile<T, TC, 0, false, true>(d, s);
    if (d->transA && d->transB) return pick_tile<T, TC, 0, true, true>(d, s);
    evp_set_error("evp_gemm: layout transA=1,transB=0 is not used on this path and not built");
    return EVP_EUNSUPPORTED;
  }
  if (epi == 1 && !d->transA && !d->transB) return pick_tile<T, TC, 1, false, false>(d, s);
  if (epi == 2 && !d->transA && d->transB) return pick_tile<T, TC, 2, false, true>(d, s);
  evp_set_error("evp_gemm: activation epilogue %d is only built for its own layout (forward: NT, backward: transB)", d->act);
  return EVP_EUNSUPPORTED;
}

template <typename T> int pick_ctype(const evp_gemm_desc *d, hipStream_t s) {
  if (d->c_dtype == EVP_F32) return pick_layout<T, float>(d, s);
  if constexpr (sizeof(T) == 2) return pick_layout<T, bf16_t>(d, s);
  evp_set_error("evp_gemm: bf16 output needs bf16 operands");
  return EVP_EUNSUPPORTED;
}

}  // namespace

extern "C" int evp_gemm_grouped_tn_bf16(const void *problems, const void *items, int n_items, void *stream) {
  EVP_CHECK_ARG(problems && items && n_items > 0, EVP_EINVAL, "evp_gemm_grouped_tn_bf16: bad argument");
  auto k = gemm_grouped_tn_kernel<128, 128>;
  constexpr int smem = 2 * (Img<bf16_t, true, 128, 64>::BYTES + Img<bf16_t, true, 128, 64>::BYTES);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_gemm_grouped_tn_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)n_items), dim3(256), smem, (hipStream_t)stream,
                     reinterpret_cast<const GroupedProblem *>(problems), reinterpret_cast<const GroupedItem *>(items), evp_gemm_next_stamp_slot());
  EVP_CHECK_LAUNCH("evp_gemm_grouped_tn_bf16");
  return EVP_OK;
}

extern "C" int evp_gemm_set_stamp_buffer(void *buf, long long n_slots) {
  g_stamp_buf = reinterpret_cast<unsigned long long *>(buf);
  g_stamp_slots = buf ? n_slots : 0;
  g_stamp_next = 0;
  return EVP_OK;
}
extern "C" long long evp_gemm_stamp_count(void) { return g_stamp_next; }

extern "C" int evp_gemm_set_variant(int v) {
  const int old = g_gemm_variant;
  if (v >= 100 && v <= 103) g_gemm_dbg = v - 100;
  if (v == 18 || v == 19) g_gemm_wt16 = v - 18;
  if (v == 1 || v == 2) g_gemm_variant = v;
  if (v >= 10 && v <= 13) g_gemm_g4_fwd = v == 11 ? 3 : v == 10 ? 0 : v - 11;     // 10 off (default), 11 on, 12 = 256x256 only, 13 = 128x256 only
  return old;
}

extern "C" int evp_gemm(const evp_gemm_desc *d, void *stream) {
  EVP_CHECK_ARG(d != nullptr, EVP_EINVAL, "evp_gemm: null descriptor");
  EVP_CHECK_ARG(d->A && d->B && d->C, EVP_EINVAL, "evp_gemm: null operand");
  EVP_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, EVP_ESHAPE, "evp_gemm: M,N,K must be positive (%d,%d,%d)", d->M, d->N, d->K);
  EVP_CHECK_ARG(d->dtype == EVP_F32 || d->dtype == EVP_BF16, EVP_EINVAL, "evp_gemm: bad dtype %d", d->dtype);
  EVP_CHECK_ARG(d->c_dtype == EVP_F32 || d->c_dtype == EVP_BF16, EVP_EINVAL, "evp_gemm: bad c_dtype %d", d->c_dtype);
  const int epc = d->dtype == EVP_BF16 ? 8 : 4;
  EVP_CHECK_ARG(d->lda % epc == 0 && d->ldb % epc == 0, EVP_ESHAPE,
                "evp_gemm: lda/ldb must be multiples of %d elements (got %lld, %lld)", epc, (long long)d->lda, (long long)d->ldb);
  EVP_CHECK_ARG(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, EVP_EINVAL, "evp_gemm: A/B must be 16-byte aligned");
  EVP_CHECK_ARG(d->strideA0 % epc == 0 && d->strideA1 % epc == 0 && d->strideB0 % epc == 0 && d->strideB1 % epc == 0,
                EVP_ESHAPE, "evp_gemm: batch strides of A/B must be multiples of %d elements", epc);
  EVP_CHECK_ARG(!d->accumulate || d->c_dtype == EVP_F32, EVP_EINVAL, "evp_gemm: accumulate needs an f32 C");
  {
    const int64_t es = d->dtype == EVP_BF16 ? 2 : 4;
    const int64_t spanA = (int64_t)(d->transA ? d->K : d->M) * d->lda * es, spanB = (int64_t)(d->transB ? d->K : d->N) * d->ldb * es;
    EVP_CHECK_ARG(spanA < 0x7FFFFFFFLL && spanB < 0x7FFFFFFFLL, EVP_ESHAPE,
                  "evp_gemm: one batch's operand must span < 2 GiB (32-bit buffer offsets)");
  }
  EVP_CHECK_ARG((d->act != EVP_ACT_DGELU && d->act != EVP_ACT_DRELU) || d->aux, EVP_EINVAL, "evp_gemm: dgelu/drelu need aux");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == EVP_BF16 ? pick_ctype<bf16_t>(d, s) : pick_ctype<float>(d, s);
}
