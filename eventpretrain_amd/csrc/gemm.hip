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
#include "evp_common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

static int g_gemm_variant = 1;
static int g_gemm_dbg = 0;
static unsigned long long *g_gemm_dbgbuf = nullptr;

namespace {

struct GemmParams {
  int M, N, K;
  const void *A; int64_t lda, sA0, sA1;
  const void *B; int64_t ldb, sB0, sB1;
  void *C; int c_dtype; int64_t ldc, sC0, sC1;
  int batch1;
  float alpha;
  const float *bias;
  int act;
  void *aux; int64_t ldaux;
  const float *residual; int64_t ldres;
  int accumulate;
  int tiles_m;
  int splitk, k_per_split;   // blockIdx.y = K slice
  int dbg;                   // measurement aid (evp_gemm_set_variant(101): skip the epilogue; results are then garbage)
  unsigned long long *dbgbuf; // measurement aid: per-workgroup cycle counters of the persistent kernel
  float *colsum;             // 256x256 TN body only: colsum[m] (+)= sum_k A[k][m] (bias gradient), written by the tile_n == 0 workgroups
  int colsum_acc;
  // stream-K (gemm_sk_kernel): partial accumulator tiles and their ready flags, in the caller's workspace
  float *sk_ws;
  int *sk_flags;
  int sk_kiters;             // K tiles per output tile
  long long sk_total;        // output tiles x sk_kiters
};

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
typedef __attribute__((address_space(3))) void lds_void;
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
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
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

// ---- epilogue helpers ----------------------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): one v_exp + one v_rcp + 5 fma. Used in bf16 mode only.
__device__ __forceinline__ void erf_pdf_fast(float x, float &erfv, float &pdf) {
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t = __frcp_rn(1.0f + 0.3275911f * az);
  const float u = __expf(-0.5f * x * x);          // e^{-x^2/2};  e^{-z^2} = u
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * u;
  erfv = z < 0.f ? -e : e;
  pdf = 0.39894228040143267794f * u;
}
__device__ __forceinline__ float gelu_sel(float x, bool fast) {
  if (!fast) return gelu_f(x);
  float e, pdf;
  erf_pdf_fast(x, e, pdf);
  return 0.5f * x * (1.0f + e);
}
__device__ __forceinline__ float dgelu_sel(float x, bool fast) {     // ragged-edge (scalar) path only
  if (!fast) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
  }
  float e, pdf;
  erf_pdf_fast(x, e, pdf);
  return 0.5f * (1.0f + e) + x * pdf;
}

// bf16-mode activations on PAIRS of values (v_pk_fma_f32 / v_pk_mul_f32: two f32 per lane and instruction) and without
// transcendentals: Phi(x) = 0.5 + xc*P(xc^2) and phi(x) = Q(xc^2) with xc = clamp(x, -4, 4), P / Q degree-7 / -8
// minimax fits (|Phi err| <= 5.3e-5, |x*phi err| <= 5.2e-5 in f32 Horner form; beyond +-4 the clamp leaves <= 5e-4).
// The GELU epilogue of a 128x128 tile was ~25 VALU-equivalents per element -- as long as the tile's whole MFMA work at
// K = 768; this is ~6. f32 parity mode keeps erff / expf.
typedef float __attribute__((ext_vector_type(2))) f32x2;
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 cdf_poly2(f32x2 xc, f32x2 t) {
  f32x2 p = splat2(-1.580980095e-09f);
  p = fma2(p, t, splat2(1.217218683e-07f));
  p = fma2(p, t, splat2(-4.101103530e-06f));
  p = fma2(p, t, splat2(8.067003135e-05f));
  p = fma2(p, t, splat2(-1.048219917e-03f));
  p = fma2(p, t, splat2(9.664920407e-03f));
  p = fma2(p, t, splat2(-6.617543876e-02f));
  p = fma2(p, t, splat2(3.988475314e-01f));
  return fma2(p, xc, splat2(0.5f));
}
__device__ __forceinline__ f32x2 pdf_poly2(f32x2 t) {
  f32x2 q = splat2(8.990855908e-10f);
  q = fma2(q, t, splat2(-7.519181097e-08f));
  q = fma2(q, t, splat2(2.756920725e-06f));
  q = fma2(q, t, splat2(-5.866515477e-05f));
  q = fma2(q, t, splat2(8.084384011e-04f));
  q = fma2(q, t, splat2(-7.582483969e-03f));
  q = fma2(q, t, splat2(4.857881561e-02f));
  q = fma2(q, t, splat2(-1.984161263e-01f));
  return fma2(q, t, splat2(3.986868918e-01f));
}
__device__ __forceinline__ f32x2 clamp4(f32x2 x) {
  return f32x2{__builtin_amdgcn_fmed3f(x.x, -4.f, 4.f), __builtin_amdgcn_fmed3f(x.y, -4.f, 4.f)};
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 xc = clamp4(x);
  return x * cdf_poly2(xc, xc * xc);
}
__device__ __forceinline__ f32x2 dgelu_fast2(f32x2 h) {       // Phi(h) + h*phi(h)
  const f32x2 xc = clamp4(h), t = xc * xc;
  return fma2(xc, pdf_poly2(t), cdf_poly2(xc, t));
}
__device__ __forceinline__ float4 gelu4(float4 v, bool fast) {
  if (fast) {
    const f32x2 a = gelu_fast2(f32x2{v.x, v.y}), b = gelu_fast2(f32x2{v.z, v.w});
    return make_float4(a.x, a.y, b.x, b.y);
  }
  return make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
}
__device__ __forceinline__ float4 dgelu_mul4(float4 v, float4 h, bool fast) {      // v * gelu'(h)
  if (fast) {
    const f32x2 a = f32x2{v.x, v.y} * dgelu_fast2(f32x2{h.x, h.y}), b = f32x2{v.z, v.w} * dgelu_fast2(f32x2{h.z, h.w});
    return make_float4(a.x, a.y, b.x, b.y);
  }
  return make_float4(v.x * dgelu_sel(h.x, false), v.y * dgelu_sel(h.y, false), v.z * dgelu_sel(h.z, false), v.w * dgelu_sel(h.w, false));
}

template <typename TC> __device__ __forceinline__ float4 ld4(const TC *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
template <typename TC> __device__ __forceinline__ void st4(TC *p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, float4 v) {
  uint2 u;
  u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
  u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
  *reinterpret_cast<uint2 *>(p) = u;
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

constexpr int SK_FLAG_BYTES = 4096;              // ready flags (one per workgroup) + the error word, at the start of the workspace
constexpr int SK_MAX_GRID = SK_FLAG_BYTES / 4 - 8;
constexpr int SK_ERR_WORD = SK_FLAG_BYTES / 4 - 1;
// stream-K segment context (gemm_sk_kernel): the XCD's iteration range [base, base + span) is cut into gx equal runs, run
// j belongs to workgroup blockIdx = j * 8 + xcd; mode 1 = park the accumulators, 2 = whole tile, 3 = collect + finish
struct SkCtx {
  int mode, kbeg, kend, j, gx, xcd;
  long long base, span, tile_begin;
  __device__ __forceinline__ long long run_begin(int jj) const { return base + (long long)jj * span / gx; }
};

// ---- the tile body (shared by the plain and the grouped kernel) ------------------------------------------------
template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, bool GLDS, int STAGES, int BK>
__device__ __forceinline__ void gemm_body(const GemmParams &p, const int tile_m, const int tile_n, const int bz, const int kslice,
                                          const SkCtx sk = SkCtx{0, 0, 0, 0, 1, 0, 0, 0, 0}) {
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

  const int kbeg = sk.mode ? sk.kbeg : kslice * p.k_per_split;
  const int kend = sk.mode ? sk.kend : ((kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K);
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

  // ---- stream-K hand-over (mode 1: this segment does not end its tile -> park the accumulators; mode 3: this segment
  //      ends a tile that earlier workgroups of the same XCD started -> collect theirs). See gemm_sk_kernel.
  if (sk.mode == 1) {
    const int slot = sk.j * 8 + sk.xcd;
    float4 *ws = reinterpret_cast<float4 *>(p.sk_ws + (size_t)slot * (BM * BN));
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        ws[(i * NI + j) * NT + tid] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    // Producer and consumer sit on ONE XCD (blockIdx and blockIdx - 8k): they share its L2, the vector L1 is
    // write-through, so "stores acknowledged" (vmcnt 0, which the workgroup-scope release waits for) is all the
    // consumer needs -- no L2 write-back / invalidate (an agent-scope fence per wave cost ~140 us per launch here).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.sk_flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (sk.mode == 3) {
    __shared__ int sk_ok;
    if (p.dbgbuf && tid == 0) p.dbgbuf[blockIdx.x * 16 + 13] = __builtin_readcyclecounter();   // main loop done
    for (int jj = sk.j - 1; jj >= 0; --jj) {
      const int slot = jj * 8 + sk.xcd;
      if (tid == 0) {
        int spins = 0, got = 0;
        while (!(got = __hip_atomic_load(p.sk_flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && ++spins < (1 << 18))
          __builtin_amdgcn_s_sleep(8);
        if (got) __hip_atomic_store(p.sk_flags + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one consumer per flag
        else __hip_atomic_store(p.sk_flags + SK_ERR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // gave up
        sk_ok = got;
        if (p.dbgbuf) p.dbgbuf[blockIdx.x * 16 + 14] = __builtin_readcyclecounter();   // flag seen
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the slot's lines were never in this CU's L1 (one reader per slot and launch)
      if (sk_ok) {
        const float4 *ws = reinterpret_cast<const float4 *>(p.sk_ws + (size_t)slot * (BM * BN));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const float4 t = ws[(i * NI + j) * NT + tid];
            acc[i][j][0] += t.x; acc[i][j][1] += t.y; acc[i][j][2] += t.z; acc[i][j][3] += t.w;
          }
      }
      __syncthreads();    // sk_ok is rewritten by the next round
      if (sk.run_begin(jj) <= sk.tile_begin) break;   // that run held the tile's first K tile
    }
    if (p.dbgbuf && tid == 0) p.dbgbuf[blockIdx.x * 16 + 15] = __builtin_readcyclecounter();   // parts added
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
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

// Block -> tile map (speed only, never correctness): (1) blocks b and b+8 share an XCD, so renumber to give every XCD
// a contiguous run of tiles (bijective for any grid size); (2) inside the run walk GROUP_M x tiles_n panels, M
// fastest, so the ~64 blocks an XCD runs at once share 8 A panels and 8 B panels that fit its 4 MiB L2.
__device__ __forceinline__ int xcd_renumber(int nblk, int bid) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int &tile_m, int &tile_n) {
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int gid = t / per_group, first_m = gid * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int in_g = t - gid * per_group;
  tile_m = first_m + in_g % gsz;
  tile_n = in_g / gsz;
}
__device__ __forceinline__ void map_tile(int nblk, int bid, int tiles_m, int &tile_m, int &tile_n) {
  tile_of(xcd_renumber(nblk, bid), tiles_m, nblk / tiles_m, tile_m, tile_n);
}

template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, bool GLDS, int STAGES, int BK, int MINW>
__global__ __launch_bounds__(WM *WN * 64, MINW) void gemm_kernel(const GemmParams p) {
  int tile_m, tile_n;
  map_tile(gridDim.x, blockIdx.x, p.tiles_m, tile_m, tile_n);
  gemm_body<T, TC, EPI, TA, TB, BM, BN, WM, WN, GLDS, STAGES, BK>(p, tile_m, tile_n, blockIdx.z, blockIdx.y);
}

// ---- stream-K form of the same body ---------------------------------------------------------------------------------
// The forward / dgrad GEMMs of this path have 294-1176 output tiles for 512 resident workgroups: a data-parallel launch
// leaves up to 43 % of the slots idle in its last round, and all workgroups reach their C-tile stores together. Here a
// grid of the resident workgroups cuts the iteration space (output tile x K tile, tile-major) into equal runs -- per
// XCD: each XCD owns an eighth of the tiles and splits it over its own workgroups (blockIdx % 8), so partial tiles
// never cross XCDs and the panels an XCD streams stay in its L2. A run is [tail of a tile][whole tiles][head of a
// tile] and is walked BACKWARDS: the head segment (which does not end its tile) comes first and parks its accumulators
// (64 KiB, raw register layout, coalesced) in the workspace; the tail segment comes last, and the workgroup that ends
// a tile collects the parked parts of the workgroups before it (blockIdx - 8, -16, ...). So a wait never depends on
// another wait (no chain), it only ever targets workgroups that were dispatched EARLIER (no deadlock even if the grid
// were not fully resident), and the summation order is fixed (results are run-to-run identical). Flags are consumed
// (reset to 0) by their single reader, so a replayed HIP graph finds them clean. The poll is bounded: a lost producer
// costs a wrong tile and an error word, never a hung GPU.
// Measured (tools/gemm_sk_check.py, gemm_sk_trace.py; MI355X): bit-reproducible and equal to the data-parallel result
// to f32 rounding, but 4-15 us SLOWER than it on every shape of this path (enc.fc2 56.9 vs 53.1 us). Per workgroup:
// park 2.2 us, wait for the neighbour's flag 5.4 us (all runs are equally long, so the part a workgroup needs is
// finished just when it asks for it -- no slack), adding the parked parts 5.0 us (32 MB of parks do not stay in the
// 4 MiB L2s), one extra pipeline fill per segment. Kept as an explicit variant (evp_gemm_desc::tile = 12) with its
// parity test; the next thing to try is the hybrid: one whole tile per CU for the first 256 tiles, stream-K only for
// the remainder, so that hand-overs sit on workgroups with slack.
template <typename T, typename TC, int EPI, bool TA, bool TB, int BM, int BN, int WM, int WN, int STAGES, int BK>
__global__ __launch_bounds__(WM *WN * 64, 2) void gemm_sk_kernel(const GemmParams p) {
  const int g = gridDim.x;                         // multiple of 8
  SkCtx sk;
  sk.xcd = blockIdx.x & 7;
  sk.j = blockIdx.x >> 3;
  sk.gx = g >> 3;
  const int tiles_n = (p.N + BN - 1) / BN;
  const long long tiles = (long long)p.tiles_m * tiles_n;
  const long long t_lo = tiles * sk.xcd / 8, t_hi = tiles * (sk.xcd + 1) / 8;   // this XCD's tiles: no hand-over crosses XCDs
  sk.base = t_lo * p.sk_kiters;
  sk.span = (t_hi - t_lo) * p.sk_kiters;
  const long long it0 = sk.run_begin(sk.j);
  long long it1 = sk.run_begin(sk.j + 1);
  // measurement aid (evp_gemm_set_debug_buffer): 16 words per workgroup: start, then (end stamp, mode << 16 | K tiles) per segment
  int dbg_n = 0;
  if (p.dbgbuf && threadIdx.x == 0) p.dbgbuf[blockIdx.x * 16] = __builtin_readcyclecounter();
  while (it1 > it0) {                              // last segment of the run first
    const int t = (int)((it1 - 1) / p.sk_kiters);
    sk.tile_begin = (long long)t * p.sk_kiters;
    const long long seg0 = it0 > sk.tile_begin ? it0 : sk.tile_begin;
    const int k0 = (int)(seg0 - sk.tile_begin), k1 = (int)(it1 - sk.tile_begin);
    int tile_m, tile_n;
    tile_of(t, p.tiles_m, tiles_n, tile_m, tile_n);
    sk.mode = k1 < p.sk_kiters ? 1 : (k0 > 0 ? 3 : 2);
    sk.kbeg = k0 * BK;
    sk.kend = k1 * BK < p.K ? k1 * BK : p.K;
    gemm_body<T, TC, EPI, TA, TB, BM, BN, WM, WN, true, STAGES, BK>(p, tile_m, tile_n, 0, 0, sk);
    __syncthreads();      // the epilogue's LDS scratch is the next segment's staging ring
    if (p.dbgbuf && threadIdx.x == 0 && dbg_n < 7) {
      p.dbgbuf[blockIdx.x * 16 + 1 + 2 * dbg_n] = __builtin_readcyclecounter();
      p.dbgbuf[blockIdx.x * 16 + 2 + 2 * dbg_n] = ((unsigned long long)sk.mode << 16) | (unsigned)(k1 - k0);
      ++dbg_n;
    }
    it1 = seg0;
  }
}

// ---- grouped weight-gradient GEMM: many independent (dY^T . X) problems in ONE launch ------------------------------
// problem g: C_g[M_g, N_g] (f32) = A_g^T . B_g with A_g stored [K_g][M_g], B_g stored [K_g][N_g] (bf16). Work item =
// one 128x128 output tile of one problem; items are listed largest-K first so the long tiles start early.
struct GroupedProblem {
  const void *A, *B;
  void *C;
  int M, N, K;
  int lda, ldb, ldc;
  int accumulate, colsum_accumulate;
  float *colsum;             // 256x256 kernel only: colsum[m] (+)= sum_k A[k][m]; NULL = none
};
struct GroupedItem { int prob, tile_m, tile_n, pad; };

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_grouped_tn_kernel(const GroupedProblem *__restrict__ probs, const GroupedItem *__restrict__ items) {
  const GroupedItem it = items[blockIdx.x];
  if (it.prob < 0) return;                       // padding of the per-XCD item lists
  const GroupedProblem g = probs[it.prob];
  GemmParams p;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.A = g.A; p.lda = g.lda; p.sA0 = 0; p.sA1 = 0;
  p.B = g.B; p.ldb = g.ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = g.C; p.c_dtype = EVP_F32; p.ldc = g.ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = g.accumulate; p.tiles_m = 0; p.splitk = 1; p.dbg = 0; p.dbgbuf = nullptr; p.colsum = nullptr; p.colsum_acc = 0;
  p.k_per_split = (g.K + 63) / 64 * 64;
  gemm_body<bf16_t, float, 0, true, true, BM, BN, 2, 2, true, 2, 64>(p, it.tile_m, it.tile_n, 0, 0);
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
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate; p.dbg = g_gemm_dbg; p.dbgbuf = g_gemm_dbgbuf; p.colsum = nullptr; p.colsum_acc = 0;
  p.tiles_m = (d->M + BM - 1) / BM;
  const int tiles_n = (d->N + BN - 1) / BN;
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


// stream-K launch of the 128x128 LDS-DMA body (bf16, A row-major). Workspace (caller-owned, evp_gemm_desc::sk_workspace):
// [0, 4096) int32 ready flags + error word -- zeroed ONCE by the caller, kept clean by the kernel -- then one 64 KiB
// accumulator slot per workgroup.
template <typename TC, int EPI, bool TB> int sk_resident_blocks() {
  static int blocks = -1;   // per instantiation
  if (blocks < 0) {
    constexpr int smem = 2 * (Img<bf16_t, false, 128, 64>::BYTES + Img<bf16_t, TB, 128, 64>::BYTES);
    auto k = gemm_sk_kernel<bf16_t, TC, EPI, false, TB, 128, 128, 2, 2, 2, 64>;
    if (smem > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return blocks = 0;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, smem) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return blocks = 0;
    if (per_cu > 2) per_cu = 2;
    blocks = per_cu * cus;
    if (blocks > SK_MAX_GRID) blocks = SK_MAX_GRID;
    blocks &= ~7;             // whole workgroups per XCD
  }
  return blocks;
}
// usable for this call? (shape and workspace); *grid = workgroups to launch
template <typename TC, int EPI, bool TB> bool sk_usable(const evp_gemm_desc *d, int *grid) {
  const int64_t nb = (int64_t)(d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
  if (nb != 1 || d->K % 64 != 0 || d->K < 256 || !d->sk_workspace || d->splitk > 1) return false;
  const int g = sk_resident_blocks<TC, EPI, TB>();
  const int64_t tiles = (int64_t)((d->M + 127) / 128) * ((d->N + 127) / 128);
  if (g < 8 || tiles * (d->K / 64) < (int64_t)g * 4) return false;
  if (d->sk_workspace_bytes < SK_FLAG_BYTES + (int64_t)g * 128 * 128 * 4) return false;
  *grid = g;
  return true;
}
template <typename TC, int EPI, bool TB> int launch_sk(const evp_gemm_desc *d, hipStream_t s, int g) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = 0; p.sA1 = 0;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1;
  p.alpha = d->alpha; p.bias = d->bias; p.act = d->act; p.aux = d->aux; p.ldaux = d->ldaux;
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate; p.dbg = g_gemm_dbg; p.dbgbuf = g_gemm_dbgbuf; p.colsum = nullptr; p.colsum_acc = 0;
  p.tiles_m = (d->M + 127) / 128;
  p.splitk = 1; p.k_per_split = d->K;
  p.sk_flags = reinterpret_cast<int *>(d->sk_workspace);
  p.sk_ws = reinterpret_cast<float *>(reinterpret_cast<char *>(d->sk_workspace) + SK_FLAG_BYTES);
  p.sk_kiters = d->K / 64;
  p.sk_total = (long long)p.tiles_m * ((d->N + 127) / 128) * p.sk_kiters;
  constexpr int smem = 2 * (Img<bf16_t, false, 128, 64>::BYTES + Img<bf16_t, TB, 128, 64>::BYTES);
  hipLaunchKernelGGL((gemm_sk_kernel<bf16_t, TC, EPI, false, TB, 128, 128, 2, 2, 2, 64>), dim3((unsigned)g), dim3(256), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm(stream-K)");
  return EVP_OK;
}

// ---- 256x256x64 tile, 8 waves, half-tile ring ("8-phase" schedule) ----------------------------------------------
// The 128x128 body above re-reads 16 KiB of LDS per wave for every 32 MFMAs, which is exactly the LDS port's rate at
// MFMA peak: that structure tops out near 900 TFLOP/s. Here a wave owns 128x64 of a 256x256 tile (24 KiB per 64 MFMAs)
// and the K loop is cut into four phases per K tile, each = [fragment ds_reads | one half-tile of LDS-DMA prefetch |
// counted vmcnt] barrier [16 MFMAs on one 64x32 quadrant] barrier. The two wave rows (wr = 0 / 1; waves w and w+4 share
// a SIMD) run one barrier apart, so one row's MFMA cluster overlaps the other row's ds_read / prefetch segment.
// (Issuing the next phase's ds_reads under the same wave's MFMAs instead measured 20 % slower: the segments must stay
// separate.) One workgroup per CU (128 KiB of LDS): it pays where there are many tiles and a long K -- the weight
// gradients (K = batch x tokens) -- and not on the forward / dgrad shapes of this path, whose 75-300 tiles of 256x256
// quantise badly on 256 CUs.
//
// Half-tiles (128 rows x 64 k, 16 KiB) in the order the phases first need them:
//   h=0 A(mh=0): rows wr*128 +      [0,64)   needed in phase 1        h=2 B(nh=1): cols wc*64 + 32 + [0,32)  phase 2
//   h=1 B(nh=0): cols wc*64 +       [0,32)   needed in phase 1        h=3 A(mh=1): rows wr*128 + 64 + [0,64) phase 3
// Ring of 8 slots (2 K tiles x 4). Half-tile s = 4t+h is issued in global phase s-5 and every phase ends its first
// segment with vmcnt(6): everything but the three newest half-tiles has landed, i.e. all s <= g+2, which is what phase
// g+1 reads (RAW: the wait sits before that phase's barriers, the read one phase later). A slot is re-staged >= 3
// phases after its last ds_read (WAR), also across the one-barrier stagger of the two wave rows.
template <typename TC, int EPI, bool TA, bool TB>
__device__ __forceinline__ void gemm256_body(const GemmParams &p, const int tile_m, const int tile_n, const int bz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HALF = 16384;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int b0 = bz / p.batch1, b1 = bz % p.batch1;
  const bf16_t *A = reinterpret_cast<const bf16_t *>(p.A) + b0 * p.sA0 + b1 * p.sA1;
  const bf16_t *B = reinterpret_cast<const bf16_t *>(p.B) + b0 * p.sB0 + b1 * p.sB1;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(B), 0, 0x7FFFFFFF, 0x00020000);
  const int lda = (int)p.lda, ldb = (int)p.ldb;
  const int nk = p.K / 64;

  // per-lane source byte offsets of this wave's two 1-KiB pieces of each half-tile (the K-tile advance goes in soffset).
  // An LDS-DMA piece is written lane-linearly, so the image's XOR swizzle is applied to the source address.
  int voff[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;
      const bool isA = (h == 0) || (h == 3);
      const bool tr = isA ? TA : TB;
      const int sub = (h >= 2) ? 1 : 0;                 // mh for the A halves, nh for the B halves
      const int ld = isA ? lda : ldb, lim = isA ? p.M : p.N, o0 = isA ? m0 : n0;
      int off;
      if (!tr) {                                        // k contiguous: 8 rows x 8 chunks per piece
        const int row = piece * 8 + (lane >> 3), pos = lane & 7;
        const int gr = isA ? o0 + (row >> 6) * 128 + sub * 64 + (row & 63) : o0 + (row >> 5) * 64 + sub * 32 + (row & 31);
        off = gr < lim ? (gr * ld + ((pos ^ (row & 7)) << 3)) * 2 : (int)0x80000000;
      } else {                                          // k strided: 4 k-rows x 16 chunks per piece
        const int k = piece * 4 + (lane >> 4), pos = lane & 15;
        const int rl = (pos ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 3;
        const int gr = isA ? o0 + (rl >> 6) * 128 + sub * 64 + (rl & 63) : o0 + (rl >> 5) * 64 + sub * 32 + (rl & 31);
        off = gr < lim ? (k * ld + gr) * 2 : (int)0x80000000;
      }
      voff[h][i] = off;
    }
  }
  const int kstepA = TA ? 64 * lda * 2 : 128, kstepB = TB ? 64 * ldb * 2 : 128;   // bytes per K tile

  auto issue = [&](auto Hc, int t) {
    constexpr int h = decltype(Hc)::value;
    char *slot = smem + (((t & 1) << 2) + h) * HALF + wave * 2048;
    if constexpr (h == 0 || h == 3) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void *)(slot), 16, voff[h][0], t * kstepA, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void *)(slot + 1024), 16, voff[h][1], t * kstepA, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void *)(slot), 16, voff[h][0], t * kstepB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void *)(slot + 1024), 16, voff[h][1], t * kstepB, 0, 0);
    }
  };
  auto wait_halves = [](int rem) {                      // leave the `rem` (0..3) newest half-tiles in flight
    if (rem >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[4][2], bq0[2][2], bq1[2][2];    // fragments, read through frag_bf16_asm (the compiler must not guard them with vmcnt(0))
  // bias gradient riding on the weight-gradient GEMM: colsum[m] = sum_k A[k][m]. The A fragments of the wc == 0 waves of
  // the tile_n == 0 workgroups already hold every A value once; v_dot2c_f32_bf16 against (1, 1) adds a fragment's 8 k values
  // in 4 VALU instructions that issue in the shadow of the MFMAs.
  const bool do_colsum = p.colsum != nullptr && tile_n == 0 && wc == 0;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int last = 4 * nk - 1;              // index of the last half-tile
  constexpr int AHEAD = 5;                  // half-tile s is issued in global phase s - AHEAD
  issue(std::integral_constant<int, 0>{}, 0);
  issue(std::integral_constant<int, 1>{}, 0);
  issue(std::integral_constant<int, 2>{}, 0);
  issue(std::integral_constant<int, 3>{}, 0);
  if (nk > 1) issue(std::integral_constant<int, 0>{}, 1);
  wait_halves((last < AHEAD - 1 ? last : AHEAD - 1) - 1);      // half-tiles 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger the second wave row by one barrier

  auto phase = [&](auto Pc, int t) {
    constexpr int P = decltype(Pc)::value;
    const char *base = smem + ((t & 1) << 2) * HALF;
    if constexpr (P == 0) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bq0[jj][ks] = frag_bf16_asm<TB, 128, 64>(base + 1 * HALF, wc * 32 + jj * 16, ks, lane);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[ii][ks] = frag_bf16_asm<TA, 128, 64>(base, wr * 64 + ii * 16, ks, lane);
    } else if constexpr (P == 1) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bq1[jj][ks] = frag_bf16_asm<TB, 128, 64>(base + 2 * HALF, wc * 32 + jj * 16, ks, lane);
    } else if constexpr (P == 2) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[ii][ks] = frag_bf16_asm<TA, 128, 64>(base + 3 * HALF, wr * 64 + ii * 16, ks, lane);
    }
    const int g = 4 * t + P;
    if (g + AHEAD <= last) issue(std::integral_constant<int, (P + AHEAD) & 3>{}, (g + AHEAD) >> 2);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) tie(af[ii][ks]);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { tie(bq0[jj][ks]); tie(bq1[jj][ks]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    constexpr int mh = (P >= 2) ? 1 : 0, nh = (P == 1 || P == 2) ? 1 : 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[mh * 4 + ii][nh * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, nh ? bq1[jj][ks] : bq0[jj][ks]), __builtin_bit_cast(bf16x8, af[ii][ks]), acc[mh * 4 + ii][nh * 2 + jj], 0, 0, 0);
    if constexpr (P == 0 || P == 2) {         // the phases that loaded a new A half
      if (do_colsum) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
        // (pairs taken with shufflevector: bit-casting the fragment to 4 dwords and indexing them made hipcc 7.2 feed
        // dword 0 to all four dot instructions)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 f = __builtin_bit_cast(bf16x8, af[ii][ks]);
            float c = csum[mh * 4 + ii];
            c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones, c, false);
            c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones, c, false);
            c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones, c, false);
            c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones, c, false);
            csum[mh * 4 + ii] = c;
          }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    {   // the counted wait sits after the MFMA cluster (the DMA has that much longer to land) and before the barrier that
        // publishes this wave's landed pieces to the readers of the next phase
      const int newest = g + AHEAD <= last ? g + AHEAD : last;
      const int rem = newest - (g + 2);
      wait_halves(rem < 0 ? 0 : rem);
    }
    __builtin_amdgcn_s_barrier();
  };

  for (int t = 0; t < nk; ++t) {
    phase(std::integral_constant<int, 0>{}, t);
    phase(std::integral_constant<int, 1>{}, t);
    phase(std::integral_constant<int, 2>{}, t);
    phase(std::integral_constant<int, 3>{}, t);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // balances the stagger barrier of the other wave row

  const int64_t coff = b0 * p.sC0 + b1 * p.sC1;
  epilogue<TC, EPI, 8, 4>(acc, p, coff, m0 + wr * 128 + (lane & 15), n0 + wc * 64 + (lane >> 4) * 4, true);
  if (do_colsum) {                              // lane (li, g) holds the k-group-g part of row li: fold the four groups
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
      float v = csum[rt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m0 + wr * 128 + rt * 16 + (lane & 15);
      if ((lane >> 4) == 0 && m < p.M) p.colsum[m] = p.colsum_acc ? p.colsum[m] + v : v;
    }
  }
}

template <typename TC, int EPI, bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  int tile_m, tile_n;
  map_tile(gridDim.x, blockIdx.x, p.tiles_m, tile_m, tile_n);
  gemm256_body<TC, EPI, TA, TB>(p, tile_m, tile_n, blockIdx.z);
}

// grouped weight gradients on the 256x256 ring: same problem / item tables as gemm_grouped_tn_kernel, 256x256 items
__global__ __launch_bounds__(512) void gemm256_grouped_tn_kernel(const GroupedProblem *__restrict__ probs, const GroupedItem *__restrict__ items) {
  const GroupedItem it = items[blockIdx.x];
  if (it.prob < 0) return;                       // padding of the per-XCD item lists (see the host-side ordering)
  const GroupedProblem g = probs[it.prob];
  GemmParams p;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.A = g.A; p.lda = g.lda; p.sA0 = 0; p.sA1 = 0;
  p.B = g.B; p.ldb = g.ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = g.C; p.c_dtype = EVP_F32; p.ldc = g.ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = g.accumulate; p.tiles_m = 0; p.splitk = 1; p.dbg = 0; p.dbgbuf = nullptr; p.colsum = nullptr; p.colsum_acc = 0;
  p.k_per_split = g.K;
  p.colsum = g.colsum; p.colsum_acc = g.colsum_accumulate;
  gemm256_body<float, 0, true, true>(p, it.tile_m, it.tile_n, 0);
}

template <typename TC, int EPI, bool TA, bool TB> int launch256(const evp_gemm_desc *d, hipStream_t s) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = d->strideA0; p.sA1 = d->strideA1;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = d->strideB0; p.sB1 = d->strideB1;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = d->strideC0; p.sC1 = d->strideC1;
  p.batch1 = d->batch1 > 0 ? d->batch1 : 1;
  p.alpha = d->alpha; p.bias = d->bias; p.act = d->act; p.aux = d->aux; p.ldaux = d->ldaux;
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate; p.dbg = g_gemm_dbg; p.dbgbuf = g_gemm_dbgbuf; p.colsum = nullptr; p.colsum_acc = 0;
  p.tiles_m = (d->M + 255) / 256;
  p.splitk = 1; p.k_per_split = d->K;
  const int tiles_n = (d->N + 255) / 256;
  const int nb = (d->batch0 > 0 ? d->batch0 : 1) * p.batch1;
  constexpr int smem = 8 * 16384;
  auto k = gemm256_kernel<TC, EPI, TA, TB>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { evp_set_error("evp_gemm: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e)); return EVP_ELAUNCH; }
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(p.tiles_m * tiles_n), 1, (unsigned)nb), dim3(512), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm");
  return EVP_OK;
}



// ---- "G4" weight-gradient body: 256x256 tile, 4 waves, ONE wave per SIMD, v_mfma_f32_32x32x16_bf16 ---------------------
// C[M][N] (f32) (+)= A^T . B, A stored [K][M], B stored [K][N] (TN), K % 32 == 0, K >= 96.
// Why a second 256x256 body: the half-tile ring above runs two waves per SIMD through barrier-separated read / MFMA phases
// and needs ~2.0-2.1 us per 64-deep K tile on the weight-gradient shapes. Here a wave owns 128x128 of the tile in 256
// accumulator registers (the whole 512-register budget belongs to it), which (a) needs 0.25 fragment reads per MFMA,
// (b) lets the wave hide its own LDS latency: the fragment reads of K step s+1 are issued in front of the 16 MFMAs of
// step s, no phase barriers -- ONE barrier per 32-deep stage. Both operands are k-strided, so stages can be 32 k-rows
// thin without splitting cache lines: A 16 KiB + B 16 KiB per stage, FOUR stages in a ring = three tiles of LDS-DMA in
// flight (measured: the DMA is hidden completely, tools/native/g4_gemm.hip `tn`: 1.25-1.3 us per 64-deep K tile on the
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
typedef __attribute__((ext_vector_type(16))) float f32x16;

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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mrow + 32 * i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = ncol + 32 * j + 8 * g;
        if (n >= p.N) continue;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        float *c = C + (int64_t)m * p.ldc + n;
        if (n + 3 < p.N) {
          if (p.accumulate) {
            const float4 o = *reinterpret_cast<const float4 *>(c);
            v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
          }
          *reinterpret_cast<float4 *>(c) = v;
        } else {
          const float e[4] = {v.x, v.y, v.z, v.w};
          for (int u = 0; u < 4 && n + u < p.N; ++u) c[u] = p.accumulate ? c[u] + e[u] : e[u];
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

__global__ __launch_bounds__(256) void gemm_g4_grouped_tn_kernel(const GroupedProblem *__restrict__ probs, const GroupedItem *__restrict__ items) {
  const GroupedItem it = items[blockIdx.x];
  if (it.prob < 0) return;                       // padding of the per-XCD item lists
  const GroupedProblem g = probs[it.prob];
  GemmParams p;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.A = g.A; p.lda = g.lda; p.sA0 = 0; p.sA1 = 0;
  p.B = g.B; p.ldb = g.ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = g.C; p.c_dtype = EVP_F32; p.ldc = g.ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = g.accumulate; p.tiles_m = 0; p.splitk = 1; p.dbg = 0; p.dbgbuf = nullptr;
  p.k_per_split = g.K;
  p.colsum = g.colsum; p.colsum_acc = g.colsum_accumulate;
  gemm_g4_tn_body(p, it.tile_m, it.tile_n);
}

__global__ __launch_bounds__(256) void gemm_g4_tn_kernel(const GemmParams p) {
  int tile_m, tile_n;
  map_tile(gridDim.x, blockIdx.x, p.tiles_m, tile_m, tile_n);
  gemm_g4_tn_body(p, tile_m, tile_n);
}

static int launch_g4_tn(const evp_gemm_desc *d, hipStream_t s) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = 0; p.sA1 = 0;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = d->accumulate; p.dbg = 0; p.dbgbuf = nullptr; p.colsum = nullptr; p.colsum_acc = 0;
  p.tiles_m = (d->M + 255) / 256;
  p.splitk = 1; p.k_per_split = d->K;
  const int tiles_n = (d->N + 255) / 256;
  constexpr int smem = 4 * 2 * 32 * 512;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_g4_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { evp_set_error("evp_gemm: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e)); return EVP_ELAUNCH; }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_g4_tn_kernel, dim3((unsigned)(p.tiles_m * tiles_n)), dim3(256), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm(g4 tn)");
  return EVP_OK;
}

// (Two persistent variants of the 128x128 body lived here in round 1 -- one workgroup per CU slot looping over tiles, the second
// with the C stores of tile i folded into the K loop of tile i+1. Measured: +3..8 % on the largest shapes, slower on the small
// ones (DESIGN.md section 4 "Epilogues"); never the default, removed in round 2.)

template <typename T, typename TC, int EPI, bool TA, bool TB> int pick_tile(const evp_gemm_desc *d, hipStream_t s) {
  int tile = d->tile;
  const int64_t nb = (int64_t)(d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
  if (tile == 0) {
    const int64_t t128 = (int64_t)((d->M + 127) / 128) * ((d->N + 127) / 128) * nb;
    const int64_t t256 = (int64_t)((d->M + 255) / 256) * ((d->N + 127) / 128) * nb;
    const bool splittable = d->transA && d->c_dtype == EVP_F32 && !d->bias && d->act == EVP_ACT_NONE && !d->residual &&
                            !d->aux && nb == 1 && d->K >= 2048;
    (void)t256;      // a 256x128 / 3-stage ring variant measured slower than 128x128 at this path's shapes (removed)
    tile = (d->M >= 128 && d->N >= 128 && (t128 >= 192 || splittable)) ? 1 : 2;
    // One round of 128 x 128 tiles that leaves some CUs with two workgroups and the rest with one (257..384 tiles) ends when the
    // doubly loaded CUs do; 96 x 128 tiles still fit one round (<= 512) and are 25 % smaller: -13..16 % on the 6272 x 768 outputs
    // of the encoder (294 -> 396 tiles; tools/gemm_tiles.py), bit-identical results (same k order per element).
    if constexpr (sizeof(T) == 2 && !TA) {
      const int64_t t96 = (int64_t)((d->M + 95) / 96) * ((d->N + 127) / 128) * nb;
      if (tile == 1 && g_gemm_variant != 2 && nb == 1 && t128 > 256 && t96 <= 512) tile = 4;
    }
  }
  // bf16: LDS-DMA staging (variant 1, default) or register staging (variant 2, kept for A/B runs); f32: registers
  if constexpr (sizeof(T) == 2 && !TA) {
    // stream-K is explicit only (tile 12): measured on this path's shapes it does not beat the data-parallel launch (see
    // the note at gemm_sk_kernel)
    if (d->tile == 12) {
      int g = 0;
      if (sk_usable<TC, EPI, TB>(d, &g)) return launch_sk<TC, EPI, TB>(d, s, g);
      evp_set_error("evp_gemm: tile 12 (stream-K) needs bf16 A row-major, no batch, K %% 64 == 0, K >= 256, >= 4 K tiles per workgroup and sk_workspace");
      return EVP_ESHAPE;
    }
  }
  if constexpr (sizeof(T) == 2 && TA) {
    if (d->tile == 12) { evp_set_error("evp_gemm: tile 12 (stream-K) is not built for transA"); return EVP_EUNSUPPORTED; }
  }
  if constexpr (sizeof(T) == 2) {
    if (tile == 7 || tile == 8) { evp_set_error("evp_gemm: tiles 7 / 8 (persistent variants) were removed"); return EVP_EUNSUPPORTED; }
    if (tile == 9) {        // G4 body (weight-gradient layout only): 256x256, one wave per SIMD, 32x32x16
      if constexpr (TA && TB && EPI == 0 && std::is_same<TC, float>::value) {
        const int64_t nbz = (int64_t)(d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
        if (nbz != 1 || d->K % 32 != 0 || d->K < 96 || d->bias || d->residual || d->aux || d->alpha != 1.0f || d->splitk > 1 ||
            d->lda % 8 != 0 || d->ldb % 8 != 0) {
          evp_set_error("evp_gemm: tile 9 (G4) needs transA, transB, f32 C, K %% 32 == 0, K >= 96, no batch / epilogue extras");
          return EVP_ESHAPE;
        }
        return launch_g4_tn(d, s);
      } else {
        evp_set_error("evp_gemm: tile 9 (G4) is built for the weight-gradient layout (transA = transB = 1, f32 C) only");
        return EVP_EUNSUPPORTED;
      }
    }
    if (tile == 6) {
      if (d->K % 64 != 0) { evp_set_error("evp_gemm: tile 6 (256x256 ring) needs K %% 64 == 0 (K=%d)", d->K); return EVP_ESHAPE; }
      return launch256<TC, EPI, TA, TB>(d, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_gemm_variant != 2) {
      // 96 x 128: for launches whose 128 x 128 tiling leaves most CUs with one workgroup and a few with two (294 tiles of a
      // 6272 x 768 output -> 396 tiles): the launch ends when the doubly loaded CUs do, and their tiles are 25 % smaller
      if (tile == 4) return launch<T, TC, EPI, TA, TB, 96, 128, 2, 2, true, 2>(d, s);
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
                     reinterpret_cast<const GroupedProblem *>(problems), reinterpret_cast<const GroupedItem *>(items));
  EVP_CHECK_LAUNCH("evp_gemm_grouped_tn_bf16");
  return EVP_OK;
}

extern "C" int evp_gemm_grouped_tn256_bf16(const void *problems, const void *items, int n_items, void *stream) {
  EVP_CHECK_ARG(problems && items && n_items > 0, EVP_EINVAL, "evp_gemm_grouped_tn256_bf16: bad argument");
  auto k = gemm256_grouped_tn_kernel;
  constexpr int smem = 8 * 16384;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_gemm_grouped_tn256_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)n_items), dim3(512), smem, (hipStream_t)stream,
                     reinterpret_cast<const GroupedProblem *>(problems), reinterpret_cast<const GroupedItem *>(items));
  EVP_CHECK_LAUNCH("evp_gemm_grouped_tn256_bf16");
  return EVP_OK;
}

extern "C" int evp_gemm_grouped_tn_g4_bf16(const void *problems, const void *items, int n_items, void *stream) {
  EVP_CHECK_ARG(problems && items && n_items > 0, EVP_EINVAL, "evp_gemm_grouped_tn_g4_bf16: bad argument");
  auto k = gemm_g4_grouped_tn_kernel;
  constexpr int smem = 4 * 2 * 32 * 512;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_gemm_grouped_tn_g4_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)n_items), dim3(256), smem, (hipStream_t)stream,
                     reinterpret_cast<const GroupedProblem *>(problems), reinterpret_cast<const GroupedItem *>(items));
  EVP_CHECK_LAUNCH("evp_gemm_grouped_tn_g4_bf16");
  return EVP_OK;
}

extern "C" int evp_gemm_set_debug_buffer(void *buf) {   // measurement aid: uint64 [4 * 512]; NULL switches it off
  g_gemm_dbgbuf = reinterpret_cast<unsigned long long *>(buf);
  return EVP_OK;
}

extern "C" int evp_gemm_set_variant(int v) {
  const int old = g_gemm_variant;
  if (v >= 100 && v <= 102) g_gemm_dbg = v - 100;
  if (v == 1 || v == 2) g_gemm_variant = v;
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
