// Shared by the GEMM translation units of libevtpretrain.so (gemm.hip: 128x128-class bodies and the dispatcher;
// gemm_g4.hip: the one-wave-per-SIMD 32x32x16 "G4" bodies). Internal header, not part of the C ABI.
#pragma once
#include "evp_common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

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
  float *colsum;             // G4 TN body only: colsum[m] (+)= sum_k A[k][m] (bias gradient), written by the tile_n == 0 workgroups
  int colsum_acc;
  int c_wt16;                // bf16 C (and aux) below 2 GiB: the LDS-staged epilogue may use 16-byte write-through buffer stores
  unsigned long long *stamp; // measurement aid (evp_gemm_set_stamp_buffer): [2 * workgroup] start / end wall-clock stamps of this launch
};

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
// Write-through stores (agent-scope relaxed atomic store = `global_store ... sc1`) for 8-byte-per-lane output pieces. A kernel's plain
// stores stay dirty in the eight XCDs' L2s until the end-of-kernel release writes them back (MI355X_MICROARCH.md price list: a kernel
// boundary costs + bytes / 6 TB/s for what the predecessor leaves dirty; "tens of KB per workgroup from a 4-8-byte-per-lane epilogue:
// write-through wins"); written through, they drain while the kernel is still computing. Round 3, same-box A/B of the ViT-Base step
// (2 x 2 interleaved runs): bf16 GEMM outputs written through 11.14 against 11.36-11.43 ms plain (-2 %); agent and system scope the
// same; the f32 outputs as two 8-byte write-through stores instead of one 16-byte plain store LOSE (11.68-11.77 against 11.52-11.62), as
// one `buffer_store_dwordx4 ... sc1` they change nothing (11.145 against 11.139), nor do non-temporal stores, nor write-through outputs of
// the attention and LayerNorm kernels (11.37-11.43 either way). EVP_WT_STORES=0 builds the plain form for A/B.
#ifndef EVP_WT_STORES
#define EVP_WT_STORES 1
#endif
__device__ __forceinline__ void st_u64_wt(void *p, unsigned long long v) {
#if EVP_WT_STORES
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *reinterpret_cast<unsigned long long *>(p) = v;
#endif
}
template <> __device__ __forceinline__ void st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// eight bf16 (16 bytes) written through with ONE instruction (`buffer_store_dwordx4 ... sc1`; an 8-byte `sc1` store costs 2.7x the time
// per byte of a 16-byte one, MI355X_MICROARCH.md "stores of each flavour"); `base` wave-uniform, byte offset below 2 GiB
typedef unsigned __attribute__((ext_vector_type(4))) st_u32x4;
__device__ __forceinline__ void st8_bf16_wt(bf16_t *base, int64_t elem_off, float4 a, float4 b) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
  st_u32x4 u;
  u.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
  u.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
  u.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
  u.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(elem_off * 2), 0, 16);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, float4 v) {
  const uint32_t lo = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
  const uint32_t hi = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
  st_u64_wt(p, (unsigned long long)lo | ((unsigned long long)hi << 32));
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

// ---- grouped weight-gradient GEMM: many independent (dY^T . X) problems in ONE launch ------------------------------
// problem g: C_g[M_g, N_g] (f32) = A_g^T . B_g with A_g stored [K_g][M_g], B_g stored [K_g][N_g] (bf16). Work item =
// one 128x128 output tile of one problem; items are listed largest-K first so the long tiles start early.
struct GroupedProblem {
  const void *A, *B;
  void *C;
  int M, N, K;
  int lda, ldb, ldc;
  int accumulate, colsum_accumulate;
  float *colsum;             // G4 kernel only: colsum[m] (+)= sum_k A[k][m]; NULL = none
};
struct GroupedItem { int prob, tile_m, tile_n, pad; };

// ---- in-kernel wall-clock stamps (measurement aid, off by default) -------------------------------------------------------
// With a stamp buffer installed (evp_gemm_set_stamp_buffer) every GEMM launch gets the next slot of EVP_STAMP_WGS x 2 uint64 and each
// of its workgroups writes s_memrealtime (the chip-wide 100 MHz counter) when it starts and after its last store has been
// acknowledged. max(end) - min(start) over a launch is its duration as it ran IN PLACE -- inside a replayed HIP graph, operands
// just produced by the previous kernel -- which HIP events around re-launches cannot see (bench.py `roofline`).
constexpr int EVP_STAMP_WGS = 4096;
// start stamps are stored complemented (so that 0 = "not written" and a max() keeps the EARLIEST start); grids larger than the slot
// fold onto it with atomic max, which keeps the earliest start and the latest end per entry
__device__ __forceinline__ void stamp_begin(unsigned long long *stamp, unsigned wg, unsigned nwg) {
  if (stamp && threadIdx.x == 0) {
    const unsigned long long t = ~(unsigned long long)__builtin_amdgcn_s_memrealtime();
    if (nwg <= (unsigned)EVP_STAMP_WGS) stamp[2 * wg] = t;
    else atomicMax(&stamp[2 * (wg % EVP_STAMP_WGS)], t);
  }
}
__device__ __forceinline__ void stamp_end(unsigned long long *stamp, unsigned wg, unsigned nwg) {
  if (stamp) {                                     // wave-uniform: a kernel argument
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = (unsigned long long)__builtin_amdgcn_s_memrealtime();
      if (nwg <= (unsigned)EVP_STAMP_WGS) stamp[2 * wg + 1] = t;
      else atomicMax(&stamp[2 * (wg % EVP_STAMP_WGS) + 1], t);
    }
  }
}

}  // namespace

// next stamp slot for a launch, or NULL when stamping is off (defined in gemm.hip)
unsigned long long *evp_gemm_next_stamp_slot();
// gemm_g4.hip: the G4 forward / data-gradient kernels behind evp_gemm (tiles 20-22) and the single-problem G4 TN launch (tile 9)
bool evp_g4_gemm_supported(const evp_gemm_desc *d);
int evp_g4_gemm_pick(const evp_gemm_desc *d);
int evp_g4_gemm(const evp_gemm_desc *d, hipStream_t s, int shape, int dbg);
int evp_g4_gemm_tn(const evp_gemm_desc *d, hipStream_t s);
