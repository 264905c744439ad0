// Fused multi-head attention core for short sequences (N <= 224 tokens, d_h = 32 / 64), bf16, gfx950.
//
// One workgroup (4 waves) per (batch, head). Q, K, V (and dO in the backward) of the head live in LDS for the whole
// kernel; the N x N score matrix never leaves registers:
//   * a 16 x 16 MFMA result tile keeps its COLUMN on the lane and its 4 ROWS in registers, so a second MFMA that sums
//     over the first one's row index takes the (bf16-packed) accumulators directly as its operand -- the k order inside a
//     32-deep step is permuted (key = 32s + 16(j>>2) + 4g + (j&3)), and the other operand is fetched in that same order
//     with ds_read_b64_tr_b16 (hardware transpose read) from the untransposed [token][d_h] LDS image.
//   * forward : "query on the lane": S = K q^T tile -> softmax over the 4 lanes x registers holding a row -> O = V^T P.
//   * backward: a query-on-lane pass gives dQ (sums over keys), a key-on-lane pass gives dK and dV (sums over queries);
//     both recompute P from Q, K and the saved log-sum-exp, so nothing is reduced across waves and there are no atomics.
// Replaces model/sub_module/vit_block.py:134-140 (scores, softmax, probs @ v) and its autograd backward.
#include "evp_common.h"

#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;

static unsigned long long *g_attn_dbg = nullptr;   // measurement aid, see evp_attention_set_debug_buffer

namespace {

// ---- LDS image of a [rows][DH] bf16 array: 16-byte chunks, XOR-swizzled for 128-byte rows ---------------------------
// d_h = 32 (64-byte rows, four rows per 256-byte bank row): unswizzled, every ds_read_b128 fragment read and every transposed
// read met a second row on the same banks -- rocprofv3 counted 42 % of the LDS cycles of the decoder's backward as bank
// conflicts (profiles/r02_pmc_sq_kernels.json). The chunk index is XORed with g((row >> 2) & 3), g = {0, 2, 3, 1}: the four
// 16-lane groups of a ds_read_b128 (rows 0-3 / 12-15 at chunk c with rows 4-11 at chunk c+1, and so on) and the 8 rows x 32 B
// of a ds_read_b64_tr_b16 half-wave then fall on 16 distinct 16-byte slots.
template <int DH> __device__ __forceinline__ int img_off(int row, int chunk) {
  if (DH == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  const int x = (row >> 2) & 3;
  const int g = (((x >> 1) ^ x) & 1) << 1 | (x >> 1);
  return row * 64 + ((chunk ^ g) << 4);
}
// fragment with the 16 rows [rb, rb+16) on the lanes and 8 consecutive d (k-step ks of 32) per lane: ds_read_b128
template <int DH> __device__ __forceinline__ bf16x8 frag_rows(const char *img, int rb, int ks, int lane) {
  const uint4 v = *reinterpret_cast<const uint4 *>(img + img_off<DH>(rb + (lane & 15), ks * 4 + (lane >> 4)));
  return __builtin_bit_cast(bf16x8, v);
}
// fragment with the 16 columns d in [d0, d0+16) on the lanes and the permuted rows of step s:
// element j of lane group g = row 32s + 16(j>>2) + 4g + (j&3)   (two transposed reads of 4 rows x 16 columns)
template <int DH> __device__ __forceinline__ bf16x8 frag_cols(const char *img, int d0, int s, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int col = d0 + 4 * p;
  const int r0 = 32 * s + 4 * g + q;
  const char *a0 = img + img_off<DH>(r0, col >> 3) + (col & 7) * 2;
  const char *a1 = img + img_off<DH>(r0 + 16, col >> 3) + (col & 7) * 2;
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)a0);
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)a1);
  i16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}
// two f32 accumulator tiles (rows 16(2s) + 4g + r and 16(2s+1) + 4g + r) -> the bf16 operand of step s
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 &a, const f32x4 &b) {
  bf16x8 v;
  v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
  v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
  return v;
}
__device__ __forceinline__ uint2 pack4(const f32x4 &a) {
  uint2 u;
  u.x = (uint32_t)f32_to_bf16(a[0]) | ((uint32_t)f32_to_bf16(a[1]) << 16);
  u.y = (uint32_t)f32_to_bf16(a[2]) | ((uint32_t)f32_to_bf16(a[3]) << 16);
  return u;
}

// copy the [N][DH] slice of one head (token stride `tok` elements) into an LDS image with NP rows (zero padded)
template <int DH>
__device__ __forceinline__ void stage_head(const bf16_t *src, int64_t tok, int N, int NP, char *img, int tid, int nthreads) {
  constexpr int CPR = DH / 8;
  for (int c = tid; c < NP * CPR; c += nthreads) {
    const int row = c / CPR, ch = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *reinterpret_cast<const uint4 *>(src + (int64_t)row * tok + ch * 8);
    *reinterpret_cast<uint4 *>(img + img_off<DH>(row, ch)) = v;
  }
}

// Batched form: every thread first ISSUES all its 16-byte loads of all NA arrays (NA * PER independent loads in
// flight), then parks them in LDS. The loop form above costs one exposed HBM round trip per iteration and array --
// about 12 (forward) / 25 (backward) serial round trips per workgroup before the first MFMA.
template <int DH, int NP, int NA, int NTHR = 256> struct HeadStage {
  static constexpr int CPR = DH / 8, TOTAL = NP * CPR, PER = (TOTAL + NTHR - 1) / NTHR;
  uint4 r[NA][PER];
  __device__ __forceinline__ void load(const bf16_t *const (&src)[NA], const int64_t (&tok)[NA], int N, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NTHR, row = c / CPR, ch = c % CPR;
      const bool ok = (TOTAL % NTHR == 0 || c < TOTAL) && row < N;
#pragma unroll
      for (int a = 0; a < NA; ++a)
        r[a][i] = ok ? *reinterpret_cast<const uint4 *>(src[a] + (int64_t)row * tok[a] + ch * 8) : make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void store(int a, char *img, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NTHR, row = c / CPR, ch = c % CPR;
      if (TOTAL % NTHR == 0 || c < TOTAL) *reinterpret_cast<uint4 *>(img + img_off<DH>(row, ch)) = r[a][i];
    }
  }
};

// Workgroup -> (batch, head). With d_h = 32 a 128-byte line of qkv / out holds the slices of TWO adjacent heads, and
// workgroups i and i+8 run on the same XCD (= share an L2): give that pair the two heads of one line, so the line is
// fetched from HBM once instead of once per XCD. Pure renumbering (bijective when B*heads is a multiple of 16).
template <int DH> __device__ __forceinline__ void head_of_block(int bid, int nblk, int heads, int &b, int &h) {
  int L = bid;
  if (DH == 32 && (nblk & 15) == 0 && (heads & 1) == 0) L = (bid & ~15) + 2 * (bid & 7) + ((bid >> 3) & 1);
  b = L / heads;
  h = L - b * heads;
}

// ---------------------------------------------------------------------------------------------------- forward
// NT = number of 16-wide score tiles (NP = 16*NT rows in LDS, NT even)
// BIAS (windowed attention, model/sub_module/swin_block.py:135-158): logits = q.k*scale + addm[group][head][query][key], where
// addm holds the gathered relative-position bias, or -100 for the pairs the reference masks (tokens of different windows, padding
// slots); group = batch index % nG; addm is [nG][heads][NP][NP] f32 (evp_window_bias_build).
// The kernel is built from three pieces so that a persistent workgroup can keep the NEXT head's loads in flight while it computes
// the current one: fwd_stage_load (global -> registers), fwd_stage_store (registers -> LDS images), attn_fwd_compute.
template <int DH, int NT, int NW> struct FwdStage {
  static constexpr int NP = 16 * NT, IMG = NP * DH * 2;
  HeadStage<DH, NP, 3, 64 * NW> st;
  __device__ __forceinline__ void load(const bf16_t *qkv, int N, int heads, int b, int h, int tid) {
    const int64_t C = (int64_t)heads * DH, tok = 3 * C;
    const bf16_t *base = qkv + (int64_t)b * N * tok + (int64_t)h * DH;
    const bf16_t *const src[3] = {base, base + C, base + 2 * C};
    const int64_t toks[3] = {tok, tok, tok};
    st.load(src, toks, N, tid);
  }
  __device__ __forceinline__ void store(char *smem, int tid) const {
    st.store(0, smem, tid);
    st.store(1, smem + IMG, tid);
    st.store(2, smem + 2 * IMG, tid);
  }
};

template <int DH, int NT, int NW, bool BIAS>
__device__ __forceinline__ void attn_fwd_compute(char *smem, bf16_t *out, float *lse, bf16_t *probs, int N, int heads, float scale, int64_t ldp,
                                                 const float *addm, int nG, const int b, const int h) {
  constexpr int NP = 16 * NT, IMG = NP * DH * 2, KS = DH / 32, DT = DH / 16;
  char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = b * heads + h;
  const int64_t C = (int64_t)heads * DH;
  const float c2 = scale * 1.44269504088896340736f;   // exp(x*scale) = exp2(x*c2)

  for (int strip = wave; strip * 16 < N; strip += NW) {
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_rows<DH>(Qs, strip * 16, ks, lane);
    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Ks, t * 16, ks, lane), qf[ks], S[t], 0, 0, 0);
    }
    // lane: query strip*16 + li; registers: keys 16t + 4g + r. The softmax is the VALU-bound part of this kernel
    // (56 scores per lane and strip at N = 196), so it is written on whole f32x4 tiles (v_pk_* packed math), with the
    // raw v_exp_f32 (arguments are <= 0: no overflow, underflow to 0 is the wanted result) and with the key mask applied
    // only to the tiles that reach past N.
    f32x4 mx4 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if constexpr (BIAS) {
      // logits in log2 units: X = S*scale*log2(e) + addm*log2(e); rows of addm are NP floats, this lane's query row
      const float *arow = addm + ((int64_t)((b % nG) * heads + h) * NP + (strip * 16 + li)) * NP + 4 * g;
      const f32x4 c2b = f32x4{c2, c2, c2, c2};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * t);
        const f32x4 al = f32x4{a.x * 1.44269504088896340736f, a.y * 1.44269504088896340736f, a.z * 1.44269504088896340736f,
                               a.w * 1.44269504088896340736f};
        S[t] = __builtin_elementwise_fma(S[t], c2b, al);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (16 * t + 15 >= N) {               // wave-uniform: only the last one or two tiles
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * g + r >= N) S[t][r] = -INFINITY;
      }
      mx4 = __builtin_elementwise_max(mx4, S[t]);
    }
    float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = BIAS ? -mx : -mx * c2;
    const float cm = BIAS ? 1.0f : c2;
    const f32x4 c2v = f32x4{cm, cm, cm, cm}, nmv = f32x4{nm, nm, nm, nm};
    f32x4 sum4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 e = __builtin_elementwise_fma(S[t], c2v, nmv);
      S[t] = f32x4{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1]), __builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])};
      sum4 += S[t];
    }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const int q = strip * 16 + li;
    if (g == 0 && q < N && lse) lse[(int64_t)bh * N + q] = (BIAS ? mx * 0.69314718055994530942f : mx * scale) + logf(sum);
#pragma unroll
    for (int t = 0; t < NT; ++t) S[t] = S[t] * inv;
    if (probs && q < N) {
      bf16_t *pr = probs + ((int64_t)bh * N + q) * ldp;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (16 * t + 4 * g < ldp) *reinterpret_cast<uint2 *>(pr + 16 * t + 4 * g) = pack4(S[t]);
    }
    // O^T tile [d][query] = sum_key V[key][d] P[query][key]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 O = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NT / 2; ++s)
        O = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Vs, dt * 16, s, lane), pack_tiles(S[2 * s], S[2 * s + 1]), O, 0, 0, 0);
      if (q < N) *reinterpret_cast<uint2 *>(out + ((int64_t)b * N + q) * C + h * DH + dt * 16 + 4 * g) = pack4(O);
    }
  }
}

template <int DH, int NT, int NW, bool BIAS = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const bf16_t *qkv, bf16_t *out, float *lse, bf16_t *probs, int N, int heads,
                                                       float scale, int64_t ldp, unsigned long long *dbg, const float *addm,
                                                       int nG) {
  unsigned long long t0 = 0, t1 = 0;
  if (dbg) t0 = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int b, h;
  head_of_block<DH>(blockIdx.x, gridDim.x, heads, b, h);
  {
    FwdStage<DH, NT, NW> fs;
    fs.load(qkv, N, heads, b, h, threadIdx.x);
    fs.store(smem, threadIdx.x);
  }
  __syncthreads();
  if (dbg) t1 = __builtin_readcyclecounter();
  attn_fwd_compute<DH, NT, NW, BIAS>(smem, out, lse, probs, N, heads, scale, ldp, addm, nG, b, h);
  if (dbg && (threadIdx.x & 63) == 0) {
    const unsigned long long t2 = __builtin_readcyclecounter();
    if ((threadIdx.x >> 6) < 4) {
      dbg[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = t1 - t0;     // staging
      dbg[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = t2 - t1;     // strips of this wave
    }
  }
}

// (Round 4, measured and removed again: PERSISTENT forms of both kernels -- a fixed grid of 256-1024 workgroups walking the (batch,
//  head) items, optionally with the second half of the grid starting 2-4 us late so that one workgroup of a CU stages while the other
//  computes, optionally with the next item's loads in flight (registers) under the current item's strips; the kernels above are
//  split into stage / compute pieces for that. Bit-identical results, and slower in every form: the step's launches hold 768 / 1024
//  items for 512-768 resident workgroups, i.e. 1-2 items per workgroup -- there is no steady state for a phase offset to pay in, the
//  item loop's hoisted addresses cost 20-40 registers (a resident workgroup of the backward, or spills at the old occupancy), and the
//  prefetch registers halve the occupancy outright. Re-launched: encoder forward 11.0 -> 11.0-19 us, backward 26.5 -> 29-38;
//  decoder forward 27.2 -> 24.8-37, backward 52.2 -> 52.6-71; in the replayed ViT-Base step 10.85-11.18 against 10.67 ms
//  (profiles/r04_attn_variants.txt, r04_ab_attn.txt).)

// ---------------------------------------------------------------------------------------------------- backward
// BIAS: addm as in the forward, addmT its transpose per (group, head) ([key][query]: the key-on-lane pass reads 4 consecutive
// queries); the d logits of this wave's query strip (one strip per wave: NW >= number of strips) are ADDED into dacc, which the
// windowed wrapper keeps across the batch items it walks and flushes once.
template <int DH, int NT, int NW> struct BwdStage {
  // Q, K, V, dO and O in one batch of loads; O is only needed for delta[q] = sum_d dO[q,d] * O[q,d], formed from the
  // staged registers (the DH/8 consecutive threads that share a row meet with shuffles)
  static constexpr int NP = 16 * NT, IMG = NP * DH * 2, NTHR = 64 * NW, NL = (NP + NTHR - 1) / NTHR;
  using ST = HeadStage<DH, NP, 5, NTHR>;
  ST st;
  float lv[NL];
  __device__ __forceinline__ void load(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse, int N, int heads, int b, int h,
                                       int tid) {
    const int bh = b * heads + h;
    const int64_t C = (int64_t)heads * DH, tok = 3 * C;
    const bf16_t *base = qkv + (int64_t)b * N * tok + (int64_t)h * DH;
    const bf16_t *go = dout + (int64_t)b * N * C + (int64_t)h * DH, *oo = out + (int64_t)b * N * C + (int64_t)h * DH;
    const bf16_t *const src[5] = {base, base + C, base + 2 * C, go, oo};
    const int64_t toks[5] = {tok, tok, tok, C, C};
    st.load(src, toks, N, tid);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = tid + i * NTHR;
      lv[i] = (r < N) ? lse[(int64_t)bh * N + r] : INFINITY;      // exp(-inf) = 0 on the padded queries
    }
  }
  __device__ __forceinline__ void store(char *smem, int tid) const {
    char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG, *Gs = smem + 3 * IMG;   // Gs = dO
    float *Ls = reinterpret_cast<float *>(smem + 4 * IMG), *Ds = Ls + NP;          // log-sum-exp, delta = rowsum(dO * O)
    st.store(0, Qs, tid);
    st.store(1, Ks, tid);
    st.store(2, Vs, tid);
    st.store(3, Gs, tid);
#pragma unroll
    for (int i = 0; i < ST::PER; ++i) {
      const int c = tid + i * NTHR, row = c / ST::CPR, ch = c % ST::CPR;
      const uint32_t aw[4] = {st.r[3][i].x, st.r[3][i].y, st.r[3][i].z, st.r[3][i].w};
      const uint32_t ow[4] = {st.r[4][i].x, st.r[4][i].y, st.r[4][i].z, st.r[4][i].w};
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        d += __uint_as_float(aw[e] << 16) * __uint_as_float(ow[e] << 16) +
             __uint_as_float(aw[e] & 0xFFFF0000u) * __uint_as_float(ow[e] & 0xFFFF0000u);
#pragma unroll
      for (int o = 1; o < ST::CPR; o <<= 1) d += __shfl_xor(d, o, 64);
      if (ch == 0 && (ST::TOTAL % NTHR == 0 || c < ST::TOTAL)) Ds[row] = d;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = tid + i * NTHR;
      if (r < NP) Ls[r] = lv[i];
    }
  }
};

#ifndef ATTN_CH
#define ATTN_CH 4
#endif

template <int DH, int NT, int NW, bool BIAS>
__device__ __forceinline__ void attn_bwd_compute(char *smem, bf16_t *dqkv, int N, int heads, float scale, const float *addm, const float *addmT, int nG,
                                                 const int b, const int h, f32x4 (&dacc)[NT], unsigned long long *t2p = nullptr) {
  constexpr int NP = 16 * NT, IMG = NP * DH * 2, KS = DH / 32, DT = DH / 16, CH = ATTN_CH;
  char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG, *Gs = smem + 3 * IMG;   // Gs = dO
  float *Ls = reinterpret_cast<float *>(smem + 4 * IMG), *Ds = Ls + NP;          // log-sum-exp, delta = rowsum(dO * O)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int64_t C = (int64_t)heads * DH, tok = 3 * C;
  const float c2 = scale * 1.44269504088896340736f, l2e = 1.44269504088896340736f;

  // ---- pass 1, query on the lane: dQ ----
  for (int strip = wave; strip * 16 < N; strip += NW) {
    bf16x8 qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = frag_rows<DH>(Qs, strip * 16, ks, lane);
      gf[ks] = frag_rows<DH>(Gs, strip * 16, ks, lane);
    }
    const float lq = Ls[strip * 16 + li] * l2e, dq_ = Ds[strip * 16 + li];
    const int q = strip * 16 + li;
    const f32x4 c2q = f32x4{c2, c2, c2, c2}, nlq = f32x4{-lq, -lq, -lq, -lq}, dqv = f32x4{dq_, dq_, dq_, dq_}, scv = f32x4{scale, scale, scale, scale};
    f32x4 accq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t arow = BIAS ? ((int64_t)((b % nG) * heads + h) * NP + q) * NP + 4 * g : 0;
    // key tiles in chunks of CH (even): only CH score tiles are live at a time, dQ accumulates across chunks
    auto chunk = [&](const int c0) {
      f32x4 P[CH];
#pragma unroll
      for (int tt = 0; tt < CH; ++tt) {
        const int t = c0 + tt;
        if (t < NT) {
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Ks, t * 16, ks, lane), qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Vs, t * 16, ks, lane), gf[ks], dp, 0, 0, 0);
          }
          // dS = P * (dP - delta) * scale on the whole tile (packed f32 math, raw v_exp_f32); keys past N only exist in
          // the last one or two tiles
          f32x4 pv;
          if constexpr (BIAS) {
            const float4 a = *reinterpret_cast<const float4 *>(addm + arow + 16 * t);
            const f32x4 al = f32x4{a.x * l2e, a.y * l2e, a.z * l2e, a.w * l2e};
            pv = __builtin_elementwise_fma(s, c2q, al + nlq);
          } else {
            pv = __builtin_elementwise_fma(s, c2q, nlq);
          }
          pv = f32x4{__builtin_amdgcn_exp2f(pv[0]), __builtin_amdgcn_exp2f(pv[1]), __builtin_amdgcn_exp2f(pv[2]), __builtin_amdgcn_exp2f(pv[3])};
          if (16 * t + 15 >= N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * t + 4 * g + r >= N) pv[r] = 0.f;
          }
          if constexpr (BIAS) dacc[t] += pv * (dp - dqv);      // d logits (unscaled) of this (query, 4 keys) piece
          P[tt] = pv * ((dp - dqv) * scv);
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ss = 0; ss < CH / 2; ++ss)
          if (c0 + 2 * ss < NT)
            accq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Ks, dt * 16, (c0 >> 1) + ss, lane),
                                                               pack_tiles(P[2 * ss], P[2 * ss + 1]), accq[dt], 0, 0, 0);
    };
    if constexpr (BIAS) {         // unrolled: dacc[] is indexed by the tile number
#pragma unroll
      for (int c0 = 0; c0 < NT; c0 += CH) chunk(c0);
    } else {
#pragma unroll 1
      for (int c0 = 0; c0 < NT; c0 += CH) chunk(c0);
    }
    if (q < N) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<uint2 *>(dqkv + ((int64_t)b * N + q) * tok + h * DH + dt * 16 + 4 * g) = pack4(accq[dt]);
    }
  }

  if (t2p) *t2p = __builtin_readcyclecounter();
  // ---- pass 2, key on the lane: dK, dV ---- (strips dealt to the waves in the opposite order of pass 1: with 13 strips
  // the wave that took four in pass 1 takes three here)
  for (int strip = NW - 1 - wave; strip * 16 < N; strip += NW) {
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = frag_rows<DH>(Ks, strip * 16, ks, lane);
      vf[ks] = frag_rows<DH>(Vs, strip * 16, ks, lane);
    }
    f32x4 av[DT], ak[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) av[dt] = ak[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c0 = 0; c0 < NT; c0 += CH) {      // query tiles in chunks: dK / dV accumulate across chunks
      f32x4 P[CH], dS[CH];
#pragma unroll
      for (int tt = 0; tt < CH; ++tt) {
        const int t = c0 + tt;
        if (t < NT) {
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Qs, t * 16, ks, lane), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Gs, t * 16, ks, lane), vf[ks], dp, 0, 0, 0);
          }
          const float4 lq = *reinterpret_cast<const float4 *>(Ls + 16 * t + 4 * g);
          const float4 dl = *reinterpret_cast<const float4 *>(Ds + 16 * t + 4 * g);
          const f32x4 nl = f32x4{-lq.x * l2e, -lq.y * l2e, -lq.z * l2e, -lq.w * l2e}, dlv = f32x4{dl.x, dl.y, dl.z, dl.w};
          const f32x4 c2k = f32x4{c2, c2, c2, c2}, sck = f32x4{scale, scale, scale, scale};
          f32x4 addl = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (BIAS) {
            const float4 a = *reinterpret_cast<const float4 *>(addmT + ((int64_t)((b % nG) * heads + h) * NP + (strip * 16 + li)) * NP + 16 * t + 4 * g);
            addl = f32x4{a.x * l2e, a.y * l2e, a.z * l2e, a.w * l2e};
          }
          f32x4 pv = __builtin_elementwise_fma(s, c2k, nl + addl);        // padded queries carry lse = +inf -> p = 0
          pv = f32x4{__builtin_amdgcn_exp2f(pv[0]), __builtin_amdgcn_exp2f(pv[1]), __builtin_amdgcn_exp2f(pv[2]), __builtin_amdgcn_exp2f(pv[3])};
          P[tt] = pv;
          dS[tt] = pv * ((dp - dlv) * sck);
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ss = 0; ss < CH / 2; ++ss)
          if (c0 + 2 * ss < NT) {
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Gs, dt * 16, (c0 >> 1) + ss, lane),
                                                             pack_tiles(P[2 * ss], P[2 * ss + 1]), av[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Qs, dt * 16, (c0 >> 1) + ss, lane),
                                                             pack_tiles(dS[2 * ss], dS[2 * ss + 1]), ak[dt], 0, 0, 0);
          }
    }
    const int key = strip * 16 + li;
    if (key < N) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16_t *o = dqkv + ((int64_t)b * N + key) * tok + h * DH + dt * 16 + 4 * g;
        *reinterpret_cast<uint2 *>(o + C) = pack4(ak[dt]);
        *reinterpret_cast<uint2 *>(o + 2 * C) = pack4(av[dt]);
      }
    }
  }
}

template <int DH, int NT, int NW, bool BIAS>
__device__ __forceinline__ void attn_bwd_body(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse, bf16_t *dqkv, int N,
                                              int heads, float scale, const float *addm, const float *addmT, int nG, const int b, const int h,
                                              f32x4 (&dacc)[NT], unsigned long long *dbg = nullptr) {
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (dbg) t0 = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  {
    BwdStage<DH, NT, NW> bs;
    bs.load(qkv, out, dout, lse, N, heads, b, h, threadIdx.x);
    bs.store(smem, threadIdx.x);
  }
  __syncthreads();
  if (dbg) t1 = __builtin_readcyclecounter();
  attn_bwd_compute<DH, NT, NW, BIAS>(smem, dqkv, N, heads, scale, addm, addmT, nG, b, h, dacc, dbg ? &t2 : nullptr);
  if (dbg && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) {      // measurement aid: per wave {staging, pass 1, pass 2} cycles
    const unsigned long long t3 = __builtin_readcyclecounter();
    unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 3;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2;
  }
}

template <int DH, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse,
                                                       bf16_t *dqkv, int N, int heads, float scale, unsigned long long *dbg) {
  int b, h;
  head_of_block<DH>(blockIdx.x, gridDim.x, heads, b, h);
  f32x4 unused[NT];
  attn_bwd_body<DH, NT, NW, false>(qkv, out, dout, lse, dqkv, N, heads, scale, nullptr, nullptr, 1, b, h, unused, dbg);
}

// Windowed form: workgroup = (group, head, batch chunk). It walks `per` batch items of its (group, head), keeping the d logits of
// its query strips in registers, and adds them into dA once at the end -- B / per adds per element instead of B (one plain store
// when a single workgroup covers the whole batch). One workgroup per (batch, head) adding every tile straight into dA measured
// 246 us per Swin-T stage-1 launch: 64 workgroups contending for each row of the same plane.
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void win_attn_bwd_mfma_kernel(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse,
                                                                bf16_t *dqkv, int N, int heads, float scale, const float *addm,
                                                                const float *addmT, float *dA, int nG, int B, int nchunk, int per) {
  constexpr int NP = 16 * NT;
  const int gh = blockIdx.x / nchunk, c = blockIdx.x - gh * nchunk;
  const int grp = gh / heads, h = gh - grp * heads;
  f32x4 dacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int rep = 0; rep < per; ++rep) {
    const int bi = c * per + rep;
    if (bi >= B) break;
    attn_bwd_body<32, NT, NW, true>(qkv, out, dout, lse, dqkv, N, heads, scale, addm, addmT, nG, bi * nG + grp, h, dacc);
    __syncthreads();                       // the next batch item re-stages the LDS images
  }
  // every chunk owns a plane set dA[c][g][h][NP][NP]: plain stores (evp_window_bias_reduce sums the chunks). With atomics into one
  // plane set, 32 workgroups per plane contended at Swin stage 3 (nG = 1): 101 us per launch, most of it the adds.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, q = wave * 16 + (lane & 15);
  if (q < N) {
    float *da = dA + (((int64_t)c * gridDim.x / nchunk + gh) * NP + q) * NP + 4 * g;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      *reinterpret_cast<float4 *>(da + 16 * t) = make_float4(dacc[t][0], dacc[t][1], dacc[t][2], dacc[t][3]);
  }
}

// Waves per workgroup. Measured on MI355X (dec shape B=64, 16 heads, N=196, d_h=32; tools/attn_bench.py, one box): the
// backward (two passes of dependent MFMA -> softmax algebra -> MFMA chains, latency-bound at two waves per SIMD) runs 62.9 us
// with 4 waves and 47.2 us with 8; the forward 25.5 us with 4 and 32.2 us with 8 (its 13 strips split worse over 8 waves and
// its staging phase is the larger share). Hence 4 forward / 8 backward; EVP_ATTN_FWD_WAVES / EVP_ATTN_BWD_WAVES override (4, 8;
// backward also 16).
static int g_attn_fwd_waves = 0, g_attn_bwd_waves = 0;
static inline int env_waves(const char *name, int dflt) {
  const char *e = getenv(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return (v == 4 || v == 8 || v == 16) ? v : dflt;
}
static inline int attn_fwd_waves() {
  if (g_attn_fwd_waves == 0) { g_attn_fwd_waves = env_waves("EVP_ATTN_FWD_WAVES", 4); if (g_attn_fwd_waves == 16) g_attn_fwd_waves = 8; }
  return g_attn_fwd_waves;
}
static inline int attn_bwd_waves() {
  if (g_attn_bwd_waves == 0) g_attn_bwd_waves = env_waves("EVP_ATTN_BWD_WAVES", 8);
  return g_attn_bwd_waves;
}

template <int DH, int NT>
int launch_fwd(const bf16_t *qkv, bf16_t *out, float *lse, bf16_t *probs, int B, int N, int heads, float scale, int64_t ldp, hipStream_t s) {
  constexpr int smem = 3 * 16 * NT * DH * 2;
  if (attn_fwd_waves() == 8) {
    auto k = attn_fwd_kernel<DH, NT, 8>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, dim3(B * heads), dim3(512), smem, s, qkv, out, lse, probs, N, heads, scale, ldp, g_attn_dbg, (const float *)nullptr, 1);
  } else {
    auto k = attn_fwd_kernel<DH, NT, 4>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, dim3(B * heads), dim3(256), smem, s, qkv, out, lse, probs, N, heads, scale, ldp, g_attn_dbg, (const float *)nullptr, 1);
  }
  EVP_CHECK_LAUNCH("evp_attention_fused_fwd");
  return EVP_OK;
}
template <int DH, int NT>
int launch_bwd(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse, bf16_t *dqkv, int B, int N, int heads, float scale,
               hipStream_t s) {
  constexpr int smem = 4 * 16 * NT * DH * 2 + 2 * 16 * NT * 4;
  auto go = [&](auto kfn, int nthr) {
    if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kfn, dim3(B * heads), dim3(nthr), smem, s, qkv, out, dout, lse, dqkv, N, heads, scale, g_attn_dbg);
  };
  const int nw = attn_bwd_waves();
  if (nw == 16) go(attn_bwd_kernel<DH, NT, 16>, 1024);
  else if (nw == 8) go(attn_bwd_kernel<DH, NT, 8>, 512);
  else go(attn_bwd_kernel<DH, NT, 4>, 256);
  EVP_CHECK_LAUNCH("evp_attention_fused_bwd");
  return EVP_OK;
}

// ---------------------------------------------------------------------------------------------------- windowed attention
// addm[g][h][i][j] = rel[g][i][j] >= 0 ? table[rel][h] : -100 (swin_block.py:135-158: gathered relative-position bias; the pairs the
// reference masks -- other window, padding slot -- carry index -1), and its transpose per (g, h); NP x NP planes, zero outside N x N.
__global__ __launch_bounds__(256) void win_bias_build_kernel(const float *__restrict__ table, const int32_t *__restrict__ rel, float *__restrict__ addm,
                                                             float *__restrict__ addmT, int nG, int N, int NP, int H, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % NP);
  const int i = (int)((e / NP) % NP);
  const int h = (int)((e / ((int64_t)NP * NP)) % H);
  const int g = (int)(e / ((int64_t)NP * NP * H));
  float v = 0.f;
  if (i < N && j < N) {
    const int r = rel[((int64_t)g * N + i) * N + j];
    v = r >= 0 ? table[(int64_t)r * H + h] : -100.0f;
  }
  addm[e] = v;
  addmT[(((int64_t)g * H + h) * NP + j) * NP + i] = v;
}

// dtable[r][h] += sum over (chunk, g, i, j) with rel[g][i][j] == r of dA[chunk][g][h][i][j]: workgroup = ((group, head), slice of the
// N x N pairs), the table privatised in LDS, at most R global atomics per workgroup (dtable zeroed by the launcher)
__global__ __launch_bounds__(256) void win_bias_reduce_kernel(const float *__restrict__ dA, const int32_t *__restrict__ rel, float *__restrict__ dtable,
                                                              int N, int NP, int H, int R, int nchunk, int nplanes) {
  // float64 cells: `ds_add_f32` serialises its lanes on gfx950 (3 cycles each), `ds_add_f64` does not (tools/native/lds_atomic_probe.hip),
  // and this table takes every add of the workgroup on R <= 169 addresses. (Measured in the Swin-B step: 14.0 us per launch either way --
  // the kernel is bound by reading the dA planes; the f64 table stays because its sum does not depend on the order of the adds.)
  extern __shared__ double tab[];
  const int g = blockIdx.x / H, h = blockIdx.x % H;
  for (int r = threadIdx.x; r < R; r += 256) tab[r] = 0.0;
  __syncthreads();
  // a thread takes 4 consecutive keys of one query row (16-byte loads of every chunk's plane); slices of the N x (NP / 4) pieces
  const int Q4 = NP / 4, total = N * Q4;
  const int per = (total + gridDim.y - 1) / gridDim.y, e0 = blockIdx.y * per, e1 = (e0 + per < total) ? e0 + per : total;
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    const int i = e / Q4, j0 = (e - i * Q4) * 4;
    if (j0 >= N) continue;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = dA + (int64_t)blockIdx.x * NP * NP + i * NP + j0;
    for (int c = 0; c < nchunk; ++c) {
      const float4 a = *reinterpret_cast<const float4 *>(src + (int64_t)c * nplanes * NP * NP);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
    const int32_t *rrow = rel + ((int64_t)g * N + i) * N + j0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + u < N) {
        const int r = rrow[u];
        if (r >= 0) atomicAdd(&tab[r], (double)vv[u]);
      }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    const float t = (float)tab[r];
    if (t != 0.f) unsafeAtomicAdd(dtable + (int64_t)r * H + h, t);
  }
}

// batch chunks of the windowed backward: at least ~512 workgroups in the launch (1024 / 2048 / 4096 measured the same or slower on
// Swin-T / Swin-B; EVP_WIN_BWD_WGS overrides), every chunk non-empty
static inline int win_target_wgs() {
  static const int v = [] {
    const char *e = getenv("EVP_WIN_BWD_WGS");
    const int n = e ? atoi(e) : 0;
    return n >= 64 && n <= 65536 ? n : 512;
  }();
  return v;
}
static inline void win_chunks(int Bg, int nG, int heads, int *nchunk, int *per) {
  const int B = Bg / nG;
  int nc = (win_target_wgs() + nG * heads - 1) / (nG * heads);
  if (nc > B) nc = B;
  if (nc < 1) nc = 1;
  *per = (B + nc - 1) / nc;
  *nchunk = (B + *per - 1) / *per;
}

template <int NT>
int launch_win_fwd(const bf16_t *qkv, const float *addm, bf16_t *out, float *lse, int Bg, int nG, int N, int heads, float scale, hipStream_t s) {
  constexpr int smem = 3 * 16 * NT * 32 * 2;
  auto k = attn_fwd_kernel<32, NT, 4, true>;
  hipLaunchKernelGGL(k, dim3(Bg * heads), dim3(256), smem, s, qkv, out, lse, (bf16_t *)nullptr, N, heads, scale, (int64_t)0,
                     (unsigned long long *)nullptr, addm, nG);
  EVP_CHECK_LAUNCH("evp_window_attention_fused_fwd");
  return EVP_OK;
}
template <int NT>
int launch_win_bwd(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse, const float *addm, const float *addmT, bf16_t *dqkv,
                   float *dA, int Bg, int nG, int N, int heads, float scale, hipStream_t s) {
  constexpr int smem = 4 * 16 * NT * 32 * 2 + 2 * 16 * NT * 4;
  constexpr int NW = NT <= 4 ? 4 : 8;          // one 16-query strip per wave: no more waves than strips
  const int B = Bg / nG;
  int nchunk, per;
  win_chunks(Bg, nG, heads, &nchunk, &per);
  auto k = win_attn_bwd_mfma_kernel<NT, NW>;
  hipLaunchKernelGGL(k, dim3(nG * heads * nchunk), dim3(64 * NW), smem, s, qkv, out, dout, lse, dqkv, N, heads, scale, addm, addmT, dA, nG, B, nchunk,
                     per);
  EVP_CHECK_LAUNCH("evp_window_attention_fused_bwd");
  return EVP_OK;
}
#define DISPATCH_WIN_NT(CALL)                         \
  if (N <= 32) { constexpr int NTV = 2; CALL; }       \
  else if (N <= 64) { constexpr int NTV = 4; CALL; }  \
  else if (N <= 96) { constexpr int NTV = 6; CALL; }  \
  else { constexpr int NTV = 8; CALL; }

#define DISPATCH_NT(DHV, CALL)                                                  \
  if (N <= 32) { constexpr int NTV = 2; constexpr int DHC = DHV; CALL; }        \
  else if (N <= 64) { constexpr int NTV = 4; constexpr int DHC = DHV; CALL; }   \
  else if (N <= 128) { constexpr int NTV = 8; constexpr int DHC = DHV; CALL; }  \
  else { constexpr int NTV = 14; constexpr int DHC = DHV; CALL; }

}  // namespace

extern "C" int evp_attention_set_debug_buffer(void *buf) {   // measurement aid: uint64 [B*heads*4*2] {staging, compute} cycles per wave
  g_attn_dbg = reinterpret_cast<unsigned long long *>(buf);
  return EVP_OK;
}

extern "C" int evp_attention_fused_supported(int dtype, int N, int dh) {
  return dtype == EVP_BF16 && N >= 1 && N <= 224 && (dh == 32 || dh == 64);
}

extern "C" int evp_attention_fused_fwd(const void *qkv, int B, int N, int heads, int dh, float scale, void *out, float *lse,
                                       void *probs, int64_t ldp, void *stream) {
  EVP_CHECK_ARG(qkv && out && lse, EVP_EINVAL, "evp_attention_fused_fwd: null pointer");
  EVP_CHECK_ARG(evp_attention_fused_supported(EVP_BF16, N, dh) && B > 0 && heads > 0, EVP_EUNSUPPORTED,
                "evp_attention_fused_fwd: needs bf16, N<=224, dh in {32,64} (N=%d dh=%d)", N, dh);
  EVP_CHECK_ARG(!probs || (ldp % 4 == 0 && ldp >= N), EVP_ESHAPE, "evp_attention_fused_fwd: bad ldp");
  hipStream_t s = (hipStream_t)stream;
  if (dh == 64) { DISPATCH_NT(64, return (launch_fwd<DHC, NTV>((const bf16_t *)qkv, (bf16_t *)out, lse, (bf16_t *)probs, B, N, heads, scale, ldp, s))) }
  DISPATCH_NT(32, return (launch_fwd<DHC, NTV>((const bf16_t *)qkv, (bf16_t *)out, lse, (bf16_t *)probs, B, N, heads, scale, ldp, s)))
}

extern "C" int evp_attention_fused_bwd(const void *qkv, const void *out, const void *dout, const float *lse, int B, int N, int heads,
                                       int dh, float scale, void *dqkv, void *stream) {
  EVP_CHECK_ARG(qkv && out && dout && lse && dqkv, EVP_EINVAL, "evp_attention_fused_bwd: null pointer");
  EVP_CHECK_ARG(evp_attention_fused_supported(EVP_BF16, N, dh) && B > 0 && heads > 0, EVP_EUNSUPPORTED,
                "evp_attention_fused_bwd: needs bf16, N<=224, dh in {32,64} (N=%d dh=%d)", N, dh);
  hipStream_t s = (hipStream_t)stream;
  if (dh == 64) { DISPATCH_NT(64, return (launch_bwd<DHC, NTV>((const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, B, N, heads, scale, s))) }
  DISPATCH_NT(32, return (launch_bwd<DHC, NTV>((const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, B, N, heads, scale, s)))
}

// ---------------------------------------------------------------------------------------------------- windowed attention (MFMA)
// The Swin blocks' attention on grouped tokens (model/sub_module/swin_block.py:113-162) on the fused kernels above: d_h = 32, a
// group of N <= 128 tokens per workgroup, logits = q.k*scale + relative-position bias, -100 on the pairs the reference masks.
// Replaces the f32 LDS kernels of window.hip in bf16 mode (those remain the f32 parity path and serve the blocks that return
// their probabilities).
extern "C" int evp_window_attention_fused_np(int N) { return N <= 32 ? 32 : N <= 64 ? 64 : N <= 96 ? 96 : 128; }

extern "C" int evp_window_bias_build(const float *table, const int32_t *rel, int nG, int N, int heads, int R, float *addm, float *addmT,
                                     void *stream) {
  EVP_CHECK_ARG(table && rel && addm && addmT, EVP_EINVAL, "evp_window_bias_build: null pointer");
  EVP_CHECK_ARG(nG > 0 && N > 0 && N <= 128 && heads > 0 && R > 0, EVP_ESHAPE, "evp_window_bias_build: bad shape (N=%d)", N);
  const int NP = evp_window_attention_fused_np(N);
  const int64_t total = (int64_t)nG * heads * NP * NP;
  hipLaunchKernelGGL(win_bias_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, rel, addm, addmT, nG, N,
                     NP, heads, total);
  EVP_CHECK_LAUNCH("evp_window_bias_build");
  return EVP_OK;
}

extern "C" int evp_window_attention_fused_fwd(const void *qkv, const float *addm, int Bg, int nG, int N, int heads, float scale, void *out,
                                              float *lse, void *stream) {
  EVP_CHECK_ARG(qkv && addm && out && lse, EVP_EINVAL, "evp_window_attention_fused_fwd: null pointer");
  EVP_CHECK_ARG(Bg > 0 && nG > 0 && Bg % nG == 0 && N > 0 && N <= 128 && heads > 0, EVP_ESHAPE,
                "evp_window_attention_fused_fwd: bad shape (Bg=%d nG=%d N=%d)", Bg, nG, N);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_WIN_NT(return (launch_win_fwd<NTV>((const bf16_t *)qkv, addm, (bf16_t *)out, lse, Bg, nG, N, heads, scale, s)))
}

extern "C" int evp_window_attention_fused_bwd(const void *qkv, const void *out, const void *dout, const float *lse, const float *addm,
                                              const float *addmT, int Bg, int nG, int N, int heads, float scale, void *dqkv, float *dA,
                                              void *stream) {
  EVP_CHECK_ARG(qkv && out && dout && lse && addm && addmT && dqkv && dA, EVP_EINVAL, "evp_window_attention_fused_bwd: null pointer");
  EVP_CHECK_ARG(Bg > 0 && nG > 0 && Bg % nG == 0 && N > 0 && N <= 128 && heads > 0, EVP_ESHAPE,
                "evp_window_attention_fused_bwd: bad shape (Bg=%d nG=%d N=%d)", Bg, nG, N);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_WIN_NT(return (launch_win_bwd<NTV>((const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, addm, addmT, (bf16_t *)dqkv, dA, Bg,
                                              nG, N, heads, scale, s)))
}

extern "C" int evp_window_attention_fused_nchunk(int Bg, int nG, int heads) {
  if (Bg <= 0 || nG <= 0 || heads <= 0 || Bg % nG != 0) return 0;
  int nchunk, per;
  win_chunks(Bg, nG, heads, &nchunk, &per);
  return nchunk;
}

extern "C" int evp_window_bias_reduce(const float *dA, const int32_t *rel, int Bg, int nG, int N, int heads, int R, float *dtable, void *stream) {
  EVP_CHECK_ARG(dA && rel && dtable, EVP_EINVAL, "evp_window_bias_reduce: null pointer");
  EVP_CHECK_ARG(nG > 0 && N > 0 && N <= 128 && heads > 0 && R > 0 && R <= 4096, EVP_ESHAPE, "evp_window_bias_reduce: bad shape");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = evp_zero_async(dtable, (size_t)R * heads * sizeof(float), s);
  EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_window_bias_reduce: clearing dtable failed: %s", hipGetErrorString(e));
  const int nchunk = evp_window_attention_fused_nchunk(Bg, nG, heads);
  EVP_CHECK_ARG(nchunk > 0, EVP_ESHAPE, "evp_window_bias_reduce: Bg must be a positive multiple of nG");
  int split = (256 + nG * heads - 1) / (nG * heads);          // ~256 workgroups
  if (split > 16) split = 16;
  hipLaunchKernelGGL(win_bias_reduce_kernel, dim3((unsigned)(nG * heads), (unsigned)split), dim3(256), (size_t)R * sizeof(double), s, dA, rel, dtable, N,
                     evp_window_attention_fused_np(N), heads, R, nchunk, nG * heads);
  EVP_CHECK_LAUNCH("evp_window_bias_reduce");
  return EVP_OK;
}
