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

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;

namespace {

// ---- LDS image of a [rows][DH] bf16 array: 16-byte chunks, XOR-swizzled for 128-byte rows ---------------------------
template <int DH> __device__ __forceinline__ int img_off(int row, int chunk) {
  if (DH == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  return row * 64 + (chunk << 4);
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

// ---------------------------------------------------------------------------------------------------- forward
// NT = number of 16-wide score tiles (NP = 16*NT rows in LDS, NT even)
template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t *qkv, bf16_t *out, float *lse, bf16_t *probs, int N, int heads,
                                                       float scale, int64_t ldp) {
  constexpr int NP = 16 * NT, IMG = NP * DH * 2, KS = DH / 32, DT = DH / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int64_t C = (int64_t)heads * DH, tok = 3 * C;
  const bf16_t *base = qkv + (int64_t)b * N * tok + (int64_t)h * DH;
  stage_head<DH>(base, tok, N, NP, Qs, tid, 256);
  stage_head<DH>(base + C, tok, N, NP, Ks, tid, 256);
  stage_head<DH>(base + 2 * C, tok, N, NP, Vs, tid, 256);
  __syncthreads();
  const float c2 = scale * 1.44269504088896340736f;   // exp(x*scale) = exp2(x*c2)

  for (int strip = wave; strip * 16 < N; strip += 4) {
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
    // lane: query strip*16 + li; registers: keys 16t + 4g + r
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * t + 4 * g + r >= N) S[t][r] = -INFINITY;
        mx = fmaxf(mx, S[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[t][r] = exp2f((S[t][r] - mx) * c2);
        sum += S[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const int q = strip * 16 + li;
    if (g == 0 && q < N && lse) lse[(int64_t)blockIdx.x * N + q] = mx * scale + logf(sum);
#pragma unroll
    for (int t = 0; t < NT; ++t) S[t] = S[t] * inv;
    if (probs && q < N) {
      bf16_t *pr = probs + ((int64_t)blockIdx.x * N + q) * ldp;
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

// ---------------------------------------------------------------------------------------------------- backward
template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse,
                                                       bf16_t *dqkv, int N, int heads, float scale) {
  constexpr int NP = 16 * NT, IMG = NP * DH * 2, KS = DH / 32, DT = DH / 16, CH = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG, *Gs = smem + 3 * IMG;   // Gs = dO
  float *Ls = reinterpret_cast<float *>(smem + 4 * IMG), *Ds = Ls + NP;          // log-sum-exp, delta = rowsum(dO * O)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int64_t C = (int64_t)heads * DH, tok = 3 * C;
  const bf16_t *base = qkv + (int64_t)b * N * tok + (int64_t)h * DH;
  stage_head<DH>(base, tok, N, NP, Qs, tid, 256);
  stage_head<DH>(base + C, tok, N, NP, Ks, tid, 256);
  stage_head<DH>(base + 2 * C, tok, N, NP, Vs, tid, 256);
  const bf16_t *go = dout + (int64_t)b * N * C + (int64_t)h * DH, *oo = out + (int64_t)b * N * C + (int64_t)h * DH;
  stage_head<DH>(go, C, N, NP, Gs, tid, 256);
  // delta[q] = sum_d dO[q,d] * O[q,d]: DH/8 consecutive threads share a row
  {
    constexpr int CPR = DH / 8;
    for (int c = tid; c < NP * CPR; c += 256) {   // NP*CPR is a multiple of CPR, rows never straddle the loop bound
      const int row = c / CPR, ch = c % CPR;
      float d = 0.f;
      if (row < N) {
        const uint4 a = *reinterpret_cast<const uint4 *>(go + (int64_t)row * C + ch * 8);
        const uint4 o = *reinterpret_cast<const uint4 *>(oo + (int64_t)row * C + ch * 8);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          d += __uint_as_float(aw[e] << 16) * __uint_as_float(ow[e] << 16) +
               __uint_as_float(aw[e] & 0xFFFF0000u) * __uint_as_float(ow[e] & 0xFFFF0000u);
      }
#pragma unroll
      for (int o = 1; o < CPR; o <<= 1) d += __shfl_xor(d, o, 64);
      if (ch == 0) Ds[row] = d;
    }
    for (int r = tid; r < NP; r += 256) Ls[r] = r < N ? lse[(int64_t)blockIdx.x * N + r] : INFINITY;   // exp(-inf) = 0 pads
  }
  __syncthreads();
  const float c2 = scale * 1.44269504088896340736f, l2e = 1.44269504088896340736f;

  // ---- pass 1, query on the lane: dQ ----
  for (int strip = wave; strip * 16 < N; strip += 4) {
    bf16x8 qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = frag_rows<DH>(Qs, strip * 16, ks, lane);
      gf[ks] = frag_rows<DH>(Gs, strip * 16, ks, lane);
    }
    const float lq = Ls[strip * 16 + li] * l2e, dq_ = Ds[strip * 16 + li];
    const int q = strip * 16 + li;
    f32x4 accq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // key tiles in chunks of CH (even): only CH score tiles are live at a time, dQ accumulates across chunks
#pragma unroll 1
    for (int c0 = 0; c0 < NT; c0 += CH) {
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
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = (16 * t + 4 * g + r < N) ? exp2f(s[r] * c2 - lq) : 0.f;
            P[tt][r] = p * (dp[r] - dq_) * scale;   // dS
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ss = 0; ss < CH / 2; ++ss)
          if (c0 + 2 * ss < NT)
            accq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Ks, dt * 16, (c0 >> 1) + ss, lane),
                                                               pack_tiles(P[2 * ss], P[2 * ss + 1]), accq[dt], 0, 0, 0);
    }
    if (q < N) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<uint2 *>(dqkv + ((int64_t)b * N + q) * tok + h * DH + dt * 16 + 4 * g) = pack4(accq[dt]);
    }
  }

  // ---- pass 2, key on the lane: dK, dV ----
  for (int strip = wave; strip * 16 < N; strip += 4) {
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
          const float lqa[4] = {lq.x, lq.y, lq.z, lq.w}, dla[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = exp2f(s[r] * c2 - lqa[r] * l2e);   // padded queries carry lse = +inf -> p = 0
            P[tt][r] = p;
            dS[tt][r] = p * (dp[r] - dla[r]) * scale;
          }
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

template <int DH, int NT>
int launch_fwd(const bf16_t *qkv, bf16_t *out, float *lse, bf16_t *probs, int B, int N, int heads, float scale, int64_t ldp, hipStream_t s) {
  constexpr int smem = 3 * 16 * NT * DH * 2;
  auto k = attn_fwd_kernel<DH, NT>;
  if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(k, dim3(B * heads), dim3(256), smem, s, qkv, out, lse, probs, N, heads, scale, ldp);
  EVP_CHECK_LAUNCH("evp_attention_fused_fwd");
  return EVP_OK;
}
template <int DH, int NT>
int launch_bwd(const bf16_t *qkv, const bf16_t *out, const bf16_t *dout, const float *lse, bf16_t *dqkv, int B, int N, int heads, float scale,
               hipStream_t s) {
  constexpr int smem = 4 * 16 * NT * DH * 2 + 2 * 16 * NT * 4;
  auto k = attn_bwd_kernel<DH, NT>;
  if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(k, dim3(B * heads), dim3(256), smem, s, qkv, out, dout, lse, dqkv, N, heads, scale);
  EVP_CHECK_LAUNCH("evp_attention_fused_bwd");
  return EVP_OK;
}

#define DISPATCH_NT(DHV, CALL)                                                  \
  if (N <= 32) { constexpr int NTV = 2; constexpr int DHC = DHV; CALL; }        \
  else if (N <= 64) { constexpr int NTV = 4; constexpr int DHC = DHV; CALL; }   \
  else if (N <= 128) { constexpr int NTV = 8; constexpr int DHC = DHV; CALL; }  \
  else { constexpr int NTV = 14; constexpr int DHC = DHV; CALL; }

}  // namespace

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
