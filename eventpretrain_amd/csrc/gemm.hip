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

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;

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
template <typename T, bool TR, int ROWS> struct Img;

template <int ROWS> struct Img<bf16_t, false, ROWS> {
  static constexpr int BYTES = ROWS * 64 * 2;
  static __device__ __forceinline__ int chunk_off(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }
};
template <int ROWS> struct Img<bf16_t, true, ROWS> {
  static constexpr int BYTES = 64 * ROWS * 2;
  // k = LDS row, ch = 16-byte chunk along the ROWS (m or n) direction
  static __device__ __forceinline__ int chunk_off(int k, int ch) {
    if (ROWS == 128) return k * 256 + ((ch ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 4);
    else return k * 128 + ((ch ^ ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1)) << 4);
  }
};
template <int ROWS> struct Img<float, false, ROWS> {
  static constexpr int LD = 16 + 1;
  static constexpr int BYTES = ((ROWS * LD * 4 + 15) / 16) * 16;
};
template <int ROWS> struct Img<float, true, ROWS> {
  static constexpr int LD = ROWS + 16;
  static constexpr int BYTES = 16 * LD * 4;
};

// ---- global -> register staging of one operand tile --------------------------------------------------------
// Non-transposed: tile = ROWS rows x BK elements, chunks along k. Transposed: BK rows (k) x ROWS elements.
template <typename T, bool TR, int ROWS, int NT> struct Stage {
  static constexpr int BK = Cfg<T>::BK, EPC = Cfg<T>::EPC;
  static constexpr int CPR = TR ? ROWS / EPC : BK / EPC;           // chunks per LDS row
  static constexpr int NROW = TR ? BK : ROWS;
  static constexpr int NCHUNK = (NROW * CPR + NT - 1) / NT;         // per thread
  uint4 r[NCHUNK];

  // base: operand pointer for this batch; row0: first m/n of the tile; k0: first k; nrows: M or N; K: depth
  __device__ __forceinline__ void load(const T *base, int64_t ld, int row0, int k0, int nrows, int K, int tid) {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
      const int cid = tid + i * NT;
      const int lr = cid / CPR, c = cid % CPR;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (NROW * CPR % NT == 0 || cid < NROW * CPR) {
        if (!TR) {
          const int gr = row0 + lr, gk = k0 + c * EPC;
          if (gr < nrows && gk < K) v = *reinterpret_cast<const uint4 *>(base + (int64_t)gr * ld + gk);
        } else {
          const int gk = k0 + lr, gr = row0 + c * EPC;
          if (gk < K && gr < nrows) v = *reinterpret_cast<const uint4 *>(base + (int64_t)gk * ld + gr);
        }
      }
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(char *img, int tid) const {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
      const int cid = tid + i * NT;
      const int lr = cid / CPR, c = cid % CPR;
      if (NROW * CPR % NT == 0 || cid < NROW * CPR) {
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<uint4 *>(img + Img<T, TR, ROWS>::chunk_off(lr, c)) = r[i];
        } else {
          float *p = reinterpret_cast<float *>(img) + lr * Img<T, TR, ROWS>::LD + c * 4;
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

// ---- fragment loads ----------------------------------------------------------------------------------------
// bf16 fragment of the 16 rows starting at `rb` for k-step `ks` (32 deep): lane holds row rb+(l&15), k 8*(l>>4)..+7
template <bool TR, int ROWS>
__device__ __forceinline__ bf16x8 frag_bf16(const char *img, int rb, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if constexpr (!TR) {
    const int row = rb + i;
    const uint4 v = *reinterpret_cast<const uint4 *>(img + Img<bf16_t, false, ROWS>::chunk_off(row, ks * 4 + g));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const int q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;
    const int ch = (rb >> 3) + (p >> 1);
    const char *a0 = img + Img<bf16_t, true, ROWS>::chunk_off(k, ch) + 8 * (p & 1);
    const char *a1 = img + Img<bf16_t, true, ROWS>::chunk_off(k + 4, ch) + 8 * (p & 1);
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(a0));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(a1));
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    i16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  }
}
template <bool TR, int ROWS>
__device__ __forceinline__ float frag_f32(const char *img, int rb, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const float *f = reinterpret_cast<const float *>(img);
  if constexpr (!TR) return f[(rb + i) * Img<float, false, ROWS>::LD + ks * 4 + g];
  else return f[(ks * 4 + g) * Img<float, true, ROWS>::LD + rb + i];
}

// ---- the kernel --------------------------------------------------------------------------------------------
template <typename T, bool TA, bool TB, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM *WN * 64) void gemm_kernel(const GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BK = Cfg<T>::BK, KSTEP = Cfg<T>::KSTEP;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int A_BYTES = Img<T, TA, BM>::BYTES, B_BYTES = Img<T, TB, BN>::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b0 = blockIdx.z / p.batch1, b1 = blockIdx.z % p.batch1;
  const T *A = reinterpret_cast<const T *>(p.A) + b0 * p.sA0 + b1 * p.sA1;
  const T *B = reinterpret_cast<const T *>(p.B) + b0 * p.sB0 + b1 * p.sB1;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<T, TA, BM, NT> sa;
  Stage<T, TB, BN, NT> sb;
  const int ntiles = (p.K + BK - 1) / BK;

  sa.load(A, p.lda, m0, 0, p.M, p.K, tid);
  sb.load(B, p.ldb, n0, 0, p.N, p.K, tid);
  sa.store(smem, tid);
  sb.store(smem + A_BYTES, tid);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    char *cur = smem + (t & 1) * (A_BYTES + B_BYTES);
    char *nxt = smem + ((t + 1) & 1) * (A_BYTES + B_BYTES);
    const bool more = (t + 1) < ntiles;
    if (more) {
      sa.load(A, p.lda, m0, (t + 1) * BK, p.M, p.K, tid);
      sb.load(B, p.ldb, n0, (t + 1) * BK, p.N, p.K, tid);
    }
    const char *ia = cur, *ib = cur + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / KSTEP; ++ks) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_bf16<TA, BM>(ia, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = frag_bf16<TB, BN>(ib, wn * WTN + j * 16, ks, lane);
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
    if (more) {
      sa.store(nxt, tid);
      sb.store(nxt + A_BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
  const int64_t coff = b0 * p.sC0 + b1 * p.sC1;
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WTN + j * 16 + lg * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      const int nv = (p.N - n) < 4 ? (p.N - n) : 4;
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nv) v[e] += p.bias[n + e];
      }
      if (p.act == EVP_ACT_GELU || p.act == EVP_ACT_RELU) {
        if (p.aux) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) st_any(p.aux, p.c_dtype, coff + (int64_t)m * p.ldaux + n + e, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p.act == EVP_ACT_GELU ? gelu_f(v[e]) : fmaxf(v[e], 0.f);
      } else if (p.act == EVP_ACT_DGELU || p.act == EVP_ACT_DRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nv) {
            const float h = ld_any(p.aux, p.c_dtype, coff + (int64_t)m * p.ldaux + n + e);
            v[e] *= p.act == EVP_ACT_DGELU ? dgelu_f(h) : (h > 0.f ? 1.f : 0.f);
          }
      }
      if (p.residual) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nv) v[e] += p.residual[coff + (int64_t)m * p.ldres + n + e];
      }
      const int64_t o = coff + (int64_t)m * p.ldc + n;
      if (p.c_dtype == EVP_F32) {
        float *c = reinterpret_cast<float *>(p.C) + o;
        if (p.accumulate) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) v[e] += c[e];
        }
        if (nv == 4 && (o & 3) == 0) *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
        else
          for (int e = 0; e < nv; ++e) c[e] = v[e];
      } else {
        bf16_t *c = reinterpret_cast<bf16_t *>(p.C) + o;
        if (nv == 4 && (o & 3) == 0) {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
          pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
          *reinterpret_cast<uint2 *>(c) = pk;
        } else
          for (int e = 0; e < nv; ++e) c[e] = f32_to_bf16(v[e]);
      }
    }
  }
}

template <typename T, bool TA, bool TB, int BM, int BN, int WM, int WN>
int launch(const evp_gemm_desc *d, hipStream_t s) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = d->strideA0; p.sA1 = d->strideA1;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = d->strideB0; p.sB1 = d->strideB1;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = d->strideC0; p.sC1 = d->strideC1;
  p.batch1 = d->batch1 > 0 ? d->batch1 : 1;
  p.alpha = d->alpha; p.bias = d->bias; p.act = d->act; p.aux = d->aux; p.ldaux = d->ldaux;
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate;
  p.tiles_m = (d->M + BM - 1) / BM;
  const int tiles_n = (d->N + BN - 1) / BN;
  const int nb = (d->batch0 > 0 ? d->batch0 : 1) * p.batch1;
  constexpr int smem = 2 * (Img<T, TA, BM>::BYTES + Img<T, TB, BN>::BYTES);
  auto k = gemm_kernel<T, TA, TB, BM, BN, WM, WN>;
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
  dim3 grid((unsigned)(p.tiles_m * tiles_n), 1, (unsigned)nb);
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm");
  return EVP_OK;
}

template <typename T, bool TA, bool TB> int pick_tile(const evp_gemm_desc *d, hipStream_t s) {
  int tile = d->tile;
  if (tile == 0) {
    const int64_t big = (int64_t)((d->M + 127) / 128) * ((d->N + 127) / 128) * (d->batch0 > 0 ? d->batch0 : 1) *
                        (d->batch1 > 0 ? d->batch1 : 1);
    tile = (d->M >= 128 && d->N >= 128 && big >= 192) ? 1 : 2;
  }
  if (tile == 1) return launch<T, TA, TB, 128, 128, 2, 2>(d, s);
  return launch<T, TA, TB, 64, 64, 2, 2>(d, s);
}

template <typename T> int pick_layout(const evp_gemm_desc *d, hipStream_t s) {
  if (!d->transA && !d->transB) return pick_tile<T, false, false>(d, s);
  if (!d->transA && d->transB) return pick_tile<T, false, true>(d, s);
  if (d->transA && d->transB) return pick_tile<T, true, true>(d, s);
  evp_set_error("evp_gemm: layout transA=1,transB=0 is not used on this path and not built");
  return EVP_EUNSUPPORTED;
}

}  // namespace

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
  EVP_CHECK_ARG((d->act != EVP_ACT_DGELU && d->act != EVP_ACT_DRELU) || d->aux, EVP_EINVAL, "evp_gemm: dgelu/drelu need aux");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == EVP_BF16 ? pick_layout<bf16_t>(d, s) : pick_layout<float>(d, s);
}
