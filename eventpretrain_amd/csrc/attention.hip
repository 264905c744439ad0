// K6: multi-head self-attention core on the packed qkv tensor, v1 "unfused" formulation for gfx950:
//   scores = scale * q k^T   (batched MFMA GEMM, f32 out)  ->  row softmax (wave per row)  ->  out = probs v.
// and its backward (dP, softmax', dV, dQ, dK as five batched GEMMs + one row kernel). The token counts on this path
// are tiny (N = 98 / 196, d_h = 64 / 32), so a whole head's K/V fit on chip; a fused per-head kernel is the planned
// replacement (DESIGN.md "next"). Replaces model/sub_module/vit_block.py:134-140.
#include "evp_common.h"

namespace {

// p[row, :] = softmax(s[row, :n_valid]); pad columns [n_valid, ld) are zeroed so that the GEMMs that consume p with
// K = n_valid rounded up to the 16-byte chunk see exact zeros.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *s, void *p, int dtype, int64_t rows, int n_valid, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const float *sr = s + r * ld;
    float mx = -INFINITY;
    for (int j = lane; j < n_valid; j += 64) mx = fmaxf(mx, sr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n_valid; j += 64) sum += expf(sr[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < ld; j += 64) st_any(p, dtype, r * ld + j, j < n_valid ? expf(sr[j] - mx) * inv : 0.f);
  }
}

// ds = p * (dp - sum_j p_j dp_j); pad columns zeroed
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const void *p, const float *dp, void *ds, int dtype, int64_t rows,
                                                               int n_valid, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float dot = 0.f;
    for (int j = lane; j < n_valid; j += 64) dot += ld_any(p, dtype, r * ld + j) * dp[r * ld + j];
    dot = wave_sum(dot);
    for (int j = lane; j < ld; j += 64) {
      const float v = j < n_valid ? ld_any(p, dtype, r * ld + j) * (dp[r * ld + j] - dot) : 0.f;
      st_any(ds, dtype, r * ld + j, v);
    }
  }
}

static inline int rows_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}
static inline size_t esize(int dtype) { return dtype == EVP_BF16 ? 2 : 4; }

}  // namespace

extern "C" int evp_softmax_rows(const float *s, void *p, int dtype, int64_t rows, int n_valid, int64_t ld, void *stream) {
  EVP_CHECK_ARG(s && p && rows > 0 && n_valid > 0 && ld >= n_valid, EVP_EINVAL, "evp_softmax_rows: bad argument");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows_grid(rows)), dim3(256), 0, (hipStream_t)stream, s, p, dtype, rows, n_valid, ld);
  EVP_CHECK_LAUNCH("evp_softmax_rows");
  return EVP_OK;
}

extern "C" int evp_softmax_rows_bwd(const void *p, const float *dp, void *ds, int dtype, int64_t rows, int n_valid, int64_t ld, void *stream) {
  EVP_CHECK_ARG(p && dp && ds && rows > 0 && n_valid > 0 && ld >= n_valid, EVP_EINVAL, "evp_softmax_rows_bwd: bad argument");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(rows_grid(rows)), dim3(256), 0, (hipStream_t)stream, p, dp, ds, dtype, rows, n_valid, ld);
  EVP_CHECK_LAUNCH("evp_softmax_rows_bwd");
  return EVP_OK;
}

// qkv: [B, N, 3, h, dh]; token stride 3*h*dh; q/k/v of head g start at g*dh (+0, +h*dh, +2*h*dh)
extern "C" int evp_attention_fwd(const void *qkv, int dtype, int B, int N, int heads, int dh, float scale, float *scores_ws,
                                 void *probs, int64_t ldp, void *out, void *stream) {
  EVP_CHECK_ARG(qkv && scores_ws && probs && out, EVP_EINVAL, "evp_attention_fwd: null pointer");
  EVP_CHECK_ARG(B > 0 && N > 0 && heads > 0 && dh > 0 && dh % 8 == 0 && ldp % 8 == 0 && ldp >= N, EVP_ESHAPE,
                "evp_attention_fwd: need dh%%8==0, ldp%%8==0, ldp>=N (N=%d dh=%d ldp=%lld)", N, dh, (long long)ldp);
  const int64_t C = (int64_t)heads * dh, tok = 3 * C;
  const size_t es = esize(dtype);
  const char *base = (const char *)qkv;
  evp_gemm_desc d = {};
  // scores[b,g] = scale * q k^T
  d.dtype = dtype; d.transA = 0; d.transB = 0; d.M = N; d.N = N; d.K = dh;
  d.A = base; d.lda = tok; d.strideA0 = (int64_t)N * tok; d.strideA1 = dh;
  d.B = base + C * es; d.ldb = tok; d.strideB0 = (int64_t)N * tok; d.strideB1 = dh;
  d.C = scores_ws; d.c_dtype = EVP_F32; d.ldc = ldp; d.strideC0 = (int64_t)heads * N * ldp; d.strideC1 = (int64_t)N * ldp;
  d.batch0 = B; d.batch1 = heads; d.alpha = scale;
  int rc = evp_gemm(&d, stream);
  if (rc) return rc;
  rc = evp_softmax_rows(scores_ws, probs, dtype, (int64_t)B * heads * N, N, ldp, stream);
  if (rc) return rc;
  // out[b, :, g] = probs v
  evp_gemm_desc o = {};
  o.dtype = dtype; o.transA = 0; o.transB = 1; o.M = N; o.N = dh; o.K = N;
  o.A = probs; o.lda = ldp; o.strideA0 = (int64_t)heads * N * ldp; o.strideA1 = (int64_t)N * ldp;
  o.B = base + 2 * C * es; o.ldb = tok; o.strideB0 = (int64_t)N * tok; o.strideB1 = dh;
  o.C = out; o.c_dtype = dtype; o.ldc = C; o.strideC0 = (int64_t)N * C; o.strideC1 = dh;
  o.batch0 = B; o.batch1 = heads; o.alpha = 1.0f;
  return evp_gemm(&o, stream);
}

extern "C" int evp_attention_bwd(const void *qkv, const void *probs, const void *dout, int dtype, int B, int N, int heads, int dh,
                                 float scale, int64_t ldp, float *dp_ws, void *ds_ws, void *dqkv, void *stream) {
  EVP_CHECK_ARG(qkv && probs && dout && dp_ws && ds_ws && dqkv, EVP_EINVAL, "evp_attention_bwd: null pointer");
  EVP_CHECK_ARG(B > 0 && N > 0 && heads > 0 && dh > 0 && dh % 8 == 0 && ldp % 8 == 0 && ldp >= N, EVP_ESHAPE, "evp_attention_bwd: bad shape");
  const int64_t C = (int64_t)heads * dh, tok = 3 * C;
  const size_t es = esize(dtype);
  const char *base = (const char *)qkv;
  char *dbase = (char *)dqkv;
  const int64_t sP0 = (int64_t)heads * N * ldp, sP1 = (int64_t)N * ldp;
  int rc;
  {  // dP = dO v^T  (f32)
    evp_gemm_desc d = {};
    d.dtype = dtype; d.transA = 0; d.transB = 0; d.M = N; d.N = N; d.K = dh;
    d.A = dout; d.lda = C; d.strideA0 = (int64_t)N * C; d.strideA1 = dh;
    d.B = base + 2 * C * es; d.ldb = tok; d.strideB0 = (int64_t)N * tok; d.strideB1 = dh;
    d.C = dp_ws; d.c_dtype = EVP_F32; d.ldc = ldp; d.strideC0 = sP0; d.strideC1 = sP1;
    d.batch0 = B; d.batch1 = heads; d.alpha = 1.0f;
    if ((rc = evp_gemm(&d, stream))) return rc;
  }
  if ((rc = evp_softmax_rows_bwd(probs, dp_ws, ds_ws, dtype, (int64_t)B * heads * N, N, ldp, stream))) return rc;
  {  // dV = P^T dO
    evp_gemm_desc d = {};
    d.dtype = dtype; d.transA = 1; d.transB = 1; d.M = N; d.N = dh; d.K = N;
    d.A = probs; d.lda = ldp; d.strideA0 = sP0; d.strideA1 = sP1;
    d.B = dout; d.ldb = C; d.strideB0 = (int64_t)N * C; d.strideB1 = dh;
    d.C = dbase + 2 * C * es; d.c_dtype = dtype; d.ldc = tok; d.strideC0 = (int64_t)N * tok; d.strideC1 = dh;
    d.batch0 = B; d.batch1 = heads; d.alpha = 1.0f;
    if ((rc = evp_gemm(&d, stream))) return rc;
  }
  {  // dQ = scale * dS k
    evp_gemm_desc d = {};
    d.dtype = dtype; d.transA = 0; d.transB = 1; d.M = N; d.N = dh; d.K = N;
    d.A = ds_ws; d.lda = ldp; d.strideA0 = sP0; d.strideA1 = sP1;
    d.B = base + C * es; d.ldb = tok; d.strideB0 = (int64_t)N * tok; d.strideB1 = dh;
    d.C = dbase; d.c_dtype = dtype; d.ldc = tok; d.strideC0 = (int64_t)N * tok; d.strideC1 = dh;
    d.batch0 = B; d.batch1 = heads; d.alpha = scale;
    if ((rc = evp_gemm(&d, stream))) return rc;
  }
  {  // dK = scale * dS^T q
    evp_gemm_desc d = {};
    d.dtype = dtype; d.transA = 1; d.transB = 1; d.M = N; d.N = dh; d.K = N;
    d.A = ds_ws; d.lda = ldp; d.strideA0 = sP0; d.strideA1 = sP1;
    d.B = base; d.ldb = tok; d.strideB0 = (int64_t)N * tok; d.strideB1 = dh;
    d.C = dbase + C * es; d.c_dtype = dtype; d.ldc = tok; d.strideC0 = (int64_t)N * tok; d.strideC1 = dh;
    d.batch0 = B; d.batch1 = heads; d.alpha = scale;
    if ((rc = evp_gemm(&d, stream))) return rc;
  }
  return EVP_OK;
}
