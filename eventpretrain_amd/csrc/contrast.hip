// K15/K16 pieces: row L2 normalisation (F.normalize) and cross entropy over logit rows, forward + backward.
// The logits themselves (einsum 'blc,clk->blk' against the queue, 'nlc,mlc->nlm' against gathered keys) are batched
// GEMMs issued from the Python layer through evp_gemm. Replaces model/pretrain/pr_hub_model.py:144-188.
#include "evp_common.h"

namespace {

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float *x, int64_t R, int C, float *y, float *norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    const float *xr = x + r * C;
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += xr[j] * xr[j];
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);  // F.normalize eps
    if (lane == 0) norm[r] = nrm;
    const float inv = 1.0f / nrm;
    for (int j = lane; j < C; j += 64) y[r * C + j] = xr[j] * inv;
  }
}
// dx = (dy - y * <dy, y>) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *dy, const float *y, const float *norm, int64_t R, int C, float *dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += dy[r * C + j] * y[r * C + j];
    s = wave_sum(s);
    const float inv = 1.0f / norm[r];
    for (int j = lane; j < C; j += 64) dx[r * C + j] = (dy[r * C + j] - y[r * C + j] * s) * inv;
  }
}

// one 256-thread block per row (n_cls may be 1+K = 65537)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld,
                                                      float *row_loss, float *dlogits) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  const float *lr = logits + r * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < n_cls; j += 256) mx = fmaxf(mx, lr[j]);
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int j = threadIdx.x; j < n_cls; j += 256) s += expf(lr[j] - mx);
  s = block_sum(s, red);
  const int64_t lab = labels[r];
  if (threadIdx.x == 0) row_loss[r] = (logf(s) + mx) - lr[lab];
  if (dlogits) {
    const float inv = 1.0f / s, invR = 1.0f / (float)R;
    float *dr = dlogits + r * ld;
    for (int j = threadIdx.x; j < ld; j += 256)
      dr[j] = j < n_cls ? (expf(lr[j] - mx) * inv - (j == lab ? 1.f : 0.f)) * invR : 0.f;
  }
}
__global__ __launch_bounds__(1024) void mean_kernel(const float *v, int64_t n, float *out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

static inline int rows_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int evp_l2norm_rows_fwd(const float *x, int64_t R, int C, float *y, float *norm, void *stream) {
  EVP_CHECK_ARG(x && y && norm && R > 0 && C > 0, EVP_EINVAL, "evp_l2norm_rows_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, x, R, C, y, norm);
  EVP_CHECK_LAUNCH("evp_l2norm_rows_fwd");
  return EVP_OK;
}
extern "C" int evp_l2norm_rows_bwd(const float *dy, const float *y, const float *norm, int64_t R, int C, float *dx, void *stream) {
  EVP_CHECK_ARG(dy && y && norm && dx && R > 0 && C > 0, EVP_EINVAL, "evp_l2norm_rows_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, dy, y, norm, R, C, dx);
  EVP_CHECK_LAUNCH("evp_l2norm_rows_bwd");
  return EVP_OK;
}
extern "C" int evp_cross_entropy(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld, float *loss,
                                 float *dlogits, float *workspace, void *stream) {
  EVP_CHECK_ARG(logits && labels && loss && workspace, EVP_EINVAL, "evp_cross_entropy: null pointer");
  EVP_CHECK_ARG(R > 0 && n_cls > 0 && ld >= n_cls && R < 2147483647LL, EVP_ESHAPE, "evp_cross_entropy: bad shape");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)R), dim3(256), 0, s, logits, labels, R, n_cls, ld, workspace, dlogits);
  EVP_CHECK_LAUNCH("evp_cross_entropy");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, workspace, R, loss);
  EVP_CHECK_LAUNCH("evp_cross_entropy(mean)");
  return EVP_OK;
}
