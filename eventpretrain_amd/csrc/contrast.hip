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
// smoothing s > 0: timm's LabelSmoothingCrossEntropy (ft_cls_trainer.py:63-64), per row
//   (1 - s) * (-log p[label]) + s * (-mean_j log p[j]);   d/dlogit_j = p_j - ((1 - s) [j == label] + s / n_cls)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld,
                                                      float smoothing, float *row_loss, float *dlogits) {
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
  const float lse = logf(s) + mx;
  float row = lse - lr[lab];
  if (smoothing > 0.f) {  // block-uniform
    float sl = 0.f;
    for (int j = threadIdx.x; j < n_cls; j += 256) sl += lr[j];
    __syncthreads();
    sl = block_sum(sl, red);
    row = (1.0f - smoothing) * row + smoothing * (lse - sl / (float)n_cls);
  }
  if (threadIdx.x == 0) row_loss[r] = row;
  if (dlogits) {
    const float inv = 1.0f / s, invR = 1.0f / (float)R, off = smoothing / (float)n_cls, on = 1.0f - smoothing;
    float *dr = dlogits + r * ld;
    for (int j = threadIdx.x; j < ld; j += 256)
      dr[j] = j < n_cls ? (expf(lr[j] - mx) * inv - ((j == lab ? on : 0.f) + off)) * invR : 0.f;
  }
}
__global__ __launch_bounds__(1024) void mean_kernel(const float *v, int64_t n, float *out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// out[r] = sum_c a[r,c] * b[r,c]
__global__ __launch_bounds__(256) void rowdot_kernel(const float *a, const float *b, int64_t R, int C, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += a[r * C + j] * b[r * C + j];
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}
// out[r,c] = s[r] * x[r,c] (+ add[r,c])
__global__ __launch_bounds__(256) void scale_rows_kernel(const float *x, const float *s, const float *add, int64_t R, int C, float *out) {
  const int64_t total = R * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = s[i / C] * x[i];
    out[i] = add ? v + add[i] : v;
  }
}
// InfoNCE against a queue: logits_r = [pos_r, neg_r[0..K)] / T, label 0 (pr_hub_model.py:150-163)
__global__ __launch_bounds__(256) void infonce_queue_kernel(const float *pos, const float *neg, int64_t R, int K, int64_t ldn, float invT,
                                                            float *row_loss, float *dpos, float *dneg) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  const float *nr = neg + r * ldn;
  const float p0 = pos[r] * invT;
  float mx = p0;
  for (int j = threadIdx.x; j < K; j += 256) mx = fmaxf(mx, nr[j] * invT);
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int j = threadIdx.x; j < K; j += 256) s += expf(nr[j] * invT - mx);
  s = block_sum(s, red) + expf(p0 - mx);
  if (threadIdx.x == 0) row_loss[r] = (logf(s) + mx) - p0;
  if (dneg) {
    const float inv = 1.0f / s, sc = invT / (float)R;
    if (threadIdx.x == 0) dpos[r] = (expf(p0 - mx) * inv - 1.0f) * sc;
    float *dr = dneg + r * ldn;
    for (int j = threadIdx.x; j < ldn; j += 256) dr[j] = j < K ? expf(nr[j] * invT - mx) * inv * sc : 0.f;
  }
}
// queue[c, l, ptr + b] = keys[b, l, c]   (pr_hub_model.py:119: keys.T on a 3-D tensor reverses all dims)
__global__ __launch_bounds__(256) void enqueue_kernel(float *queue, const float *keys, int ptr, int B, int L, int C, int K) {
  const int64_t total = (int64_t)B * L * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i % B);
    const int64_t t = i / B;
    const int l = (int)(t % L), c = (int)(t / L);
    queue[((int64_t)c * L + l) * K + ptr + b] = keys[((int64_t)b * L + l) * C + c];
  }
}

// the same with the write position read from (and advanced in) device memory: no host round trip, HIP-graph capturable
__global__ __launch_bounds__(256) void enqueue_dev_kernel(float *queue, const float *keys, const int64_t *qptr, int B, int L, int C, int K) {
  const int ptr = (int)(*qptr);
  if (ptr < 0 || ptr + B > K) return;        // K % B == 0 is checked on the host; a corrupt pointer must not write out of range
  const int64_t total = (int64_t)B * L * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i % B);
    const int64_t t = i / B;
    const int l = (int)(t % L), c = (int)(t / L);
    queue[((int64_t)c * L + l) * K + ptr + b] = keys[((int64_t)b * L + l) * C + c];
  }
}
// Tiled form: queue is [C][L][K] (K contiguous), keys are [B][L][C] (C contiguous): a block moves a 64-batch x 64-channel tile of one
// token l through LDS, reading 256-byte runs along C and writing 256-byte runs along K (the element-wise form above reads every
// key with a stride of L*C floats: 153 us per step at B=64, L=197, C=256).
__global__ __launch_bounds__(256) void enqueue_dev_tiled_kernel(float *queue, const float *keys, const int64_t *qptr, int B, int L, int C, int K) {
  __shared__ float tile[64][65];
  const int ptr = (int)(*qptr);
  if (ptr < 0 || ptr + B > K) return;
  const int l = blockIdx.x, c0 = blockIdx.y * 64, b0 = blockIdx.z * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r, c = c0 + tx;
    tile[r][tx] = (b < B && c < C) ? keys[((int64_t)b * L + l) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, b = b0 + tx;
    if (c < C && b < B) queue[((int64_t)c * L + l) * K + ptr + b] = tile[tx][r];
  }
}
__global__ void advance_ptr_kernel(int64_t *qptr, int B, int K) { *qptr = (*qptr + B) % K; }

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
extern "C" int evp_cross_entropy_smooth(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld, float smoothing,
                                        float *loss, float *dlogits, float *workspace, void *stream) {
  EVP_CHECK_ARG(logits && labels && loss && workspace, EVP_EINVAL, "evp_cross_entropy: null pointer");
  EVP_CHECK_ARG(R > 0 && n_cls > 0 && ld >= n_cls && R < 2147483647LL, EVP_ESHAPE, "evp_cross_entropy: bad shape");
  EVP_CHECK_ARG(smoothing >= 0.f && smoothing < 1.f, EVP_EINVAL, "evp_cross_entropy: smoothing must be in [0, 1) (got %g)", (double)smoothing);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)R), dim3(256), 0, s, logits, labels, R, n_cls, ld, smoothing, workspace, dlogits);
  EVP_CHECK_LAUNCH("evp_cross_entropy");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, workspace, R, loss);
  EVP_CHECK_LAUNCH("evp_cross_entropy(mean)");
  return EVP_OK;
}

extern "C" int evp_cross_entropy(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld, float *loss,
                                 float *dlogits, float *workspace, void *stream) {
  return evp_cross_entropy_smooth(logits, labels, R, n_cls, ld, 0.f, loss, dlogits, workspace, stream);
}

extern "C" int evp_rowdot_f32(const float *a, const float *b, int64_t R, int C, float *out, void *stream) {
  EVP_CHECK_ARG(a && b && out && R > 0 && C > 0, EVP_EINVAL, "evp_rowdot_f32: bad argument");
  hipLaunchKernelGGL(rowdot_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, a, b, R, C, out);
  EVP_CHECK_LAUNCH("evp_rowdot_f32");
  return EVP_OK;
}
extern "C" int evp_scale_rows_f32(const float *x, const float *s, const float *add, int64_t R, int C, float *out, void *stream) {
  EVP_CHECK_ARG(x && s && out && R > 0 && C > 0, EVP_EINVAL, "evp_scale_rows_f32: bad argument");
  int64_t g = (R * C + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, s, add, R, C, out);
  EVP_CHECK_LAUNCH("evp_scale_rows_f32");
  return EVP_OK;
}
extern "C" int evp_infonce_queue(const float *pos, const float *neg, int64_t R, int K, int64_t ldn, float inv_T, float *loss,
                                 float *dpos, float *dneg, float *workspace, void *stream) {
  EVP_CHECK_ARG(pos && neg && loss && workspace, EVP_EINVAL, "evp_infonce_queue: null pointer");
  EVP_CHECK_ARG(R > 0 && K > 0 && ldn >= K && R < 2147483647LL && ((dpos == nullptr) == (dneg == nullptr)), EVP_ESHAPE, "evp_infonce_queue: bad shape");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(infonce_queue_kernel, dim3((unsigned)R), dim3(256), 0, s, pos, neg, R, K, ldn, inv_T, workspace, dpos, dneg);
  EVP_CHECK_LAUNCH("evp_infonce_queue");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, workspace, R, loss);
  EVP_CHECK_LAUNCH("evp_infonce_queue(mean)");
  return EVP_OK;
}
extern "C" int evp_enqueue_keys_dev(float *queue, const float *keys, int64_t *queue_ptr, int B, int L, int C, int K, void *stream) {
  EVP_CHECK_ARG(queue && keys && queue_ptr, EVP_EINVAL, "evp_enqueue_keys_dev: null pointer");
  EVP_CHECK_ARG(B > 0 && L > 0 && C > 0 && K > 0 && K % B == 0, EVP_ESHAPE, "evp_enqueue_keys_dev: queue length %d must be a multiple of the batch %d", K, B);
  if (L <= 65535 && (C + 63) / 64 <= 65535 && (B + 63) / 64 <= 65535) {
    hipLaunchKernelGGL(enqueue_dev_tiled_kernel, dim3((unsigned)L, (unsigned)((C + 63) / 64), (unsigned)((B + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       queue, keys, queue_ptr, B, L, C, K);
  } else {
    int64_t g = ((int64_t)B * L * C + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(enqueue_dev_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, queue, keys, queue_ptr, B, L, C, K);
  }
  EVP_CHECK_LAUNCH("evp_enqueue_keys_dev");
  hipLaunchKernelGGL(advance_ptr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, queue_ptr, B, K);
  EVP_CHECK_LAUNCH("evp_enqueue_keys_dev(advance)");
  return EVP_OK;
}

extern "C" int evp_enqueue_keys(float *queue, const float *keys, int ptr, int B, int L, int C, int K, void *stream) {
  EVP_CHECK_ARG(queue && keys, EVP_EINVAL, "evp_enqueue_keys: null pointer");
  EVP_CHECK_ARG(B > 0 && L > 0 && C > 0 && ptr >= 0 && ptr + B <= K, EVP_ESHAPE, "evp_enqueue_keys: ptr+B exceeds the queue length");
  int64_t g = ((int64_t)B * L * C + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(enqueue_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, queue, keys, ptr, B, L, C, K);
  EVP_CHECK_LAUNCH("evp_enqueue_keys");
  return EVP_OK;
}
