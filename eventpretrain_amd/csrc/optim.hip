// K21: fused multi-tensor AdamW (torch.optim.AdamW semantics, main_pretrain.py:341-343) and the global gradient
// norm (utils/misc.py:303-315). One launch covers every parameter: a host-built chunk table maps each workgroup
// to (tensor, offset); weight decay / lr scale are per tensor. The step also writes the bf16 shadow of each weight
// that the next forward's MFMA GEMMs read, so no separate cast pass touches the 112 M parameters.
#include "evp_common.h"

namespace {

struct AdamArgs {
  float *const *params; const float *const *grads; float *const *exp_avg; float *const *exp_avg_sq;
  uint16_t *const *lp; const int64_t *numel; const float *wd; const float *lr_scale;
  const int32_t *chunk_tensor; const int64_t *chunk_offset; int chunk_elems;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale;
  const float *dev_hyper;  // optional device-resident {bc1, bc2_sqrt, grad_scale, lr_mult}: graph-replay safe
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
  const int t = a.chunk_tensor[blockIdx.x];
  const int64_t off = a.chunk_offset[blockIdx.x];
  const int64_t n = a.numel[t];
  int64_t cnt = n - off; if (cnt > a.chunk_elems) cnt = a.chunk_elems;
  float *p = a.params[t] + off; const float *g = a.grads[t] + off;
  float *m = a.exp_avg[t] + off; float *v = a.exp_avg_sq[t] + off;
  uint16_t *lp = a.lp && a.lp[t] ? a.lp[t] + off : nullptr;
  float bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt, grad_scale = a.grad_scale, lr_mult = 1.0f;
  if (a.dev_hyper) { bc1 = a.dev_hyper[0]; bc2_sqrt = a.dev_hyper[1]; grad_scale = a.dev_hyper[2]; lr_mult = a.dev_hyper[3]; }
  const float lr = a.lr * lr_mult * a.lr_scale[t], wd = a.wd[t];
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  const bool vec = ((off & 3) == 0);
  const int64_t c4 = vec ? (cnt >> 2) : 0;
  for (int64_t i = threadIdx.x; i < c4; i += 256) {
    float4 pp = reinterpret_cast<float4 *>(p)[i];
    float4 gg = reinterpret_cast<const float4 *>(g)[i];
    float4 mm = reinterpret_cast<float4 *>(m)[i];
    float4 vv = reinterpret_cast<float4 *>(v)[i];
    float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = G[e] * grad_scale;
      P[e] *= decay;
      M[e] = a.beta1 * M[e] + (1.0f - a.beta1) * gr;
      V[e] = a.beta2 * V[e] + (1.0f - a.beta2) * gr * gr;
      const float denom = sqrtf(V[e]) / bc2_sqrt + a.eps;
      P[e] -= step_size * (M[e] / denom);
    }
    reinterpret_cast<float4 *>(p)[i] = pp;
    reinterpret_cast<float4 *>(m)[i] = mm;
    reinterpret_cast<float4 *>(v)[i] = vv;
    if (lp) {
      uint2 u;
      u.x = (uint32_t)f32_to_bf16(pp.x) | ((uint32_t)f32_to_bf16(pp.y) << 16);
      u.y = (uint32_t)f32_to_bf16(pp.z) | ((uint32_t)f32_to_bf16(pp.w) << 16);
      reinterpret_cast<uint2 *>(lp)[i] = u;
    }
  }
  for (int64_t i = c4 * 4 + threadIdx.x; i < cnt; i += 256) {
    const float gr = g[i] * grad_scale;
    float pe = p[i] * decay;
    const float me = a.beta1 * m[i] + (1.0f - a.beta1) * gr;
    const float ve = a.beta2 * v[i] + (1.0f - a.beta2) * gr * gr;
    pe -= step_size * (me / (sqrtf(ve) / bc2_sqrt + a.eps));
    p[i] = pe; m[i] = me; v[i] = ve;
    if (lp) lp[i] = f32_to_bf16(pe);
  }
}

__global__ __launch_bounds__(256) void sumsq_chunks(const float *const *grads, const int64_t *numel, const int32_t *chunk_tensor,
                                                    const int64_t *chunk_offset, int chunk_elems, float *part) {
  __shared__ float red[16];
  const int t = chunk_tensor[blockIdx.x];
  const int64_t off = chunk_offset[blockIdx.x];
  int64_t cnt = numel[t] - off; if (cnt > chunk_elems) cnt = chunk_elems;
  const float *g = grads[t] + off;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void sqrt_sum(const float *part, int n, float *out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = sqrtf(s);
}

}  // namespace

extern "C" int evp_adamw_multi(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                               uint16_t *const *lp_shadow, const int64_t *numel, const float *weight_decay, const float *lr_scale,
                               const int32_t *chunk_tensor, const int64_t *chunk_offset, int n_chunks, int chunk_elems, float lr,
                               float beta1, float beta2, float eps, int step, float grad_scale, const float *dev_hyper, void *stream) {
  EVP_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel && weight_decay && lr_scale && chunk_tensor && chunk_offset,
                EVP_EINVAL, "evp_adamw_multi: null table");
  EVP_CHECK_ARG(n_chunks > 0 && chunk_elems > 0 && chunk_elems % 4 == 0 && step >= 1, EVP_EINVAL, "evp_adamw_multi: bad chunking/step");
  AdamArgs a;
  a.params = params; a.grads = grads; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.lp = lp_shadow; a.numel = numel;
  a.wd = weight_decay; a.lr_scale = lr_scale; a.chunk_tensor = chunk_tensor; a.chunk_offset = chunk_offset; a.chunk_elems = chunk_elems;
  a.dev_hyper = dev_hyper;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, a);
  EVP_CHECK_LAUNCH("evp_adamw_multi");
  return EVP_OK;
}

extern "C" int evp_grad_norm_multi(const float *const *grads, const int64_t *numel, const int32_t *chunk_tensor,
                                   const int64_t *chunk_offset, int n_chunks, int chunk_elems, float *workspace, float *out,
                                   void *stream) {
  EVP_CHECK_ARG(grads && numel && chunk_tensor && chunk_offset && workspace && out && n_chunks > 0, EVP_EINVAL, "evp_grad_norm_multi: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_chunks, dim3(n_chunks), dim3(256), 0, s, grads, numel, chunk_tensor, chunk_offset, chunk_elems, workspace);
  EVP_CHECK_LAUNCH("evp_grad_norm_multi");
  hipLaunchKernelGGL(sqrt_sum, dim3(1), dim3(1024), 0, s, workspace, n_chunks, out);
  EVP_CHECK_LAUNCH("evp_grad_norm_multi(final)");
  return EVP_OK;
}
