// Stochastic regularisers of the fine-tuning recipe (reference main_finetune_cls.py:151-153: drop_path_rate 0.1 by default):
//   DropPath (timm 0.3.2 `drop_path`, used at model/sub_module/vit_block.py:252-253, conv_block.py:43-49, swin_block.py:270-271):
//       out = x / keep_prob * floor(keep_prob + u_b),  u_b ~ U[0,1) drawn once per SAMPLE b
//   Dropout (nn.Dropout at vit_block.py:137-141,226-231, vit.py:114): out = x * keep / (1 - p), keep ~ Bernoulli(1 - p) per element
// timm is a pip dependency that is absent from /root/reference and from this image; its published forward is restated. The
// arithmetic is pinned against the CPU oracle for GIVEN u / masks; the random STREAM (who draws which number) is this build's own
// (parity unpinned): u comes from the caller (torch.rand on the device, like the mask noise), the element masks from Philox4x32-10
// keyed by (seed, element index / 4).
#include "evp_common.h"

namespace {

// out[m,:] = (res ? res[m,:] : 0) + s_b * x[m,:], b = m / rows_per_sample, s_b = floor(kp + u[b]) / kp (kp >= 1 or u == NULL: 1);
// lp (optional): bf16 copy of s_b * x[m,:]. D % 4 == 0.
__global__ __launch_bounds__(256) void rows_scale_kernel(const float *__restrict__ x, const float *__restrict__ u, float kp, const float *__restrict__ res,
                                                         int64_t M, int D, int rps, float *__restrict__ out, bf16_t *__restrict__ lp) {
  const int64_t n4 = M * (int64_t)(D / 4);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / (D / 4);
    const float s = (u && kp < 1.f) ? floorf(kp + u[m / rps]) / kp : 1.f;
    const float4 v = reinterpret_cast<const float4 *>(x)[e];
    float4 o = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    if (lp) {
      uint2 w;
      w.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
      w.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
      reinterpret_cast<uint2 *>(lp)[e] = w;
    }
    if (out) {
      if (res) {
        const float4 r = reinterpret_cast<const float4 *>(res)[e];
        o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
      }
      reinterpret_cast<float4 *>(out)[e] = o;
    }
  }
}

// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// 4 elements per thread: keep_i = (bits_i * 2^-32 >= p); out = x * keep / (1 - p); mask byte = keep
template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T *__restrict__ x, T *__restrict__ out, uint8_t *__restrict__ mask, int64_t n, float p,
                                                          uint64_t seed, const uint64_t *__restrict__ seed_dev, uint64_t offset) {
  // seed_dev: a device scalar added to the key -- drawn per step ON the device, so that a replayed HIP graph (whose kernel
  // arguments are frozen at capture) still draws fresh masks every replay
  if (seed_dev) seed += *seed_dev;
  const float inv = 1.f / (1.f - p);
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const uint64_t ctr = offset + (uint64_t)q;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = q * 4 + i;
      if (e < n) {
        const bool keep = (float)c[i] * 2.3283064365386963e-10f >= p;
        mask[e] = keep ? 1 : 0;
        ElemIO<T>::st(out + e, keep ? ElemIO<T>::ld(x + e) * inv : 0.f);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask, T *__restrict__ out, int64_t n, float scale) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    ElemIO<T>::st(out + e, mask[e] ? ElemIO<T>::ld(x + e) * scale : 0.f);
}

inline unsigned grid_for(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (unsigned)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int evp_rows_scale_f32(const float *x, const float *u, float keep_prob, const float *res, int64_t M, int D, int rows_per_sample, float *out,
                                  void *out_lp, void *stream) {
  EVP_CHECK_ARG(x && (out || out_lp) && M > 0 && D > 0 && D % 4 == 0 && rows_per_sample > 0, EVP_EINVAL, "evp_rows_scale_f32: bad argument (D %% 4 == 0)");
  EVP_CHECK_ARG(keep_prob > 0.f, EVP_EINVAL, "evp_rows_scale_f32: keep_prob must be > 0");
  hipLaunchKernelGGL(rows_scale_kernel, dim3(grid_for(M * (int64_t)(D / 4))), dim3(256), 0, (hipStream_t)stream, x, u, keep_prob, res, M, D, rows_per_sample, out,
                     reinterpret_cast<bf16_t *>(out_lp));
  EVP_CHECK_LAUNCH("evp_rows_scale_f32");
  return EVP_OK;
}

extern "C" int evp_dropout_fwd(const void *x, int dtype, void *out, void *mask, int64_t n, float p, uint64_t seed, const void *seed_dev, uint64_t offset,
                               void *stream) {
  EVP_CHECK_ARG(x && out && mask && n > 0 && p >= 0.f && p < 1.f, EVP_EINVAL, "evp_dropout_fwd: bad argument (0 <= p < 1)");
  EVP_CHECK_ARG(dtype == EVP_F32 || dtype == EVP_BF16, EVP_EINVAL, "evp_dropout_fwd: bad dtype %d", dtype);
  const unsigned g = grid_for((n + 3) / 4);
  if (dtype == EVP_F32)
    hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float *)x, (float *)out, (uint8_t *)mask, n, p, seed, (const uint64_t *)seed_dev, offset);
  else
    hipLaunchKernelGGL(dropout_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, (bf16_t *)out, (uint8_t *)mask, n, p, seed, (const uint64_t *)seed_dev, offset);
  EVP_CHECK_LAUNCH("evp_dropout_fwd");
  return EVP_OK;
}

extern "C" int evp_dropout_apply(const void *x, int dtype, const void *mask, void *out, int64_t n, float scale, void *stream) {
  EVP_CHECK_ARG(x && out && mask && n > 0, EVP_EINVAL, "evp_dropout_apply: bad argument");
  EVP_CHECK_ARG(dtype == EVP_F32 || dtype == EVP_BF16, EVP_EINVAL, "evp_dropout_apply: bad dtype %d", dtype);
  const unsigned g = grid_for(n);
  if (dtype == EVP_F32)
    hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float *)x, (const uint8_t *)mask, (float *)out, n, scale);
  else
    hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, (const uint8_t *)mask, (bf16_t *)out, n, scale);
  EVP_CHECK_LAUNCH("evp_dropout_apply");
  return EVP_OK;
}
