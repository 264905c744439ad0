// K12: reconstruction loss of the temporal-difference-map decoder (pr_hub_model.py:125-141 with
// utils/reshape.py:15-22 frame2emb folded into the addressing). One wave per patch row; HBM-bound and tiny.
#include "evp_common.h"

namespace {

struct PatchGeom { int C, H, W, p, gw, L, P; };

// element j of patch row (b, l): order (py, px, c)
__device__ __forceinline__ float tgt_at(const float *target, const PatchGeom &g, int64_t b, int l, int j) {
  const int c = j % g.C, q = j / g.C, px = q % g.p, py = q / g.p;
  const int gy = l / g.gw, gx = l % g.gw;
  return target[((b * g.C + c) * g.H + gy * g.p + py) * g.W + gx * g.p + px];
}

// Per row: mean / unbiased variance of the target patch (when norm_pix), MSE against pred.
// ws[row] = per-patch loss. If dpred != NULL (second pass) writes scale[row-independent] * 2 (pred - tgt) / P * mask.
__global__ __launch_bounds__(256) void rec_loss_rows(const float *pred, const float *target, const float *mask, PatchGeom g,
                                                     int64_t rows, int norm_pix, float *per_row, float *dpred, const float *inv_den) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t b = r / g.L;
    const int l = (int)(r % g.L);
    float mu = 0.f, inv = 1.f;
    if (norm_pix) {
      float s = 0.f;
      for (int j = lane; j < g.P; j += 64) s += tgt_at(target, g, b, l, j);
      mu = wave_sum(s) / (float)g.P;
      float v = 0.f;
      for (int j = lane; j < g.P; j += 64) {
        const float d = tgt_at(target, g, b, l, j) - mu;
        v += d * d;
      }
      v = wave_sum(v) / (float)(g.P - 1);  // torch.var default: unbiased
      inv = 1.0f / sqrtf(v + 1.e-6f);
    }
    const float *pr = pred + r * g.P;
    if (!dpred) {
      float s = 0.f;
      for (int j = lane; j < g.P; j += 64) {
        const float d = pr[j] - (tgt_at(target, g, b, l, j) - mu) * inv;
        s += d * d;
      }
      s = wave_sum(s) / (float)g.P;
      if (lane == 0) per_row[r] = s;
    } else {
      const float sc = (mask ? mask[r] : 1.f) * inv_den[0] * 2.0f / (float)g.P;
      float *dp = dpred + r * g.P;
      for (int j = lane; j < g.P; j += 64) dp[j] = sc * (pr[j] - (tgt_at(target, g, b, l, j) - mu) * inv);
    }
  }
}

// loss = sum(mask*l)/sum(mask)   (mask != NULL)   or mean(l); also stores 1/denominator for the backward pass
__global__ __launch_bounds__(1024) void rec_loss_reduce(const float *per_row, const float *mask, int64_t rows, float *loss, float *inv_den) {
  __shared__ float red[16];
  float a = 0.f, m = 0.f;
  for (int64_t r = threadIdx.x; r < rows; r += blockDim.x) {
    const float w = mask ? mask[r] : 1.f;
    a += w * per_row[r];
    m += w;
  }
  a = block_sum(a, red);
  m = block_sum(m, red);
  if (threadIdx.x == 0) {
    loss[0] = a / m;
    inv_den[0] = 1.0f / m;
  }
}

}  // namespace

extern "C" int evp_rec_loss(const float *pred, const float *target, const float *mask, int B, int C, int H, int W, int patch,
                            int norm_pix, float *loss, float *dpred, float *workspace, void *stream) {
  EVP_CHECK_ARG(pred && target && loss && workspace, EVP_EINVAL, "evp_rec_loss: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_rec_loss: bad shape");
  PatchGeom g;
  g.C = C; g.H = H; g.W = W; g.p = patch; g.gw = W / patch; g.L = (H / patch) * (W / patch); g.P = patch * patch * C;
  EVP_CHECK_ARG(g.P > 1, EVP_ESHAPE, "evp_rec_loss: patch too small");
  const int64_t rows = (int64_t)B * g.L;
  hipStream_t s = (hipStream_t)stream;
  float *per_row = workspace, *inv_den = workspace + rows;
  int64_t grid = (rows + 3) / 4; if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(rec_loss_rows, dim3((int)grid), dim3(256), 0, s, pred, target, mask, g, rows, norm_pix, per_row, (float *)nullptr, inv_den);
  EVP_CHECK_LAUNCH("evp_rec_loss(rows)");
  hipLaunchKernelGGL(rec_loss_reduce, dim3(1), dim3(1024), 0, s, per_row, mask, rows, loss, inv_den);
  EVP_CHECK_LAUNCH("evp_rec_loss(reduce)");
  if (dpred) {
    hipLaunchKernelGGL(rec_loss_rows, dim3((int)grid), dim3(256), 0, s, pred, target, mask, g, rows, norm_pix, per_row, dpred, inv_den);
    EVP_CHECK_LAUNCH("evp_rec_loss(grad)");
  }
  return EVP_OK;
}
