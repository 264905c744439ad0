// Product kernels on the "G4" bodies (gemm_g4_body.h): one wave per SIMD, v_mfma_f32_32x32x16_bf16, LDS-DMA ring of 32-deep stages.
//   evp_gemm_grouped_tn_g4_bf16 : the step's weight gradients, one grouped launch (C ABI, include/evtpretrain.h)
//   evp_g4_gemm                 : forward / data-gradient GEMMs with wide outputs, called by evp_gemm's tile selection (gemm.hip)
#include "gemm_g4_body.h"

namespace {

__global__ __launch_bounds__(256) void gemm_g4_grouped_tn_kernel(const GroupedProblem *__restrict__ probs, const GroupedItem *__restrict__ items,
                                                                 unsigned long long *stamp) {
  const GroupedItem it = items[blockIdx.x];
  if (it.prob < 0) return;                       // padding of the per-XCD item lists
  stamp_begin(stamp, blockIdx.x, gridDim.x);
  const GroupedProblem g = probs[it.prob];
  GemmParams p;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.A = g.A; p.lda = g.lda; p.sA0 = 0; p.sA1 = 0;
  p.B = g.B; p.ldb = g.ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = g.C; p.c_dtype = EVP_F32; p.ldc = g.ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.ldaux = 0;
  p.residual = nullptr; p.ldres = 0; p.accumulate = g.accumulate; p.tiles_m = 0; p.splitk = 1; p.dbg = 0; p.stamp = nullptr; p.c_wt16 = 0;
  p.k_per_split = g.K;
  p.colsum = g.colsum; p.colsum_acc = g.colsum_accumulate;
  gemm_g4_tn_body(p, it.tile_m, it.tile_n);
  stamp_end(stamp, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void gemm_g4_tn_kernel(const GemmParams p) {
  int tile_m, tile_n;
  stamp_begin(p.stamp, blockIdx.x, gridDim.x);
  map_tile(gridDim.x, blockIdx.x, p.tiles_m, tile_m, tile_n);
  gemm_g4_tn_body(p, tile_m, tile_n);
  stamp_end(p.stamp, blockIdx.x, gridDim.x);
}

template <bool AKC, bool BKC, int FI, int FJ, int NST, typename TC, int EPI, int MINW>
__global__ __launch_bounds__(256, MINW) void g4x_kernel(const GemmParams p) {
  stamp_begin(p.stamp, blockIdx.x, gridDim.x);
  // blocks b and b + 8 share an XCD: give every XCD a contiguous run of tiles, walked along N inside one row of tiles, so the
  // workgroups resident on an XCD share one A panel and stream neighbouring B panels through its L2
  const int t_lin = xcd_renumber(gridDim.x, blockIdx.x);
  const int tiles_n = gridDim.x / p.tiles_m;
  g4x_body<AKC, BKC, FI, FJ, NST, TC, EPI>(p, t_lin / tiles_n, t_lin % tiles_n);
  stamp_end(p.stamp, blockIdx.x, gridDim.x);
}

void fill_params(GemmParams &p, const evp_gemm_desc *d) {
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.sA0 = 0; p.sA1 = 0;
  p.B = d->B; p.ldb = d->ldb; p.sB0 = 0; p.sB1 = 0;
  p.C = d->C; p.c_dtype = d->c_dtype; p.ldc = d->ldc; p.sC0 = 0; p.sC1 = 0;
  p.batch1 = 1; p.alpha = d->alpha; p.bias = d->bias; p.act = d->act; p.aux = d->aux; p.ldaux = d->ldaux;
  p.residual = d->residual; p.ldres = d->ldres; p.accumulate = d->accumulate; p.dbg = 0; p.colsum = nullptr; p.colsum_acc = 0;
  p.splitk = 1; p.k_per_split = d->K;
  p.stamp = evp_gemm_next_stamp_slot(); p.c_wt16 = 0;
}

template <bool AKC, bool BKC, int FI, int FJ, int NST, typename TC, int EPI> int launch_g4x(const evp_gemm_desc *d, hipStream_t s, int dbg) {
  GemmParams p;
  fill_params(p, d);
  p.dbg = dbg;
  p.tiles_m = (d->M + 64 * FI - 1) / (64 * FI);
  const int tiles_n = (d->N + 64 * FJ - 1) / (64 * FJ);
  constexpr int smem = NST * (64 * FI + 64 * FJ) * 64;
  constexpr int MINW = NST == 3 ? 2 : 1;           // the 3-stage forms are sized for two workgroups per CU (<= 256 registers)
  auto k = g4x_kernel<AKC, BKC, FI, FJ, NST, TC, EPI, MINW>;
  static bool attr_done = false;                   // one flag per instantiation
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { evp_set_error("evp_gemm(g4): hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e)); return EVP_ELAUNCH; }
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(p.tiles_m * tiles_n)), dim3(256), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm(g4)");
  return EVP_OK;
}

// tile shapes: 20 = 256x256 (4-stage ring, one workgroup per CU), 21 = 256x128, 22 = 128x256 (3-stage ring, two per CU)
template <bool AKC, bool BKC, typename TC, int EPI> int launch_shape(const evp_gemm_desc *d, hipStream_t s, int shape, int dbg) {
  if (shape == 20) return launch_g4x<AKC, BKC, 4, 4, 4, TC, EPI>(d, s, dbg);
  if (shape == 21) return launch_g4x<AKC, BKC, 4, 2, 3, TC, EPI>(d, s, dbg);
  if (shape == 22) return launch_g4x<AKC, BKC, 2, 4, 3, TC, EPI>(d, s, dbg);
  evp_set_error("evp_gemm(g4): unknown tile shape %d", shape);
  return EVP_EINVAL;
}

}  // namespace

// Which evp_gemm calls the G4 forward / data-gradient kernels take: bf16 operands, A row-major, one problem (no batch),
// K % 32 == 0, K >= 96, N % 8 == 0, and one of the built (layout, C type, epilogue) combinations:
//   NT: bf16 C linear / activation forward;  NN: bf16 C linear / activation backward.
bool evp_g4_gemm_supported(const evp_gemm_desc *d) {
  const int64_t nb = (int64_t)(d->batch0 > 0 ? d->batch0 : 1) * (d->batch1 > 0 ? d->batch1 : 1);
  if (d->dtype != EVP_BF16 || d->transA || nb != 1 || d->K % 32 != 0 || d->K < 96 || d->N % 8 != 0 || d->splitk > 1) return false;
  if (d->lda % 8 != 0 || d->ldb % 8 != 0 || d->ldc % 8 != 0 || (d->aux && d->ldaux % 8 != 0)) return false;
  if (((uintptr_t)d->C & 15) != 0 || (d->aux && ((uintptr_t)d->aux & 15) != 0)) return false;
  if (d->c_dtype != EVP_BF16 || d->residual || d->accumulate) return false;
  const bool fwd_act = d->act == EVP_ACT_GELU || d->act == EVP_ACT_RELU, bwd_act = d->act == EVP_ACT_DGELU || d->act == EVP_ACT_DRELU;
  if (fwd_act && d->transB) return false;
  if (bwd_act && (!d->transB || !d->aux)) return false;
  if ((int64_t)d->M * d->lda * 2 >= 0x7FFFFFFFLL || (int64_t)(d->transB ? d->K : d->N) * d->ldb * 2 >= 0x7FFFFFFFLL) return false;
  return true;
}

// Automatic choice (0 = leave the call to the 128x128-class kernels). Measured in the step (bench.py `gemm_kernels_in_step`, in-kernel
// stamps inside the replayed graph) and in tools/gemm_g4_sweep.py: with wide outputs (N >= 1024)
//   * ONE round of 256x256 tiles (<= 256 of them: encoder qkv, 225 tiles) beats the 128x128 body: 29.2 against 34-36 us;
//   * data gradients (B k-strided; with or without GELU') gain 5-20 % on 128x256 tiles at two workgroups per CU
//     (fc2 data gradient 57.0 against 62.5 us in the step; 12544 x 4096 x 768: 92 against 116 us);
//   * forward launches with more than 256 such tiles do not (GELU forward 58.3 against 53.3 us): they stay on 128x128.
int evp_g4_gemm_pick(const evp_gemm_desc *d) {
  if (!evp_g4_gemm_supported(d)) return 0;
  if (d->N < 1024 || d->M < 2048) return 0;
  if (d->act == EVP_ACT_GELU || d->act == EVP_ACT_RELU) return 0;
  const int64_t t256 = (int64_t)((d->M + 255) / 256) * ((d->N + 255) / 256);
  if (t256 <= 256) return 20;
  return d->transB ? 22 : 0;
}

int evp_g4_gemm(const evp_gemm_desc *d, hipStream_t s, int shape, int dbg) {
  if (!evp_g4_gemm_supported(d)) {
    evp_set_error("evp_gemm: tiles 20-22 (G4 forward / data-gradient bodies) need bf16 operands and C, A row-major, no batch, K %% 32 == 0, "
                  "K >= 96, N %% 8 == 0, no residual / accumulate, and activation forward only with transB = 0, backward only with transB = 1");
    return EVP_ESHAPE;
  }
  const int epi = (d->act == EVP_ACT_GELU || d->act == EVP_ACT_RELU) ? 1 : (d->act == EVP_ACT_DGELU || d->act == EVP_ACT_DRELU) ? 2 : 0;
  if (!d->transB) {
    if (epi == 0) return launch_shape<true, true, bf16_t, 0>(d, s, shape, dbg);
    return launch_shape<true, true, bf16_t, 1>(d, s, shape, dbg);
  }
  if (epi == 0) return launch_shape<true, false, bf16_t, 0>(d, s, shape, dbg);
  return launch_shape<true, false, bf16_t, 2>(d, s, shape, dbg);
}

int evp_g4_gemm_tn(const evp_gemm_desc *d, hipStream_t s) {
  GemmParams p;
  fill_params(p, d);
  p.alpha = 1.0f; p.bias = nullptr; p.act = EVP_ACT_NONE; p.aux = nullptr; p.residual = nullptr;
  p.tiles_m = (d->M + 255) / 256;
  const int tiles_n = (d->N + 255) / 256;
  constexpr int smem = 4 * 2 * 32 * 512;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_g4_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { evp_set_error("evp_gemm: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e)); return EVP_ELAUNCH; }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_g4_tn_kernel, dim3((unsigned)(p.tiles_m * tiles_n)), dim3(256), smem, s, p);
  EVP_CHECK_LAUNCH("evp_gemm(g4 tn)");
  return EVP_OK;
}

extern "C" int evp_gemm_grouped_tn_g4_bf16(const void *problems, const void *items, int n_items, void *stream) {
  EVP_CHECK_ARG(problems && items && n_items > 0, EVP_EINVAL, "evp_gemm_grouped_tn_g4_bf16: bad argument");
  auto k = gemm_g4_grouped_tn_kernel;
  constexpr int smem = 4 * 2 * 32 * 512;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_gemm_grouped_tn_g4_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)n_items), dim3(256), smem, (hipStream_t)stream,
                     reinterpret_cast<const GroupedProblem *>(problems), reinterpret_cast<const GroupedItem *>(items), evp_gemm_next_stamp_slot());
  EVP_CHECK_LAUNCH("evp_gemm_grouped_tn_g4_bf16");
  return EVP_OK;
}
