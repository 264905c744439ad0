/* evtpretrain.h -- C-ABI of libevtpretrain.so: the MI355X (gfx950) kernels behind the event pre-training hot
 * path of BIT-Vision/EventPretrain (main_pretrain.py -> trainer/pretrain -> model/pretrain).
 *
 * The reference is 100 % Python on stock ATen ops; it has NO plugin/operator/FFI interface for this path
 * (SURVEY.md 8b). Each entry point below therefore cites the reference *Python* lines whose ATen composition it
 * replaces (paths relative to the reference root). The Python layer in eventpretrain_amd/ binds these with ctypes
 * (eventpretrain_amd/_lib.py) and wraps them in torch.autograd.Function objects (eventpretrain_amd/ops.py).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library never allocates,
 *    frees or keeps a pointer after the call returns).
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no host sync), so calls are
 *    capturable in a HIP graph.
 *  - Return value: EVP_OK (0) or a negative evp_status; evp_last_error() gives a thread-local message. Nothing
 *    throws across the ABI.
 *  - dtype codes: EVP_F32 = 0 (float), EVP_BF16 = 1 (bfloat16, raw uint16 storage).
 *  - Row-major everywhere; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef EVTPRETRAIN_H
#define EVTPRETRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { EVP_OK = 0, EVP_EINVAL = -1, EVP_ESHAPE = -2, EVP_ELAUNCH = -3, EVP_EUNSUPPORTED = -4 } evp_status;
enum { EVP_F32 = 0, EVP_BF16 = 1 };
enum { EVP_ACT_NONE = 0, EVP_ACT_GELU = 1, EVP_ACT_DGELU = 2, EVP_ACT_RELU = 3, EVP_ACT_DRELU = 4 };

const char *evp_last_error(void);
/* ABI version of this header; bumped on any signature change (2: evp_dropout_fwd takes a device-side seed; evp_gemm_desc lost its
 * stream-K workspace fields in round 3; 3: evp_events_draw_erase_add; 4: keep flags on the window attention; 5: evp_events_plan_batch). A host binding must compare evp_abi_version() with EVP_ABI_VERSION at load time. */
#define EVP_ABI_VERSION 5
int evp_abi_version(void);
/* Name of the code object's target ("gfx950"). */
const char *evp_target_arch(void);

/* ------------------------------------------------------------------------------------------------ K1 voxel
 * Replaces dataset/dataset_utils/events_to_voxel_grid.py:4-61 (two index_add_ scatters, :46-57) for a BATCH of
 * clips. events: float64 [n_total,4], columns (x,y,t,p) or (t,x,y,p) if is_txyp; clip c owns rows
 * [clip_offsets[c], clip_offsets[c+1]). out: float32 [n_clips,bins,H,W], fully overwritten.
 * assume_sorted == 1: rows of each clip are non-decreasing in t (what every reference dataset produces; the
 * reference itself relies on it for t0/t1, :19-22) -> each block scans only its time slab; NOT checked.
 * assume_sorted == 2 (algo 0 / 3): the same fast path, VERIFIED on the device with no host sync: while binning, every row is
 * checked to lie in the time slab its position says (the one property of sortedness the slab schedule uses); clips that
 * fail are redone by a repair pass that scans the whole clip, the others cost one extra empty launch. The result is the
 * reference's for any row order. workspace then also holds int32 [n_clips] flags after the cuts.
 * assume_sorted == 0: correct for any row order (every block scans the whole clip).
 * n_events_total = clip_offsets[n_clips] (known to the host that built the offsets).
 * algo 0 (default) = single-pass LDS-binned straight from the float64 rows, workspace int64 [n_clips*(bins+2) + (n_clips+1)/2];
 * algo 3 = the same with float64 cells in the LDS tile (`ds_add_f64`; an order-independent sum rounded to float32 once; three y-tiles
 * instead of two at 224 x 224: measured 1.2x slower, kept for A/B).
 * algo 2 = decode-once two-pass form: pass A packs every event into 12 bytes (pixel/bin key + the two float32
 * contributions), pass B bins the packed streams; workspace int64 [n_clips*(bins+2) + (3*n_events_total + 1)/2]
 * (measured 1.3x slower than algo 0 on MI355X; kept for A/B). algo 1 = memset + global float atomics, no workspace
 * (5.3x slower).
 * tile_rows: rows of one bin plane held in LDS per block (0 = auto). */
int evp_voxel_scatter_f32(const double *events, const int64_t *clip_offsets, int n_clips, int64_t n_events_total, int bins,
                          int H, int W, int is_txyp, int assume_sorted, int algo, int tile_rows, int64_t *workspace,
                          float *out, void *stream);
/* Measurement aid (tools/voxel_parts.py): 1 = no LDS atomics, 2 = no time normalisation, 3 = rows are only loaded; results are garbage
 * while set. Returns the previous value; 0 = off. */
int evp_voxel_set_debug(int v);
/* The same with the loader's sensor -> input rescale fused in (reference dataset/augmentation/events_augment.py:22-26,
 * `events[:,0] *= input_w/sensor_w; events[:,1] *= input_h/sensor_h`, applied by pr_n_imagenet_dataset.py:85-86 between
 * the event-level augmentation and the voxelisation): x and y are multiplied by scale_x / scale_y in float64 before
 * the truncation, bit-identical to rescaling the array first. scale 1.0 = evp_voxel_scatter_f32. */
int evp_voxel_scatter_scaled_f32(const double *events, const int64_t *clip_offsets, int n_clips, int64_t n_events_total,
                                 int bins, int H, int W, int is_txyp, int assume_sorted, int algo, int tile_rows,
                                 double scale_x, double scale_y, int64_t *workspace, float *out, void *stream);
/* Event-level augmentation, data movement (reference dataset/augmentation/events_augment.py:28-55
 * `erase_and_add_events`): for every clip drop the rows erase_idx lists, add the rows
 * events[add_idx[j]] + add_noise[j] (x, y, t; x clipped to [0, sensor_w-1], y to [0, sensor_h-1]; p copied) and keep the
 * clip time-sorted. Input clips must be time-sorted (x,y,t,p) rows; the random decisions are the caller's, drawn in the
 * reference's order (host side: eventpretrain_amd/dataset/augmentation/events_augment.py).
 * erase_idx: int64, clip-relative, strictly ascending inside [erase_offsets[c], erase_offsets[c+1]);
 * add_idx: int64 clip-relative [A_total]; add_noise: float64 [A_total,3]; add_offsets: int64 [n_clips+1];
 * max_add_per_clip <= 8192 (the added rows of a clip are sorted in LDS); add_rows_ws: float64 [A_total,4];
 * out_offsets: int64 [n_clips+1], out_offsets[c+1]-out_offsets[c] = n_c - erased_c + added_c;
 * out_events: float64 [out_offsets[n_clips],4]. Rows with equal stamps: original rows first (numpy's argsort leaves
 * their order unspecified). */
int evp_events_erase_add_f64(const double *events, const int64_t *clip_offsets, int n_clips, const int64_t *erase_idx,
                             const int64_t *erase_offsets, const int64_t *add_idx, const double *add_noise,
                             const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h,
                             double *add_rows_ws, const int64_t *out_offsets, double *out_events, void *stream);
/* The same on a WINDOW of every clip: clip c is rows [win_begin[c], win_end[c]) of `events` (int64 [n_clips] each) -- the loader's
 * `get_random_index` pick (events_augment.py:5-20, pr_n_imagenet_dataset.py:83-84) taken on clips already resident in HBM; all
 * indices stay relative to the window. The merge pass compacts the windows into out_events while it erases / adds. */
int evp_events_erase_add_win_f64(const double *events, const int64_t *win_begin, const int64_t *win_end, int n_clips,
                                 const int64_t *erase_idx, const int64_t *erase_offsets, const int64_t *add_idx, const double *add_noise,
                                 const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows_ws,
                                 const int64_t *out_offsets, double *out_events, void *stream);
/* Only the ADDED rows of the above (events_augment.py:40-49): add_rows[add_offsets[c] + r] = the copy of window row add_idx[.] of clip c plus
 * its noise, x / y clipped to the sensor, the clip's rows sorted by stamp -- what evp_voxel_scatter_fused_f32 takes beside the window. */
int evp_events_build_added_f64(const double *events, const int64_t *win_begin, int n_clips, const int64_t *add_idx, const double *add_noise,
                               const int64_t *add_offsets, int max_add_per_clip, double sensor_w, double sensor_h, double *add_rows,
                               void *stream);
/* The DECISIONS of erase_and_add_events drawn on the device (events_augment.py:31-44: which rows to erase, which to copy, the three
 * normal noise columns of the copies): clip c (rows [win_begin[c], win_end[c]) of some event array, n_c of them) gets
 * erase_offsets[c+1] - erase_offsets[c] DISTINCT rows in erase_idx (ascending, clip-relative) and add_offsets[c+1] - add_offsets[c]
 * distinct rows in add_idx (draw order) with add_noise [.,3] ~ N(0,1.5), N(0,1.5), N(0,0.001) -- uniform draws without replacement
 * from Philox4x32-10 keyed by (seed, step, first_sample + c). The COUNTS (the offsets) are the caller's: two numbers per clip, uniform
 * in [int(0.001 n), int(0.01 n)). max_per_clip = the largest count of any list (<= ~7200; an upper bound will do). step_first_dev
 * (optional, device int64[2]) overrides (step, first_sample) at run time, so that one captured HIP graph serves every batch. Feeds
 * evp_events_erase_add(_win)_f64. */
int evp_events_draw_erase_add(const int64_t *win_begin, const int64_t *win_end, int n_clips, const int64_t *erase_offsets,
                              const int64_t *add_offsets, uint64_t seed, uint64_t step, int64_t first_sample, const int64_t *step_first_dev,
                              int max_per_clip, int64_t *erase_idx, int64_t *add_idx, double *add_noise, void *stream);
/* The per-batch PLAN of the loader chain computed on the device -- what the host otherwise prepares per batch from the same counter stream
 * (get_random_index's window start, events_augment.py:5-20; the two counts of erase_and_add_events, :31-38; evg_augment's / frame_augment's
 * crop box and flip coins, view_augment.py:9-77): for clip c with n_c = clip_offsets[c+1] - clip_offsets[c] rows (device int64 [n_clips+1])
 *   tabs int64 [5][n_clips+1]: window begin / end (absolute rows), exclusive prefix sums of the erase counts, the add counts and the rows out;
 *   params / frame_params int32 [n_clips,6] = (x0, y0, w, h, hflip, tflip) for a grid_h x grid_w view / a frame_h x frame_w frame (same
 *   uniforms, the frame's time flip = the grid's; frame_params may be NULL).
 * state: device int64[2] = (step, first sample); copied to step_first_out (what evp_events_draw_erase_add's step_first_dev reads for this
 * batch) and, with advance != 0, step += 1 -- a captured HIP graph of the chain then needs nothing from the host per batch. */
int evp_events_plan_batch(const int64_t *clip_offsets, int n_clips, int64_t fix_events_num, uint64_t seed, int64_t *state, int advance,
                          int64_t *step_first_out, int grid_h, int grid_w, int frame_h, int frame_w, double crop_min, int64_t *tabs,
                          int32_t *params, int32_t *frame_params, void *stream);
/* sorted_flags[c] = 1 if clip c's stamps are non-decreasing, else 0 (device int32 [n_clips]). */
int evp_events_sorted_check(const double *events, const int64_t *clip_offsets, int n_clips, int is_txyp,
                            int32_t *sorted_flags, void *stream);

/* K1 fused with the loader's event-level augmentation (events_augment.py:28-61 erase_and_add_events followed by events_reshape and
 * events_to_voxel_grid): the voxel grids of the clips [win_begin[c], win_end[c]) of `events` MINUS the rows erase_idx (window-relative,
 * ascending, erase_offsets [n_clips+1]) PLUS added_rows (float64 [.,4] x,y,t,p, time-sorted per clip as evp_events_erase_add_*'s workspace
 * holds them, add_offsets [n_clips+1]) -- without writing the merged clip: the grid is a sum over the kept rows, the order only decides
 * t0 / t1 (min / max stamp of the merged clip). max_window >= every window's rows (<= 393216: one bit per row in LDS). Window stamps
 * non-decreasing is verified on the device; a clip that fails is redone by a full scan. workspace: n_clips * (bins + 5) int64.
 * view_params (optional, device int32 [n_clips,6] as evp_view_augment_f32 takes them): the grids leave THROUGH the view augmentation --
 * out is then float32 [n_clips, bins, view_h, view_w] = evp_view_augment_f32 of the grids, which are never stored; NULL: out holds the
 * grids [n_clips, bins, H, W]. */
int evp_voxel_scatter_fused_f32(const double *events, const int64_t *win_begin, const int64_t *win_end, int n_clips,
                                const int64_t *erase_idx, const int64_t *erase_offsets, const double *added_rows,
                                const int64_t *add_offsets, int64_t max_window, int bins, int H, int W, double scale_x, double scale_y,
                                const int32_t *view_params, int view_h, int view_w, int negate_on_time_flip, int64_t *workspace, float *out,
                                void *stream);

/* ------------------------------------------------------------------------------------------------ K2 masking
 * Replaces ViT.random_masking, model/backbone/vit.py:91-103 (argsort, argsort, slice, gather) with the noise as an
 * explicit input. Stable ascending order (ties: lower index first; NaN last). noise float32 [B,L] ->
 * ids_keep int64 [B,len_keep], mask float32 [B,L] (1 = removed), ids_restore int64 [B,L]. L <= 4096. */
int evp_mask_from_noise(const float *noise, int B, int L, int len_keep, int64_t *ids_keep, float *mask,
                        int64_t *ids_restore, void *stream);
/* vit.py:80-89 density / anti-density noise: |sum_c x| average-pooled over patch x patch; sign = +1 / -1.
 * x float32 [B,C,H,W] -> noise float32 [B,(H/p)*(W/p)]. */
int evp_density_noise(const float *x, int B, int C, int H, int W, int patch, float sign, float *noise, void *stream);

/* ------------------------------------------------------------------------------------------------ GEMM family
 * One batched, strided GEMM:  C[b0,b1] = epilogue(alpha * A[b0,b1] (M x K) . B[b0,b1]^T (N x K))
 *   transA = 0: A stored [M][K] (K contiguous);  transA = 1: A stored [K][M].
 *   transB = 0: B stored [N][K] (nn.Linear weight layout);  transB = 1: B stored [K][N].
 * Replaces every nn.Linear / matmul / einsum on the path: model/sub_module/vit_block.py:133,136,140,141,226,230,
 * model/pretrain/pr_rec_decoder.py:54,68, the Conv2d-as-GEMM of vit_block.py:65, mlp_head.py:10 and the logits of
 * model/pretrain/pr_hub_model.py:153-155,179, plus their autograd backward (dgrad: transB=1, wgrad: transA=transB=1).
 * dtype is the A/B element type (EVP_F32 -> exact f32 MFMA, EVP_BF16 -> bf16 MFMA, f32 accumulate).
 * Epilogue, in order: v = alpha*acc; v += bias[n]; if aux (ACT_GELU/RELU) aux[m,n] = v; v = act(v) or
 * v *= act'(aux[m,n]) (ACT_DGELU/DRELU); v += residual[m,n]; if accumulate C += v else C = v.
 * Contiguous dimensions of A, B, C, aux and residual must be multiples of 8 elements (16-byte chunks for bf16)
 * except M/N/K edges handled by predication as documented in DESIGN.md. */
typedef struct {
  int dtype, transA, transB;
  int M, N, K;
  const void *A; int64_t lda, strideA0, strideA1;
  const void *B; int64_t ldb, strideB0, strideB1;
  void *C; int c_dtype; int64_t ldc, strideC0, strideC1;
  int batch0, batch1;
  float alpha;
  const float *bias;
  int act;
  void *aux; int64_t ldaux;            /* element type = c_dtype; batch strides = C's */
  const float *residual; int64_t ldres; /* batch strides = C's */
  int accumulate;
  int tile;                             /* 0 auto; 1 = 128x128, 2 = 64x64, 4 = 96x128 (LDS-staged 16x16x32 body, two workgroups per CU);
                                         * 9 = G4 256x256 (bf16 TN, f32 C, K % 32 == 0); 20 / 21 / 22 = G4 forward / data-gradient bodies
                                         * 256x256 / 256x128 / 128x256 (bf16 operands and C, A row-major, K % 32 == 0, K >= 96, N % 8 == 0,
                                         * linear or activation epilogue, no residual). Auto sends 257..384-tile outputs to 96x128, small outputs
                                         * to 64x64, the rest to 128x128 (and, with evp_gemm_set_variant(11), wide outputs to the G4 bodies). */
  int splitk;                           /* 0 auto, n = cut K into n slices summed with f32 atomics (plain f32 C only) */
} evp_gemm_desc;
int evp_gemm(const evp_gemm_desc *d, void *stream);
/* Grouped weight-gradient GEMM: n problems C_g[M_g,N_g] (f32) = A_g^T . B_g, A_g stored [K_g][M_g] and B_g stored
 * [K_g][N_g] (bf16), in ONE launch -- the deferred dW = dY^T . X of every Linear of the step (autograd backward of
 * vit_block.py:133,141,226,230 etc.). `problems` is a device array of
 *   struct { const void *A, *B; void *C; int M, N, K; int lda, ldb, ldc; int accumulate, colsum_accumulate;
 *            float *colsum; }   (64 bytes each; accumulate != 0: C_g += ...; colsum: see the 256x256 entry, NULL here)
 * and `items` a device array of  struct { int prob, tile_m, tile_n, pad; }  listing every 128x128 output tile
 * (prob < 0 = padding entry, skipped: lets the caller lay the list out per XCD, workgroups i and i+8 share an L2). */
int evp_gemm_grouped_tn_bf16(const void *problems, const void *items, int n_items, void *stream);
/* Same problem table, but `items` lists 256x256 output tiles, every K_g a multiple of 32 and >= 96: the "G4" body -- 4 waves, one per SIMD, each holding
 * 128x128 of the tile in 256 accumulator registers (v_mfma_f32_32x32x16_bf16), 32-deep stages in a four-slot LDS-DMA ring,
 * one barrier per stage. The default for the step's weight gradients (dW = dY^T X of every nn.Linear on the path,
 * model/sub_module/vit_block.py:131-143,225-231). Items with prob < 0 are skipped (padding of per-XCD lists).
 * A non-NULL `colsum` (float32 [M_g]) also receives colsum[m] (+)= sum_k A_g[k][m] -- the bias gradient db = sum over
 * rows of dY of the same Linear -- computed from the A fragments already in registers (no second pass over dY). */
int evp_gemm_grouped_tn_g4_bf16(const void *problems, const void *items, int n_items, void *stream);
/* out[i] (+)= sum_s ws[s*numel + i], float32, numel % 4 == 0: reduction of split-K partials when a long-K problem was
 * entered into the grouped launch as several K-slice problems writing to a workspace (ConvViT stage 1: K = B*56*56). */
int evp_sum_slices_f32(const float *ws, float *out, int n_slices, int64_t numel, int accumulate, void *stream);
/* Tuning switches for A/B measurements (results are identical up to f32 summation order): 1 = LDS-DMA (buffer_load ... lds)
 * staging for bf16 (default), 2 = register staging; 10 = wide forward / data-gradient GEMMs on 128x128 tiles (default: the G4
 * bodies lose inside the replayed step), 11 = on the G4 bodies, 12 / 13 = only the one-round 256x256 forward
 * tiles / only the 128x256 data-gradient tiles. Returns the previous staging variant; any other argument only queries. */
int evp_gemm_set_variant(int v);
/* Measurement aid: in-kernel wall-clock stamps. `buf` = device uint64 [n_slots][2 * 4096] (NULL switches stamping off and resets
 * the slot counter): every following GEMM launch (evp_gemm, the grouped entries) takes the next slot (mod n_slots) and each of its
 * workgroups b < 4096 writes s_memrealtime (100 MHz) to [2 b] when it starts and to [2 b + 1] after its last store was acknowledged.
 * max(end) - min(start) of a slot = that launch's duration where it ran -- also inside a replayed HIP graph (bench.py `roofline`).
 * evp_gemm_stamp_count() = launches stamped since the buffer was installed. */
int evp_gemm_set_stamp_buffer(void *buf, long long n_slots);
long long evp_gemm_stamp_count(void);

/* ------------------------------------------------------------------------------------------------ K4/K9 LayerNorm
 * Replaces nn.LayerNorm over the last dim (vit_block.py:247,249; vit.py:126-128 with the 3-tap sum fused:
 * y = LN(x + x2 + x3), x2/x3 may be NULL). x*: float32 [M,D]; y: y_dtype [M,D]; mean/rstd float32 [M]. */
int evp_layernorm_fwd(const float *x, const float *x2, const float *x3, const float *gamma, const float *beta,
                      int64_t M, int D, float eps, void *y, int y_dtype, float *mean, float *rstd, void *stream);
/* dx = (gres ? gres : 0) + LN'(dy): dy dy_dtype [M,D]; x.. as forward; dx float32 [M,D]; dx_lp (optional, bf16) a
 * low-precision copy of dx for the next GEMM; dgamma/dbeta float32 [D] are OVERWRITTEN (partials: workspace float32
 * [2*nblk*D], nblk = evp_layernorm_bwd_nblk(M)). With dgamma == dbeta == NULL the final reduction is skipped and the
 * per-block partials stay in workspace as [nblk][2][D] (dgamma rows first) for the caller to column-sum, e.g. with
 * the step's one evp_colsum_grouped launch. */
int evp_layernorm_bwd_nblk(int64_t M);
int evp_layernorm_bwd(const void *dy, int dy_dtype, const float *x, const float *x2, const float *x3,
                      const float *gamma, const float *mean, const float *rstd, const float *gres, int64_t M, int D,
                      float *dx, void *dx_lp, float *dgamma, float *dbeta, float *workspace, void *stream);
/* The same with a third partial row per block: workspace [nblk][3][D] = {dgamma, dbeta, column sums of the OUTPUT dx}. dx is
 * the gradient of the f32 residual stream, and its column sum is the bias gradient of the Linear that wrote into the
 * stream (vit_block.py:139,230: proj / fc2): the grouped column-sum launch then reduces nblk rows instead of re-reading the
 * whole gradient. No finalize: the caller reduces all three partial rows (evp_colsum_grouped). */
int evp_layernorm_bwd_cs(const void *dy, int dy_dtype, const float *x, const float *x2, const float *x3, const float *gamma,
                         const float *mean, const float *rstd, const float *gres, int64_t M, int D, float *dx, void *dx_lp,
                         float *workspace, void *stream);

/* column sums: out[n] (+)= sum_m x[m,n]  (bias gradients). x dtype [M,N] with ld. workspace float32 [nblk*N],
 * nblk = evp_colsum_nblk(M). */
int evp_colsum_nblk(int64_t M);
int evp_colsum(const void *x, int x_dtype, int64_t M, int N, int64_t ld, float *out, float *workspace, void *stream);

/* Grouped column sums: out_g[n] += sum_m x_g[m,n] for many tensors in one launch (the deferred bias gradients of a
 * step). problems: device array of struct { const void *x; float *out; int64 M; int N, ld, dtype, pad; } (40 bytes);
 * items: device array of struct { int prob, col_block (128 columns), row_slab (evp_colsum rows per slab = 256),
 * pad; }. The outputs must be zeroed by the caller (they accumulate with f32 atomics). */
int evp_colsum_grouped(const void *problems, const void *items, int n_items, void *stream);

/* ------------------------------------------------------------------------------------------------ K6 attention core
 * Replaces vit_block.py:134-140 on the packed qkv tensor ([B,N,3,h,dh], element type dtype) of the fused qkv Linear:
 * scores = q k^T * scale (batched GEMM), row softmax, out = probs v. probs: dtype [B,h,N,ldp] (ldp = N rounded up
 * to 8, pad columns zero) is both the saved tensor for backward and the tensor the dense branch returns
 * (vit.py:144). out: dtype [B,N,h*dh]. */
int evp_attention_fwd(const void *qkv, int dtype, int B, int N, int heads, int dh, float scale, float *scores_ws,
                      void *probs, int64_t ldp, void *out, void *stream);
/* dqkv (dtype, [B,N,3,h,dh]) from dout (dtype [B,N,h*dh]). Scratch: scores_ws / dp_ws float32 [B,h,N,ldp] (the
 * logits are kept in f32 between the GEMM and the softmax in both precisions); ds_ws dtype [B,h,N,ldp]. */
int evp_attention_bwd(const void *qkv, const void *probs, const void *dout, int dtype, int B, int N, int heads, int dh,
                      float scale, int64_t ldp, float *dp_ws, void *ds_ws, void *dqkv, void *stream);
/* Fused form of the same core for the shapes of this path (bf16, N <= 224 tokens, dh in {32,64}): one workgroup per
 * (batch, head), scores stay in registers, softmax statistics are saved as the per-row log-sum-exp
 * (lse float32 [B,h,N]) instead of the N x N probabilities; probs (bf16 [B,h,N,ldp], may be NULL) is written only
 * when the caller needs the attention map (dense branch, vit.py:144). The backward recomputes the probabilities from
 * qkv and lse; out is the forward output (needed for rowsum(dout*out)). */
int evp_attention_fused_supported(int dtype, int N, int dh);
/* Measurement aid: device buffer uint64 [B*heads*4*2] that evp_attention_fused_fwd fills per wave with {staging, compute}
 * cycles; NULL (default) switches it off. */
int evp_attention_set_debug_buffer(void *buf);
int evp_attention_fused_fwd(const void *qkv, int B, int N, int heads, int dh, float scale, void *out, float *lse,
                            void *probs, int64_t ldp, void *stream);
int evp_attention_fused_bwd(const void *qkv, const void *out, const void *dout, const float *lse, int B, int N,
                            int heads, int dh, float scale, void *dqkv, void *stream);
/* p[r,:] = softmax(s[r,:n_valid]) with s float32 [rows,ld], p dtype [rows,ld]; pad columns [n_valid,ld) = 0. */
int evp_softmax_rows(const float *s, void *p, int dtype, int64_t rows, int n_valid, int64_t ld, void *stream);
/* ds = p * (dp - rowsum(p*dp)); dp float32, p/ds dtype; pad columns = 0. */
int evp_softmax_rows_bwd(const void *p, const float *dp, void *ds, int dtype, int64_t rows, int n_valid, int64_t ld,
                         void *stream);

/* ------------------------------------------------------------------------------------------------ K3 patch embed
 * vit_block.py:60-68 + vit.py:110-115. Non-overlapping patches make Conv2d(k=s=p) a GEMM on a patch matrix:
 * evp_patchify gathers ONLY the kept tokens (ids_keep int64 [B,n_keep], NULL = all L tokens in order) of
 * x float32 [B,C,H,W] into cols: dtype [B*n_keep, C*p*p], inner order (c,py,px) = Conv2d weight.view(D,-1). */
int evp_patchify(const float *x, const int64_t *ids_keep, int B, int C, int H, int W, int patch, int n_keep,
                 void *cols, int dtype, void *stream);
/* tokens = GELU(LN_{eps}(y)) + pos[token id]: y float32 [M,D] (conv output incl. bias), pos float32 [L,D],
 * ids_keep as above (NULL = identity, M = B*L). Saves mean/rstd. out float32 [M,D]. */
int evp_embed_post_fwd(const float *y, const float *gamma, const float *beta, const float *pos, const int64_t *ids_keep,
                       int B, int n_keep, int L, int D, float eps, float *out, float *mean, float *rstd, void *stream);
/* dy (dtype [M,D]) = LN'(GELU'(LN(y)) * g); dgamma/dbeta overwritten (workspace as evp_layernorm_bwd); both NULL: the per-block
 * partials stay in workspace [nblk][2][D] for the caller's grouped column sum, as with evp_layernorm_bwd. */
int evp_embed_post_bwd(const float *g, const float *y, const float *gamma, const float *beta, const float *mean,
                       const float *rstd, int64_t M, int D, void *dy, int dy_dtype, float *dgamma, float *dbeta,
                       float *workspace, void *stream);

/* out[b,j,:] = x[b,j,:] + table[ids[b,j],:] (ids NULL: j): the `x + pos_embed` then gather of convvit.py:158-159 with
 * the gather done first. x/out float32 [B,n,D], table float32 [L,D]. */
int evp_add_rows_gather_f32(const float *x, const float *table, const int64_t *ids, int B, int n, int L, int D, float *out,
                            void *stream);

/* ------------------------------------------------------------------------------------------------ K17 ConvViT stages
 * Stage-1/2 feature maps are kept channels-last ([B,H,W,C] float32 tokens). A Conv2d with kernel = stride = p
 * (PatchEmbed convvit.py:20-25; fusion convs :48-49) is a GEMM on the gathered patch matrix:
 * cols[(b,j), c*p*p + py*p + px] = x[b, gy*p+py, gx*p+px, c] for token ids_keep[b,j] (NULL = all tokens in order);
 * evp_unpatchify_nhwc is its adjoint (zero-fills dropped patches unless accumulate). */
int evp_patchify_nhwc(const float *x, const int64_t *ids_keep, int B, int H, int W, int C, int patch, int n_keep, void *cols,
                      int dtype, void *stream);
int evp_unpatchify_nhwc(const void *dcols, int dtype, const int64_t *ids_keep, int B, int H, int W, int C, int patch,
                        int n_keep, int accumulate, float *dx, void *stream);
/* Depthwise 5x5 convolution, padding 2, groups = C (conv_block.py:30,43-46) on a channels-last map of `dtype`, with
 * ConvBlock's keep-mask multiply fused into the input read: in_eff = (1 - mask[b, (y/s)*(W/s) + x/s]) * in, s =
 * mask_scale (mask NULL: no masking). w float32 [C,1,5,5], bias float32 [C]. */
int evp_dwconv5x5_fwd(const void *in, int dtype, const float *mask, int mask_scale, const float *w, const float *bias, int B,
                      int H, int W, int C, void *out, void *stream);
/* din (dtype), dw float32 [C,25], dbias float32 [C] (both overwritten); workspace float32
 * [evp_dwconv5x5_bwd_nslab(B,H,W) * 26 * C]. `in` is the un-masked forward input. */
int evp_dwconv5x5_bwd_nslab(int B, int H, int W);
/* A/B switch (process-wide): 1 (default) = LDS-band kernels where the shape allows (H % 4 == 0, 8 <= W <= 64, C a multiple of
 * 64 (bf16) / 32 (f32)), 0 = the register row walkers everywhere. Results agree to f32 summation order. */
int evp_dwconv_set_band(int on);
int evp_dwconv5x5_bwd(const void *dout, const void *in, int dtype, const float *mask, int mask_scale, const float *w, int B,
                      int H, int W, int C, void *din, float *dw, float *dbias, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------ K10 decoder unshuffle
 * pr_rec_decoder.py:56-62: out[b,l,:] = (ids_restore[b,l] < n_keep ? emb[b,ids_restore[b,l],:] : mask_token) + pos[l].
 * emb float32 [B,n_keep,D] -> out float32 [B,L,D]. */
int evp_unshuffle_fwd(const float *emb, const float *mask_token, const float *pos, const int64_t *ids_restore, int B,
                      int n_keep, int L, int D, float *out, void *stream);
/* demb[b,ids_restore[b,l],:] = g[b,l,:] for kept positions (ids_restore[b,l] < n_keep); dmask_token[d] = sum of g
 * over removed positions (overwritten; workspace float32 [evp_colsum_nblk(B*L)*D]). */
int evp_unshuffle_bwd(const float *g, const int64_t *ids_restore, int B, int n_keep, int L, int D, float *demb,
                      float *dmask_token, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------ K12 reconstruction loss
 * pr_hub_model.py:125-141 + utils/reshape.py:15-22: patchify target (B,C,H,W) -> (B,L,p*p*C) order (py,px,c);
 * per-patch normalise with UNBIASED variance (+1e-6) when norm_pix; per-patch MSE; loss = sum(mask*l)/sum(mask)
 * (mask NULL or mask_ratio==0 path: mean). pred: float32 [B,L,P]. loss: float32 [1]. dpred (optional, float32
 * [B,L,P]) = d loss / d pred (scaled by upstream gradient 1). workspace float32 [2*B*L + 4]. */
int evp_rec_loss(const float *pred, const float *target, const float *mask, int B, int C, int H, int W, int patch,
                 int norm_pix, float *loss, float *dpred, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------ small helpers
 * out = a + b (+ c) float32, n elements. */
int evp_add_f32(const float *a, const float *b, const float *c, int64_t n, float *out, void *stream);
int evp_cast(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, void *stream);
/* x[i] *= scalar[0] with the scalar in device memory (upstream gradient of a scalar loss, no host sync). */
int evp_scale_f32(float *x, const float *scalar, int64_t n, void *stream);
/* dst[c][r] = src[r][c] for a [rows, cols] matrix of `dtype` (used for weight shadows). */
int evp_transpose(const void *src, void *dst, int dtype, int64_t rows, int64_t cols, void *stream);

/* ------------------------------------------------------------------------------------------------ K21 optimiser
 * torch.optim.AdamW step as main_pretrain.py:341-343 configures it, over a list of tensors in ONE launch.
 * Per-tensor tables (device memory, one entry per tensor): params/grads/exp_avg/exp_avg_sq float32*, lp_shadow
 * (bf16*, table or entries may be NULL) receives the updated weights rounded to bf16, numel int64, weight_decay
 * float32, lr_scale float32. Work is cut into chunks of chunk_elems (multiple of 4) elements: chunk i updates
 * tensor chunk_tensor[i] from element chunk_offset[i] (host-built once per parameter set). step is the 1-based
 * step count; grad_scale multiplies every gradient first (1/accum or 1/world). dev_hyper (optional, device float32
 * [4] = {1-beta1^step, sqrt(1-beta2^step), grad_scale, lr multiplier}) overrides the host scalars so that a
 * captured HIP graph can be replayed with per-step values refreshed by a tiny H2D copy. */
int evp_adamw_multi(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                    uint16_t *const *lp_shadow, const int64_t *numel, const float *weight_decay, const float *lr_scale,
                    const int32_t *chunk_tensor, const int64_t *chunk_offset, int n_chunks, int chunk_elems, float lr,
                    float beta1, float beta2, float eps, int step, float grad_scale, const float *dev_hyper,
                    void *stream);
/* utils/misc.py:303-315: total 2-norm over a tensor list -> out float32 [1]. workspace float32 [n_chunks]. */
int evp_grad_norm_multi(const float *const *grads, const int64_t *numel, const int32_t *chunk_tensor,
                        const int64_t *chunk_offset, int n_chunks, int chunk_elems, float *workspace, float *out,
                        void *stream);

/* ------------------------------------------------------------------------------------------------ K13 BatchNorm on tokens
 * mlp_head.py:13,18 applied as pr_hub_model.py:223-237 does: per-channel batch statistics over the B*L rows of
 * x (dtype [R,C]). Training mode only (the pre-training loop never evaluates). gamma/beta may be NULL
 * (affine=False). y dtype [R,C] (relu != 0 fuses the following ReLU); saves mean/invstd float32 [C]; updates
 * running_mean/var (momentum 0.1, unbiased var) in place. workspace float32 [(2*evp_batchnorm_nblk(R)+2)*C]. */
int evp_batchnorm_fwd(const void *x, int dtype, int64_t R, int C, const float *gamma, const float *beta, float eps,
                      float momentum, int relu, void *y, float *mean, float *invstd, float *running_mean,
                      float *running_var, float *workspace, void *stream);
int evp_batchnorm_bwd(const void *dy, const void *x, const void *y, int dtype, int64_t R, int C, const float *gamma,
                      const float *mean, const float *invstd, int relu, void *dx, float *dgamma, float *dbeta,
                      float *workspace, void *stream);
int evp_batchnorm_nblk(int64_t R);

/* ------------------------------------------------------------------------------------------------ K15/K16 InfoNCE
 * F.normalize(x, dim=-1) rows (pr_hub_model.py:145-146,172-173): y = x / max(||x||, 1e-12); saves norm. */
int evp_l2norm_rows_fwd(const float *x, int64_t R, int C, float *y, float *norm, void *stream);
int evp_l2norm_rows_bwd(const float *dy, const float *y, const float *norm, int64_t R, int C, float *dx, void *stream);
/* Cross entropy over rows of logits float32 [R,ld] (n_cls valid columns), integer label per row (int64 [R]):
 * loss[0] = mean_r(-log softmax(logits[r])[label[r]]) and, if dlogits != NULL, dlogits = (softmax - onehot)/R.
 * workspace float32 [R]. */
int evp_cross_entropy(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld, float *loss,
                      float *dlogits, float *workspace, void *stream);
/* The same with label smoothing s in [0,1): timm 0.3.2 `LabelSmoothingCrossEntropy(smoothing=s)` as the reference's
 * fine-tune loop applies it (trainer/finetune_cls/ft_cls_trainer.py:63-64; timm is not vendored in the reference, its
 * published formula is restated): per row (1-s)*(-log p[label]) + s*(-mean_j log p[j]), mean over rows;
 * dlogits = (softmax - ((1-s)*onehot + s/n_cls))/R. s = 0 is evp_cross_entropy. */
int evp_cross_entropy_smooth(const float *logits, const int64_t *labels, int64_t R, int n_cls, int64_t ld, float smoothing,
                             float *loss, float *dlogits, float *workspace, void *stream);

/* out[r] = <a[r,:], b[r,:]> (the positive logits l_pos of pr_hub_model.py:151) and out[r,c] = s[r]*x[r,c] (+ add). */
int evp_rowdot_f32(const float *a, const float *b, int64_t R, int C, float *out, void *stream);
int evp_scale_rows_f32(const float *x, const float *s, const float *add, int64_t R, int C, float *out, void *stream);
/* InfoNCE against the per-position queue (pr_hub_model.py:150-160): row r has logits [pos[r], neg[r,0..K)] * inv_T,
 * positive at class 0; loss[0] = mean CE; optional dpos float32 [R], dneg float32 [R,ldn] (pad columns zeroed).
 * workspace float32 [R]. */
int evp_infonce_queue(const float *pos, const float *neg, int64_t R, int K, int64_t ldn, float inv_T, float *loss,
                      float *dpos, float *dneg, float *workspace, void *stream);
/* _dequeue_and_enqueue (pr_hub_model.py:112-122): queue[c,l,ptr+b] = keys[b,l,c]; queue float32 [C,L,K]. */
int evp_enqueue_keys(float *queue, const float *keys, int ptr, int B, int L, int C, int K, void *stream);
/* The same with the pointer in device memory (the reference's `queue_ptr` buffer, int64 [1]): read by the kernel and
 * advanced to (ptr + B) % K afterwards on the stream -- no host read-back, capturable in a HIP graph. K % B == 0. */
int evp_enqueue_keys_dev(float *queue, const float *keys, int64_t *queue_ptr, int B, int L, int C, int K, void *stream);

/* ------------------------------------------------------------------------------------------------ K17 Swin windows
 * Grouped window attention with the gathered relative-position bias, replacing WindowAttention.forward's core
 * (model/sub_module/swin_block.py:135-158) for every head width d_h = 32 (all four Swin-T stages):
 *   S = (scale*q) k^T + bias,  bias[g,i,j] = rel[g,i,j] >= 0 ? table[rel[g,i,j], h] : -100,  P = softmax(S),  out = P v.
 * qkv dtype [Bg, N, 3, H, 32] (the packed qkv Linear output), Bg = batch * nG with group index g = bg % nG;
 * table float32 [R, H] (relative_position_bias_table); rel int32 [nG, N, N] with -1 marking the pairs the
 * reference masks (different window, or padding: it zeroes their bias and adds -100, :140-149);
 * out dtype [Bg, N, H*32]; probs (may be NULL) float32 [Bg, H, N, N]. N <= 128, R <= 512. */
int evp_window_attention_fwd(const void *qkv, const float *table, const int32_t *rel, void *out, float *probs, int Bg,
                             int nG, int N, int H, int R, float scale, int dtype, const void *keep, float keep_scale, void *stream);
/* `keep` (both calls; NULL = none): dropout on the probabilities (reference swin_block.py:113,152), uint8 [Bg, H, N, N] keep flags,
 * P <- P * keep * keep_scale with keep_scale = 1 / (1 - p) before the product with V; `probs` then holds the dropped map, as the
 * reference returns it (:157).
 * Backward of the above: recomputes P, writes dqkv (same layout as qkv) and dtable float32 [R, H] (zeroed here,
 * accumulated with LDS-privatised atomics; masked pairs contribute nothing, as in the reference). `out` is the
 * forward output (used for rowsum(P*dP) = dO.O). */
int evp_window_attention_bwd(const void *qkv, const float *table, const int32_t *rel, const void *out,
                             const void *dout, void *dqkv, float *dtable, int Bg, int nG, int N, int H, int R,
                             float scale, int dtype, const void *keep, float keep_scale, void *stream);
/* The same windowed attention on the MFMA kernels of evp_attention_fused_* (bf16, d_h = 32, N <= 128), in three calls per block:
 *   evp_window_bias_build   addm[g,h,i,j] = rel[g,i,j] >= 0 ? table[rel[g,i,j],h] : -100 and its transpose addmT[g,h,j,i], float32
 *                           [nG, H, NP, NP] each with NP = evp_window_attention_fused_np(N) (32 / 64 / 96 / 128), zero outside N x N;
 *   evp_window_attention_fused_fwd   out = softmax(scale*q k^T + addm[bg % nG, h]) v; qkv / out bf16 as above; lse float32
 *                           [Bg*H, N] (log-sum-exp of the logits, kept for the backward; probabilities are never stored);
 *   evp_window_attention_fused_bwd   dqkv (bf16) and dA float32 [nchunk, nG, H, NP, NP], nchunk = evp_window_attention_fused_nchunk(Bg, nG,
 *                           H): every batch chunk's workgroups store the d logits they summed over their batch items into their own
 *                           planes (plain stores, the N x N part of every plane is written); evp_window_bias_reduce then sums the
 *                           chunks and folds them through rel into dtable float32 [R, H] (cleared there; masked pairs contribute
 *                           nothing, swin_block.py:140-149). */
int evp_window_attention_fused_np(int N);
int evp_window_attention_fused_nchunk(int Bg, int nG, int heads);
int evp_window_bias_build(const float *table, const int32_t *rel, int nG, int N, int heads, int R, float *addm, float *addmT,
                          void *stream);
int evp_window_attention_fused_fwd(const void *qkv, const float *addm, int Bg, int nG, int N, int heads, float scale, void *out,
                                   float *lse, void *stream);
int evp_window_attention_fused_bwd(const void *qkv, const void *out, const void *dout, const float *lse, const float *addm,
                                   const float *addmT, int Bg, int nG, int N, int heads, float scale, void *dqkv, float *dA,
                                   void *stream);
int evp_window_bias_reduce(const float *dA, const int32_t *rel, int Bg, int nG, int N, int heads, int R, float *dtable, void *stream);
/* Token row gather used by GroupingModule.group/merge (swin_block.py:454-466) and PatchMerging's 2x2 regrouping
 * (:193-201): out[b, s, :] = idx[s] >= 0 ? x[b, idx[s], :] : 0; x float32 [B, n_in, C], out float32 [B, n_out, C],
 * idx int32 [n_out] (shared by the batch) or [B, n_out] when idx_per_sample != 0. C % 4 == 0. The backward of a
 * gather whose real (non-padding) indices form a permutation is the same call with the inverse index. */
int evp_gather_rows_f32(const float *x, const int32_t *idx, float *out, int B, int n_in, int n_out, int C,
                        int idx_per_sample, void *stream);
/* Stage-fusion patch rows (model/backbone/swin.py:201-208,213-218,223-228: zero grid, scatter visible tokens,
 * Conv2d(k, stride k), gather by ids_keep): A float32 [B*K, C*k*k] holds, for each kept decoder cell
 * ids_keep[b][j] (int64 [B,K]) of the (R/k)x(R/k) grid, the k*k tokens under it in Conv2d weight order (c, ky, kx);
 * hidden positions give zeros. x float32 [B, n, C] visible tokens, tokmap int32 [R*R] dense position -> token
 * index or -1. The conv itself is then one GEMM against weight.view(out, C*k*k). */
int evp_swin_fuse_gather_f32(const float *x, const int32_t *tokmap, const int64_t *ids_keep, float *A, int B, int n,
                             int K, int C, int R, int k, void *stream);
/* Its backward: dx float32 [B, n, C] from dA; coords int32 [n, 2] (row, col of each visible token),
 * ids_restore int64 [B, (R/k)^2] (a cell is kept iff ids_restore < K, and then sits at row ids_restore). */
int evp_swin_fuse_gather_bwd_f32(const float *dA, const int32_t *coords, const int64_t *ids_restore, float *dx, int B,
                                 int n, int K, int C, int R, int k, void *stream);
/* HOST function (no GPU work): group_windows of swin_block.py:322-347 with knapsack :277-319. counts int32
 * [n_windows] visible tokens per window (1..cap); outputs group_of_window int32 [n_windows], group_sizes int32
 * [<= n_windows] tokens per group, *n_groups. Inside a group the windows keep increasing index order. */
int evp_swin_group_windows(const int32_t *counts, int n_windows, int cap, int32_t *group_of_window,
                           int32_t *group_sizes, int32_t *n_groups);

/* ------------------------------------------------------------------------------------------------ K22 view augmentation
 * evg_augment of the loader (dataset/augmentation/view_augment.py:84-95) on voxel grids already in HBM: per sample a
 * crop box, F.interpolate(mode="nearest") to Hout x Wout (source index min(floorf(dst * float(in)/float(out)), in-1)),
 * horizontal flip of the resized view, time flip = reversed bin order and, when negate_on_time_flip (5/6-bin polarity
 * grids, :51-52), negation. in float32 [B,C,Hin,Win]; params int32 [B,6] = {x0, y0, w, h, hflip, tflip} (the random
 * decisions are the caller's: a box with 0 <= x0, x0 + w <= Win, 0 <= y0, y0 + h <= Hin); out float32 [B,C,Hout,Wout]. */
int evp_view_augment_f32(const float *in, const int32_t *params, float *out, int B, int C, int Hin, int Win, int Hout,
                         int Wout, int negate_on_time_flip, void *stream);
/* Difference-map target of the same sample (reference dataset/augmentation/view_augment.py:79-89 `frame_augment`: view_crop
 * -> view_resize(mode='bicubic') -> view_horizontal_flip -> negate when evg_augment time-flipped): same params rows as
 * evp_view_augment_f32 (the reference re-seeds numpy with the same seed, so crop box and flip coin are the voxel grid's;
 * params[5] carries evg_augment's time-flip flag). Bicubic = ATen upsample_bicubic2d, align_corners = False, A = -0.75. */
int evp_frame_augment_f32(const float *in, const int32_t *params, float *out, int B, int C, int Hin, int Win, int Hout, int Wout,
                          void *stream);

/* ------------------------------------------------------------------------------------------------ K23 token mean pool
 * Classification fine-tuning head (model/finetune_cls/ft_cls_hub_model.py:136): out[b,:] = mean_n x[b,n,:] for float32
 * tokens [B,N,D] (D % 4 == 0), and dx[b,n,:] = g[b,:] / N. The head's Linear and cross-entropy are evp_gemm and
 * evp_cross_entropy. */
int evp_token_mean_fwd(const float *x, int B, int N, int D, float *out, void *stream);
int evp_token_mean_bwd(const float *g, int B, int N, int D, float *dx, void *stream);

/* ------------------------------------------------------------------------------------------------ K24 stochastic depth / dropout
 * The regularisers of the fine-tuning recipe (main_finetune_cls.py:151-153, drop_path_rate 0.1 by default).
 * DropPath (timm 0.3.2 `drop_path`; model/sub_module/vit_block.py:241,252-253, conv_block.py:35,43-49, swin_block.py:257,270-271):
 *   evp_rows_scale_f32: out[m,:] = (res ? res[m,:] : 0) + s_b * x[m,:] with b = m / rows_per_sample and
 *   s_b = floor(keep_prob + u[b]) / keep_prob (u: float32 [M / rows_per_sample] uniform draws of the caller; u == NULL or
 *   keep_prob >= 1: s_b = 1); out_lp (optional) receives s_b * x[m,:] in bf16. Forward: x + drop_path(branch); backward: the
 *   branch's incoming gradient s_b * g. float32, D % 4 == 0; out or out_lp may be NULL.
 * Dropout (nn.Dropout; vit_block.py:137-141,226-231, vit.py:114): evp_dropout_fwd draws keep ~ Bernoulli(1 - p) per element
 *   from Philox4x32-10 keyed by (seed + *seed_dev, offset + element / 4) -- seed_dev (optional) is a uint64 scalar in DEVICE memory,
 *   drawn per step on the device so that a replayed HIP graph, whose kernel arguments are frozen, draws fresh masks -- writes out = x * keep / (1 - p) and mask (uint8, 1 = kept);
 *   evp_dropout_apply: out = x * mask * scale (the backward, and the forward for a given mask). dtype = EVP_F32 | EVP_BF16. */
int evp_rows_scale_f32(const float *x, const float *u, float keep_prob, const float *res, int64_t M, int D, int rows_per_sample,
                       float *out, void *out_lp, void *stream);
int evp_dropout_fwd(const void *x, int dtype, void *out, void *mask, int64_t n, float p, uint64_t seed, const void *seed_dev, uint64_t offset, void *stream);
int evp_dropout_apply(const void *x, int dtype, const void *mask, void *out, int64_t n, float scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EVTPRETRAIN_H */
