/* CPU restatement (plain C) of the reference's event -> voxel-grid histogram.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker / reported CPU baseline. The product (eventpretrain_amd/) never links or calls this file.
 *
 * Restates dataset/dataset_utils/events_to_voxel_grid.py:4-61 of the reference:
 *   :19-25  t0 = first row's stamp, t1 = last row's stamp (rows are assumed time-sorted), dT = t1 - t0, 0 -> 1.0
 *   :32-34  xs, ys = trunc-toward-zero to int64; ts = (bins-1)*(t - t0)/dT in float64
 *   :35-36  ps = float32(p); 0 -> -1
 *   :38-42  tis = floor(ts); dts = ts - tis; left = ps*(1 - float32(dts)); right = ps*float32(dts)  (float32 math)
 *   :44-49  left add where 0 <= tis < bins at x + y*W + tis*W*H
 *   :51-57  right add where 0 <= tis and tis+1 < bins at x + y*W + (tis+1)*W*H
 * All left adds are applied first (event order), then all right adds, as two index_add_ calls do on one thread;
 * float32 accumulation, so the result is bit-identical to the reference run with torch.set_num_threads(1)
 * (main_pretrain.py:12) -- pinned by tests/golden/voxel.npz.
 *
 * A flat index outside [0, bins*H*W) makes the reference raise IndexError; here it is skipped and counted.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* events: n rows of 4 doubles. Column order (x,y,t,p), or (t,x,y,p) when is_txyp != 0.
 * out: bins*H*W floats, overwritten. Returns the number of skipped out-of-range adds. */
int64_t evp_oracle_voxel_f32(const double *events, int64_t n, int bins, int H, int W, int is_txyp, float *out) {
  const int64_t plane = (int64_t)H * W, total = plane * bins;
  memset(out, 0, sizeof(float) * (size_t)total);
  if (n <= 0) return 0;
  const int cx = is_txyp ? 1 : 0, cy = is_txyp ? 2 : 1, ct = is_txyp ? 0 : 2, cp = 3;
  const double t0 = events[ct], t1 = events[(n - 1) * 4 + ct];
  double dT = t1 - t0;
  if (dT == 0) dT = 1.0;
  int64_t skipped = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t i = 0; i < n; ++i) {
      const double *e = events + i * 4;
      const int64_t x = (int64_t)e[cx], y = (int64_t)e[cy];
      const double ts = (double)(bins - 1) * (e[ct] - t0) / dT;
      float p = (float)e[cp];
      if (p == 0.0f) p = -1.0f;
      const double tf = floor(ts);
      const float dt = (float)(ts - tf);
      if (!(tf >= 0)) continue;
      float val;
      double bin;
      if (pass == 0) {
        if (!(tf < bins)) continue;
        val = p * (1.0f - dt);
        bin = tf;
      } else {
        if (!(tf + 1 < bins)) continue;
        val = p * dt;
        bin = tf + 1;
      }
      const int64_t idx = x + y * W + (int64_t)bin * plane;
      if (idx < 0 || idx >= total) { ++skipped; continue; }
      out[idx] += val;
    }
  }
  return skipped;
}

/* Batch form used as the CPU baseline: clip c owns rows [offsets[c], offsets[c+1]) and grid c of `out`. */
int64_t evp_oracle_voxel_batch_f32(const double *events, const int64_t *offsets, int n_clips, int bins, int H, int W,
                                   int is_txyp, float *out) {
  int64_t skipped = 0;
  for (int c = 0; c < n_clips; ++c)
    skipped += evp_oracle_voxel_f32(events + offsets[c] * 4, offsets[c + 1] - offsets[c], bins, H, W, is_txyp,
                                    out + (int64_t)c * bins * H * W);
  return skipped;
}

#ifdef __cplusplus
}
#endif
