// ConvViT stage-1/2 kernels on channels-last token maps ([B,H,W,C], C contiguous), gfx950. All HBM-bound.
//   * patch gather / scatter for the non-overlapping strided convolutions (PatchEmbed k=s=2/4 and the multi-scale
//     fusion convs k=s=4/2): the conv itself is a GEMM on the gathered patch matrix (csrc/gemm.hip).
//   * depthwise 5x5 convolution (pad 2, groups=C) with the keep-mask multiply of ConvBlock fused into the input read.
// Replaces model/sub_module/conv_block.py:41-51 (ConvBlock.attn + mask) and the Conv2d(k=s) layers of
// model/backbone/convvit.py:20-25,48-49.
#include "evp_common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }

// cols[(b,j), c*p*p + py*p + px] = x[b, gy*p+py, gx*p+px, c]; token = ids_keep[b,j] (or j) -> (gy, gx)
// one thread per (row, c): p*p strided loads (coalesced over c), p*p contiguous stores
template <typename TO>
__global__ __launch_bounds__(256) void patchify_nhwc_kernel(const float *x, const int64_t *ids_keep, int H, int W, int C, int p,
                                                            int n_keep, int64_t rows, TO *cols) {
  const int gw = W / p, L = (H / p) * gw, pp = p * p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int64_t b = row / n_keep;
    const int tok = ids_keep ? (int)ids_keep[row] : (int)(row % L);
    const int gy = tok / gw, gx = tok % gw;
    TO *o = cols + row * ((int64_t)C * pp) + (int64_t)c * pp;
    for (int py = 0; py < p; ++py)
      for (int px = 0; px < p; ++px) {
        const float v = x[(((int64_t)b * H + gy * p + py) * W + gx * p + px) * C + c];
        ElemIO<TO>::st(o + py * p + px, v);
      }
  }
}
// dx[b, gy*p+py, gx*p+px, c] (+)= dcols[(b,j), c*p*p + py*p + px]; dx must be zero-initialised when ids_keep != NULL
template <typename TI>
__global__ __launch_bounds__(256) void unpatchify_nhwc_kernel(const TI *dcols, const int64_t *ids_keep, int H, int W, int C, int p,
                                                              int n_keep, int64_t rows, int accumulate, float *dx) {
  const int gw = W / p, L = (H / p) * gw, pp = p * p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int64_t b = row / n_keep;
    const int tok = ids_keep ? (int)ids_keep[row] : (int)(row % L);
    const int gy = tok / gw, gx = tok % gw;
    const TI *s = dcols + row * ((int64_t)C * pp) + (int64_t)c * pp;
    for (int py = 0; py < p; ++py)
      for (int px = 0; px < p; ++px) {
        float *d = dx + (((int64_t)b * H + gy * p + py) * W + gx * p + px) * C + c;
        const float v = ldf<TI>(s + py * p + px);
        *d = accumulate ? *d + v : v;
      }
  }
}

// keep factor of position (y,x): 1 - mask[b, (y/s)*gw + x/s]  (convvit.py:129-130,142-143), or 1 without a mask
__device__ __forceinline__ float keep_at(const float *mask, int b, int y, int x, int s, int gw, int L) {
  return mask ? 1.0f - mask[(int64_t)b * L + (y / s) * gw + x / s] : 1.0f;
}

// Forward:  y[b,y,x,c]   = bias[c] + sum_{i,j} w[c,i,j] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c]
// Backward: din[b,y,x,c] = keep(y,x) * sum_{i,j} w[c,i,j] * dout[b,y-i+2,x-j+2,c]     (the flipped kernel on dout)
// One kernel for both: a wave walks an image row left to right with a 5x5 register window per channel (lane = 2
// adjacent channels, 64 lanes = 128 channels = one contiguous 256-byte / 512-byte row segment), so a position costs
// 5 loads + 1 store instead of 25 + 1. BWD: window elements are not masked, the result is; the taps are read flipped.
template <typename T> __device__ __forceinline__ void ld2(const T *p, float &a, float &b);
template <> __device__ __forceinline__ void ld2<float>(const float *p, float &a, float &b) {
  const float2 v = *reinterpret_cast<const float2 *>(p);
  a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t *p, float &a, float &b) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void st2(T *p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t *p, float a, float b) {
  *reinterpret_cast<uint32_t *>(p) = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
constexpr int DWF_RPW = 2;                    // rows per wave
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const T *src, const float *mask, const float *w, const float *bias, int B, int H,
                                                          int W, int C, int ms, int mgw, int mL, T *dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  if (c >= C) return;
  const int nrows = B * H;
  float k0[5][5], k1[5][5];                   // taps of the two channels (flipped for the backward)
#pragma unroll
  for (int ki = 0; ki < 5; ++ki)
#pragma unroll
    for (int kj = 0; kj < 5; ++kj) {
      const int t = BWD ? (4 - ki) * 5 + (4 - kj) : ki * 5 + kj;
      k0[ki][kj] = w[c * 25 + t];
      k1[ki][kj] = w[(c + 1) * 25 + t];
    }
  const float b0 = BWD ? 0.f : bias[c], b1 = BWD ? 0.f : bias[c + 1];
  const int r0 = (blockIdx.y * 4 + wave) * DWF_RPW;
  for (int r = r0; r < r0 + DWF_RPW && r < nrows; ++r) {
    const int b = r / H, y = r - b * H;
    float w0[5][5], w1[5][5];
    auto load_col = [&](int xx, int kj) {
#pragma unroll
      for (int ki = 0; ki < 5; ++ki) {
        const int yy = y + ki - 2;
        float a_ = 0.f, b_ = 0.f;
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
          const float k = BWD ? 1.0f : keep_at(mask, b, yy, xx, ms, mgw, mL);
          if (k != 0.f) {
            ld2<T>(src + (((int64_t)b * H + yy) * W + xx) * C + c, a_, b_);
            a_ *= k; b_ *= k;
          }
        }
        w0[ki][kj] = a_; w1[ki][kj] = b_;
      }
    };
    load_col(-2, 0); load_col(-1, 1); load_col(0, 2); load_col(1, 3);
    for (int x = 0; x < W; ++x) {
      load_col(x + 2, 4);
      float s0 = b0, s1 = b1;
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
          s0 += k0[ki][kj] * w0[ki][kj];
          s1 += k1[ki][kj] * w1[ki][kj];
        }
      if (BWD) {
        const float k = keep_at(mask, b, y, x, ms, mgw, mL);
        s0 *= k; s1 *= k;
      }
      st2<T>(dst + ((int64_t)r * W + x) * C + c, s0, s1);
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
    }
  }
}

// dw[c,i,j] = sum_{b,y,x} dout[b,y,x,c] * keep(y+i-2,x+j-2) * in[b,y+i-2,x+j-2,c];  dbias[c] = sum dout
// A wave walks image rows (b, y) left to right with a 5x5 register window of the (masked) input per channel: one new
// column (5 loads) + one dout load per position instead of 25 + 1, and the keep mask is looked up once per loaded
// element. Lane = 2 adjacent channels (one 4-byte / 8-byte load: 64 lanes cover 128 channels = 256 contiguous bytes).
// block = 4 waves x DW_RPW rows each; partials part[slab][26][C] (25 taps + bias), slab = blockIdx.y.
constexpr int DW_RPW = 4;                     // rows per wave
constexpr int DW_ROWS = 4 * DW_RPW;           // rows per block (= slab)
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T *dout, const T *in, const float *mask, int B, int H, int W, int C,
                                                                int ms, int mgw, int mL, float *part) {
  __shared__ float sh[4][26][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  const int nrows = B * H;
  float acc0[26], acc1[26];
#pragma unroll
  for (int k = 0; k < 26; ++k) acc0[k] = acc1[k] = 0.f;
  if (c < C) {
    const int r0 = blockIdx.y * DW_ROWS + wave * DW_RPW;
    for (int r = r0; r < r0 + DW_RPW && r < nrows; ++r) {
      const int b = r / H, y = r - b * H;
      float w0[5][5], w1[5][5];          // window: w[ki][kj] = keep * in[b, y+ki-2, x+kj-2, c / c+1]
      auto load_col = [&](int xx, float (&o0)[5], float (&o1)[5]) {
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) {
          const int yy = y + ki - 2;
          o0[ki] = o1[ki] = 0.f;
          if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const float k = keep_at(mask, b, yy, xx, ms, mgw, mL);
            if (k != 0.f) {
              float a_, b_;
              ld2<T>(in + (((int64_t)b * H + yy) * W + xx) * C + c, a_, b_);
              o0[ki] = k * a_; o1[ki] = k * b_;
            }
          }
        }
      };
      float c0[5], c1[5];
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {     // columns x-2 .. x+1 for x = 0
        load_col(kj - 2, c0, c1);
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) { w0[ki][kj] = c0[ki]; w1[ki][kj] = c1[ki]; }
      }
      for (int x = 0; x < W; ++x) {
        load_col(x + 2, c0, c1);
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) { w0[ki][4] = c0[ki]; w1[ki][4] = c1[ki]; }
        float g0, g1;
        ld2<T>(dout + ((int64_t)r * W + x) * C + c, g0, g1);
        acc0[25] += g0; acc1[25] += g1;
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 5; ++kj) {
            acc0[ki * 5 + kj] += g0 * w0[ki][kj];
            acc1[ki * 5 + kj] += g1 * w1[ki][kj];
          }
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 26; ++k) { sh[wave][k][lane * 2] = acc0[k]; sh[wave][k][lane * 2 + 1] = acc1[k]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 26 * 128; e += 256) {
    const int k = e / 128, cc = e % 128;
    if (blockIdx.x * 128 + cc < C)
      part[((int64_t)blockIdx.y * 26 + k) * C + blockIdx.x * 128 + cc] = sh[0][k][cc] + sh[1][k][cc] + sh[2][k][cc] + sh[3][k][cc];
  }
}
// ---- band form (round 2) ---------------------------------------------------------------------------------------------------
// The row walkers above issue the 5 loads of a column when they need them: one exposed load latency per position (measured on
// ConvViT-Base stage 1, B=64, 56x56x256 bf16: forward 294 us, data gradient 250 us, weight gradient 470 us against ~60 us of HBM
// time). Here a workgroup first stages a BAND of the (masked) input -- R output rows + 4 halo rows, W + 4 columns, CB channels =
// 128 bytes per position -- into LDS with all its 16-byte loads in flight at once (zero halo, keep mask applied while staging),
// then every thread walks its row segment with the same 5x5 register window, fed by 5 LDS reads per position.
// Thread = (channel pair, segment): NPAIR = 64 / sizeof(T) pairs x NSEG = 256 / NPAIR segments = R rows x SPR segments per row.
template <typename T> struct BandCfg {
  static constexpr int EPC = 16 / sizeof(T);              // elements per 16-byte chunk
  static constexpr int CB = 128 / sizeof(T);              // channels per workgroup (128 bytes per position)
  static constexpr int NPAIR = CB / 2, NSEG = 256 / NPAIR;
};
template <typename T> __device__ __forceinline__ void chunk_scale(uint4 &v, float k);
template <> __device__ __forceinline__ void chunk_scale<float>(uint4 &v, float k) {
  v.x = __float_as_uint(__uint_as_float(v.x) * k); v.y = __float_as_uint(__uint_as_float(v.y) * k);
  v.z = __float_as_uint(__uint_as_float(v.z) * k); v.w = __float_as_uint(__uint_as_float(v.w) * k);
}
template <> __device__ __forceinline__ void chunk_scale<bf16_t>(uint4 &v, float k) {
  uint32_t *u = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = __uint_as_float(u[i] << 16) * k, b = __uint_as_float(u[i] & 0xFFFF0000u) * k;
    u[i] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
  }
}
// stage rows [y0 - 2, y0 + R + 2) x columns [-2, W + 2) of image b, channels [c0, c0 + CB): band[row][col + 2][128 bytes].
// MASKED: the keep factor of every band position is worked out ONCE per position into `keep_s` (one mask load and two integer
// divisions per position instead of per 16-byte chunk; 0 for the halo), and positions with keep 0 -- half of them at mask ratio
// 0.5 -- are not loaded at all.
template <typename T, int R, bool MASKED>
__device__ __forceinline__ void stage_band(const T *__restrict__ src, const float *__restrict__ mask, int b, int y0, int H, int W, int C, int c0,
                                           int ms, int mgw, int mL, char *band, float *keep_s, int tid) {
  constexpr int EPC = BandCfg<T>::EPC;
  const int WP = W + 4, npos = (R + 4) * WP, total = npos * 8;
  if constexpr (MASKED) {
    for (int pos = tid; pos < npos; pos += 256) {
      const int col = pos % WP, row = pos / WP;
      const int yy = y0 - 2 + row, xx = col - 2;
      float k = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) k = mask ? keep_at(mask, b, yy, xx, ms, mgw, mL) : 1.f;
      keep_s[pos] = k;
    }
    __syncthreads();
  }
  for (int base = tid; base < total; base += 256 * 4) {
    uint4 v[4];
    float k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * 256;
      v[u] = make_uint4(0, 0, 0, 0);
      k[u] = 1.f;
      if (i < total) {
        const int ch = i & 7, col = (i >> 3) % WP, row = (i >> 3) / WP;
        const int yy = y0 - 2 + row, xx = col - 2;
        bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        if constexpr (MASKED) {
          k[u] = keep_s[i >> 3];
          in = k[u] != 0.f;
        }
        if (in) v[u] = *reinterpret_cast<const uint4 *>(src + (((int64_t)b * H + yy) * W + xx) * C + c0 + ch * EPC);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * 256;
      if (i < total) {
        if (MASKED && k[u] != 1.f && k[u] != 0.f) chunk_scale<T>(v[u], k[u]);
        *reinterpret_cast<uint4 *>(band + (int64_t)(i >> 3) * 128 + (i & 7) * 16) = v[u];
      }
    }
  }
}
// segment s of SPR covers columns [seg_x0(s), seg_x0(s + 1)); cut points are made ODD multiples apart where possible so that the two
// half-waves of a wave (adjacent segments of one row) read LDS positions an odd number of 128-byte steps apart (no bank conflict)
__device__ __forceinline__ int seg_x0(int s, int spr, int W) {
  if (s <= 0) return 0;
  if (s >= spr) return W;
  int x = (W * s) / spr;
  if (((x - (W * (s - 1)) / spr) & 1) == 0 && x + 1 < W) ++x;
  return x;
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dwconv_band_kernel(const T *__restrict__ src, const float *__restrict__ mask, const float *__restrict__ w,
                                                          const float *__restrict__ bias, int B, int H, int W, int C, int ms, int mgw, int mL,
                                                          T *__restrict__ dst) {
  constexpr int R = 4, NPAIR = BandCfg<T>::NPAIR, NSEG = BandCfg<T>::NSEG, SPR = NSEG / R, CB = BandCfg<T>::CB;
  extern __shared__ __attribute__((aligned(16))) char band[];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * CB;
  const int r0 = blockIdx.y * R, b = r0 / H, y0 = r0 - b * H;          // H % R == 0: a band never crosses images
  stage_band<T, R, !BWD>(src, mask, b, y0, H, W, C, c0, ms, mgw, mL, band, reinterpret_cast<float *>(band + (int64_t)(R + 4) * (W + 4) * 128), tid);
  const int pair = tid % NPAIR, seg = tid / NPAIR, row = seg / SPR, sx = seg % SPR;
  const int c = c0 + pair * 2;
  float k0[5][5], k1[5][5];
#pragma unroll
  for (int ki = 0; ki < 5; ++ki)
#pragma unroll
    for (int kj = 0; kj < 5; ++kj) {
      const int t = BWD ? (4 - ki) * 5 + (4 - kj) : ki * 5 + kj;
      k0[ki][kj] = w[c * 25 + t];
      k1[ki][kj] = w[(c + 1) * 25 + t];
    }
  const float b0 = BWD ? 0.f : bias[c], b1 = BWD ? 0.f : bias[c + 1];
  __syncthreads();
  const int WP = W + 4, x_lo = seg_x0(sx, SPR, W), x_hi = seg_x0(sx + 1, SPR, W);
  // window column kj of output x = band column x + kj (band column = image column + 2)
  const char *brow = band + ((int64_t)row * WP) * 128 + pair * 2 * sizeof(T);
  float w0[5][5], w1[5][5];
  auto load_col = [&](int bc, int kj) {
#pragma unroll
    for (int ki = 0; ki < 5; ++ki) ld2<T>(reinterpret_cast<const T *>(brow + ((int64_t)ki * WP + bc) * 128), w0[ki][kj], w1[ki][kj]);
  };
  if (x_lo < x_hi) {
    load_col(x_lo, 0); load_col(x_lo + 1, 1); load_col(x_lo + 2, 2); load_col(x_lo + 3, 3);
    const int y = y0 + row;
    for (int x = x_lo; x < x_hi; ++x) {
      load_col(x + 4, 4);
      float s0 = b0, s1 = b1;
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
          s0 += k0[ki][kj] * w0[ki][kj];
          s1 += k1[ki][kj] * w1[ki][kj];
        }
      if (BWD) {
        const float k = keep_at(mask, b, y, x, ms, mgw, mL);
        s0 *= k; s1 *= k;
      }
      st2<T>(dst + (((int64_t)b * H + y) * W + x) * C + c, s0, s1);
#pragma unroll
      for (int ki = 0; ki < 5; ++ki)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
    }
  }
}

// Weight gradient, band form: a workgroup owns DW_ROWS image rows (one slab, as above) and walks them in bands of R = 2 rows: the
// masked input band (R + 4 rows) and the R rows of dout are staged in LDS, every thread accumulates its 25 + 1 sums for its channel
// pair over its segment, the segments meet in LDS once per slab.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_band_kernel(const T *__restrict__ dout, const T *__restrict__ in, const float *__restrict__ mask,
                                                                     int B, int H, int W, int C, int ms, int mgw, int mL, float *__restrict__ part) {
  constexpr int R = 2, NPAIR = BandCfg<T>::NPAIR, NSEG = BandCfg<T>::NSEG, SPR = NSEG / R, CB = BandCfg<T>::CB, EPC = BandCfg<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, WP = W + 4;
  char *band = smem, *gband = smem + (int64_t)(R + 4) * WP * 128;       // gband[R][W][128 bytes]
  const int c0 = blockIdx.x * CB, nrows = B * H;
  const int pair = tid % NPAIR, seg = tid / NPAIR, row = seg / SPR, sx = seg % SPR;
  const int x_lo = seg_x0(sx, SPR, W), x_hi = seg_x0(sx + 1, SPR, W);
  float acc0[26], acc1[26];
#pragma unroll
  for (int k = 0; k < 26; ++k) acc0[k] = acc1[k] = 0.f;
  for (int r0 = blockIdx.y * DW_ROWS; r0 < (blockIdx.y + 1) * DW_ROWS && r0 < nrows; r0 += R) {
    const int b = r0 / H, y0 = r0 - b * H;                               // H % R == 0
    __syncthreads();                                                     // the previous band has been consumed
    stage_band<T, R, true>(in, mask, b, y0, H, W, C, c0, ms, mgw, mL, band, reinterpret_cast<float *>(gband + (int64_t)R * W * 128), tid);
    for (int i = tid; i < R * W * 8; i += 256) {
      const int ch = i & 7, pos = i >> 3;
      *reinterpret_cast<uint4 *>(gband + (int64_t)pos * 128 + ch * 16) =
          *reinterpret_cast<const uint4 *>(dout + ((int64_t)r0 * W + pos) * C + c0 + ch * EPC);
    }
    __syncthreads();
    if (x_lo < x_hi) {
      const char *brow = band + ((int64_t)row * WP) * 128 + pair * 2 * sizeof(T);
      const char *grow = gband + ((int64_t)row * W) * 128 + pair * 2 * sizeof(T);
      float w0[5][5], w1[5][5];
      auto load_col = [&](int bc, int kj) {
#pragma unroll
        for (int ki = 0; ki < 5; ++ki) ld2<T>(reinterpret_cast<const T *>(brow + ((int64_t)ki * WP + bc) * 128), w0[ki][kj], w1[ki][kj]);
      };
      load_col(x_lo, 0); load_col(x_lo + 1, 1); load_col(x_lo + 2, 2); load_col(x_lo + 3, 3);
      for (int x = x_lo; x < x_hi; ++x) {
        load_col(x + 4, 4);
        float g0, g1;
        ld2<T>(reinterpret_cast<const T *>(grow + (int64_t)x * 128), g0, g1);
        acc0[25] += g0; acc1[25] += g1;
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 5; ++kj) {
            acc0[ki * 5 + kj] += g0 * w0[ki][kj];
            acc1[ki * 5 + kj] += g1 * w1[ki][kj];
          }
#pragma unroll
        for (int ki = 0; ki < 5; ++ki)
#pragma unroll
          for (int kj = 0; kj < 4; ++kj) { w0[ki][kj] = w0[ki][kj + 1]; w1[ki][kj] = w1[ki][kj + 1]; }
      }
    }
  }
  // the NSEG segments of a channel pair meet in LDS (reusing the band): red[seg][26][CB]
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int k = 0; k < 26; ++k) {
    red[((int64_t)seg * 26 + k) * CB + pair * 2] = acc0[k];
    red[((int64_t)seg * 26 + k) * CB + pair * 2 + 1] = acc1[k];
  }
  __syncthreads();
  for (int e = tid; e < 26 * CB; e += 256) {
    const int k = e / CB, cc = e % CB;
    float sum = 0.f;
    for (int sg = 0; sg < NSEG; ++sg) sum += red[((int64_t)sg * 26 + k) * CB + cc];
    part[((int64_t)blockIdx.y * 26 + k) * C + c0 + cc] = sum;
  }
}

template <typename T> static inline bool band_ok(int H, int W, int C) {
  return H % 4 == 0 && DW_ROWS % 2 == 0 && W >= 8 && W <= 64 && C % BandCfg<T>::CB == 0;
}
template <typename T> static inline size_t band_smem_fwd(int W) { return (size_t)8 * (W + 4) * 128 + (size_t)8 * (W + 4) * sizeof(float); }
template <typename T> static inline size_t band_smem_bwdw(int W) {
  const size_t a = (size_t)6 * (W + 4) * 128 + (size_t)2 * W * 128 + (size_t)6 * (W + 4) * sizeof(float),
               r = (size_t)BandCfg<T>::NSEG * 26 * BandCfg<T>::CB * sizeof(float);
  return a > r ? a : r;
}

// dw[c*25 + k] = sum_slab part[slab][k][c] (k < 25), dbias[c] = sum_slab part[slab][25][c]
__global__ __launch_bounds__(256) void dwconv_bwd_weight_finalize(const float *part, int nslab, int C, float *dw, float *dbias) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 26 * C) return;
  const int k = e / C, c = e % C;
  float s = 0.f;
  for (int i = 0; i < nslab; ++i) s += part[((int64_t)i * 26 + k) * C + c];
  if (k < 25) dw[c * 25 + k] = s;
  else dbias[c] = s;
}

static int g_dwconv_band = 1;                 // 0: the row walkers (A/B, evp_dwconv_set_band)

static inline int ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int evp_patchify_nhwc(const float *x, const int64_t *ids_keep, int B, int H, int W, int C, int patch, int n_keep, void *cols,
                                 int dtype, void *stream) {
  EVP_CHECK_ARG(x && cols, EVP_EINVAL, "evp_patchify_nhwc: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_patchify_nhwc: bad shape");
  const int L = (H / patch) * (W / patch);
  EVP_CHECK_ARG(n_keep > 0 && n_keep <= L && (ids_keep || n_keep == L), EVP_ESHAPE, "evp_patchify_nhwc: bad n_keep");
  const int64_t rows = (int64_t)B * n_keep;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EVP_F32) hipLaunchKernelGGL(patchify_nhwc_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, s, x, ids_keep, H, W, C, patch, n_keep, rows, (float *)cols);
  else hipLaunchKernelGGL(patchify_nhwc_kernel<bf16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, s, x, ids_keep, H, W, C, patch, n_keep, rows, (bf16_t *)cols);
  EVP_CHECK_LAUNCH("evp_patchify_nhwc");
  return EVP_OK;
}

extern "C" int evp_unpatchify_nhwc(const void *dcols, int dtype, const int64_t *ids_keep, int B, int H, int W, int C, int patch, int n_keep,
                                   int accumulate, float *dx, void *stream) {
  EVP_CHECK_ARG(dcols && dx, EVP_EINVAL, "evp_unpatchify_nhwc: null pointer");
  EVP_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, EVP_ESHAPE, "evp_unpatchify_nhwc: bad shape");
  const int L = (H / patch) * (W / patch);
  EVP_CHECK_ARG(n_keep > 0 && n_keep <= L && (ids_keep || n_keep == L), EVP_ESHAPE, "evp_unpatchify_nhwc: bad n_keep");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && n_keep < L) {
    hipError_t e = evp_zero_async(dx, sizeof(float) * (size_t)B * H * W * C, s);
    EVP_CHECK_ARG(e == hipSuccess, EVP_ELAUNCH, "evp_unpatchify_nhwc: memset failed: %s", hipGetErrorString(e));
  }
  const int64_t rows = (int64_t)B * n_keep;
  if (dtype == EVP_F32) hipLaunchKernelGGL(unpatchify_nhwc_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, s, (const float *)dcols, ids_keep, H, W, C, patch, n_keep, rows, accumulate, dx);
  else hipLaunchKernelGGL(unpatchify_nhwc_kernel<bf16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, s, (const bf16_t *)dcols, ids_keep, H, W, C, patch, n_keep, rows, accumulate, dx);
  EVP_CHECK_LAUNCH("evp_unpatchify_nhwc");
  return EVP_OK;
}

#define DW_COMMON_CHECK(name)                                                                                              \
  EVP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVP_ESHAPE, name ": need C %% 4 == 0");                     \
  EVP_CHECK_ARG(!mask || (mask_scale > 0 && H % mask_scale == 0 && W % mask_scale == 0), EVP_ESHAPE, name ": bad mask scale"); \
  const int mgw = mask ? W / mask_scale : 1, mL = mask ? (H / mask_scale) * (W / mask_scale) : 1;

extern "C" int evp_dwconv5x5_fwd(const void *in, int dtype, const float *mask, int mask_scale, const float *w, const float *bias, int B,
                                 int H, int W, int C, void *out, void *stream) {
  EVP_CHECK_ARG(in && w && bias && out, EVP_EINVAL, "evp_dwconv5x5_fwd: null pointer");
  DW_COMMON_CHECK("evp_dwconv5x5_fwd")
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  (void)n;
  if (g_dwconv_band && (dtype == EVP_F32 ? band_ok<float>(H, W, C) : band_ok<bf16_t>(H, W, C))) {
    auto go = [&](auto kfn, auto tag) {
      using T = decltype(tag);
      const size_t sm = band_smem_fwd<T>(W);
      if (sm > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
      hipLaunchKernelGGL(kfn, dim3((unsigned)(C / BandCfg<T>::CB), (unsigned)((int64_t)B * H / 4)), dim3(256), sm, s, (const T *)in, mask, w, bias, B,
                         H, W, C, mask_scale, mgw, mL, (T *)out);
    };
    if (dtype == EVP_F32) go(dwconv_band_kernel<float, false>, float{});
    else go(dwconv_band_kernel<bf16_t, false>, bf16_t{});
    EVP_CHECK_LAUNCH("evp_dwconv5x5_fwd(band)");
    return EVP_OK;
  }
  const dim3 rg((unsigned)((C + 127) / 128), (unsigned)(((int64_t)B * H + 4 * DWF_RPW - 1) / (4 * DWF_RPW)));
  if (dtype == EVP_F32) hipLaunchKernelGGL((dwconv_rows_kernel<float, false>), rg, dim3(256), 0, s, (const float *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (float *)out);
  else hipLaunchKernelGGL((dwconv_rows_kernel<bf16_t, false>), rg, dim3(256), 0, s, (const bf16_t *)in, mask, w, bias, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)out);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_fwd");
  return EVP_OK;
}

extern "C" int evp_dwconv_set_band(int on) { g_dwconv_band = on ? 1 : 0; return EVP_OK; }

extern "C" int evp_dwconv5x5_bwd_nslab(int B, int H, int W) { (void)W; return (int)(((int64_t)B * H + DW_ROWS - 1) / DW_ROWS); }

extern "C" int evp_dwconv5x5_bwd(const void *dout, const void *in, int dtype, const float *mask, int mask_scale, const float *w, int B, int H,
                                 int W, int C, void *din, float *dw, float *dbias, float *workspace, void *stream) {
  EVP_CHECK_ARG(dout && in && w && din && dw && dbias && workspace, EVP_EINVAL, "evp_dwconv5x5_bwd: null pointer");
  DW_COMMON_CHECK("evp_dwconv5x5_bwd")
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  const int nslab = evp_dwconv5x5_bwd_nslab(B, H, W);
  dim3 wg((C + 127) / 128, nslab);
  (void)n;
  if (g_dwconv_band && (dtype == EVP_F32 ? band_ok<float>(H, W, C) : band_ok<bf16_t>(H, W, C))) {
    auto go = [&](auto kd, auto kw, auto tag) {
      using T = decltype(tag);
      const size_t sd = band_smem_fwd<T>(W), sw = band_smem_bwdw<T>(W);
      if (sd > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sd);
      if (sw > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sw);
      hipLaunchKernelGGL(kd, dim3((unsigned)(C / BandCfg<T>::CB), (unsigned)((int64_t)B * H / 4)), dim3(256), sd, s, (const T *)dout, mask, w,
                         (const float *)nullptr, B, H, W, C, mask_scale, mgw, mL, (T *)din);
      hipLaunchKernelGGL(kw, dim3((unsigned)(C / BandCfg<T>::CB), (unsigned)nslab), dim3(256), sw, s, (const T *)dout, (const T *)in, mask, B, H, W, C,
                         mask_scale, mgw, mL, workspace);
    };
    if (dtype == EVP_F32) go(dwconv_band_kernel<float, true>, dwconv_bwd_weight_band_kernel<float>, float{});
    else go(dwconv_band_kernel<bf16_t, true>, dwconv_bwd_weight_band_kernel<bf16_t>, bf16_t{});
    EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd(band)");
    hipLaunchKernelGGL(dwconv_bwd_weight_finalize, dim3((26 * C + 255) / 256), dim3(256), 0, s, workspace, nslab, C, dw, dbias);
    EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd(finalize)");
    return EVP_OK;
  }
  const dim3 rg((unsigned)((C + 127) / 128), (unsigned)(((int64_t)B * H + 4 * DWF_RPW - 1) / (4 * DWF_RPW)));
  if (dtype == EVP_F32) {
    hipLaunchKernelGGL((dwconv_rows_kernel<float, true>), rg, dim3(256), 0, s, (const float *)dout, mask, w, (const float *)nullptr, B, H, W, C, mask_scale, mgw, mL, (float *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<float>, wg, dim3(256), 0, s, (const float *)dout, (const float *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  } else {
    hipLaunchKernelGGL((dwconv_rows_kernel<bf16_t, true>), rg, dim3(256), 0, s, (const bf16_t *)dout, mask, w, (const float *)nullptr, B, H, W, C, mask_scale, mgw, mL, (bf16_t *)din);
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel<bf16_t>, wg, dim3(256), 0, s, (const bf16_t *)dout, (const bf16_t *)in, mask, B, H, W, C, mask_scale, mgw, mL, workspace);
  }
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd");
  hipLaunchKernelGGL(dwconv_bwd_weight_finalize, dim3((26 * C + 255) / 256), dim3(256), 0, s, workspace, nslab, C, dw, dbias);
  EVP_CHECK_LAUNCH("evp_dwconv5x5_bwd(finalize)");
  return EVP_OK;
}
