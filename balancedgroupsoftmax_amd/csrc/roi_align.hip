// Multi-level RoIAlign forward + backward for gfx950 (MI355X), NHWC features.
//
// Replaces SingleRoIExtractor.forward (mmdet/models/roi_extractors/single_level.py:89-107:
// map_roi_levels :54-73, then per level a boolean mask, `inds.any()` host sync, a kernel launch
// and an index_put) together with the reference's RoIAlign kernel
// (mmdet/ops/roi_align/src/roi_align_kernel.cu:16-124, legacy semantics:
//   roi_end = (x2 + 1) * scale, no half-pixel shift, samples with y < -1 or y > H contribute 0).
//
// The reference kernel uses one thread per output element on NCHW features: the 16 taps of
// an output are 16 uncoalesced 4-byte loads.  Here features are NHWC and one wave64 owns one
// output bin (roi, ph, pw) for ALL channels: every tap is a contiguous C*4-byte vector
// (1 KB for C = 256) read as one 16-byte load per lane, fully coalesced; the level of the
// RoI is computed in the kernel (no per-level masks, no host sync, one launch).
// Output layout [K, PH, PW, C] (bin-major, channels contiguous) — the FC that consumes it
// permutes its weight columns once instead.
// Algorithmic bytes per RoI (C=256, 7x7): 50,176 B written + the unique input footprint.
//
// Backward (roi_align_kernel.cu:149-266, only on the `selectp = 0` path): the same wave-per-bin
// mapping scatters g * w / 4 into the level's gradient map with hardware fp32 atomics
// (global_atomic_add_f32), 16 taps x C channels per bin; the maps are accumulated INTO (the RPN
// head's data gradient is already there), so no separate zero-fill + add pass.
//
// HTC semantic fusion (mmdet/models/detectors/htc.py:57-64,88-96): the pooled semantic feature is
// RoIAligned at 14x14, `F.adaptive_avg_pool2d`-ed to the box head's 7x7 and ADDED to the box
// features.  POOL = 2 evaluates the 2x2 block of fine bins (same sample coordinates as the 14x14
// grid) inside the wave and `accumulate` adds into the existing output: the 14x14 intermediate
// (205 MB for 1024 RoIs), the pooling pass and the add pass never touch HBM.
#include <math.h>
#include <stdlib.h>

#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxLevels = 8;

struct RoiLevels {
  const float* feat[kMaxLevels];  // [N, H_l, W_l, C]
  int H[kMaxLevels], W[kMaxLevels];
  float scale[kMaxLevels];        // 1 / stride_l
  int num_levels;
  int num_images;
  float finest_scale;             // 56
};

// bilinear tap weights/offsets of one sample point (roi_align_kernel.cu:16-61)
struct Tap {
  int o[4];
  float w[4];
};

__device__ __forceinline__ Tap make_tap(float y, float x, int H, int W) {
  Tap t;
  // (negated form: a NaN coordinate is treated as out of bounds instead of indexing wildly)
  if (!(y >= -1.0f && y <= (float)H && x >= -1.0f && x <= (float)W)) {
    t.o[0] = t.o[1] = t.o[2] = t.o[3] = 0;
    t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  t.o[0] = y_low * W + x_low;
  t.o[1] = y_low * W + x_high;
  t.o[2] = y_high * W + x_low;
  t.o[3] = y_high * W + x_high;
  t.w[0] = hy * hx;
  t.w[1] = hy * lx;
  t.w[2] = ly * hx;
  t.w[3] = ly * lx;
  return t;
}

// one wave per (roi, ph, pw); lanes stride over channel quads.
// BWD = false: out[k,bin,:] (+)= mean of the bilinear samples.  BWD = true: `out` is the incoming
// gradient and L.feat are the per-level gradient maps that receive the scattered taps.
// POOL: every output bin is the average of POOL x POOL bins of the (PH*POOL) x (PW*POOL) RoIAlign
// grid (adaptive_avg_pool2d of an exact multiple), each with SAMPLES x SAMPLES sample points.
template <int SAMPLES, bool BWD, int POOL, bool ACC>
__global__ __launch_bounds__(256) void roi_align_nhwc_kernel(RoiLevels L,
                                                             const float* __restrict__ rois,
                                                             int K, int C, int PH, int PW,
                                                             float* __restrict__ out,
                                                             int* __restrict__ lvl_out, int xcd_chunk) {
  const int lane = threadIdx.x & 63;
  // xcd_chunk > 0 (A/B, BGS_ROI_XCD=1): workgroup b runs on XCD b % 8 and takes workgroup slot (b % 8) * chunk + b / 8,
  // so that the 49 bins of a RoI (12.25 consecutive workgroups) and the RoIs next to it in the list share ONE L2
  // instead of being dealt round-robin over the eight of them
  const int wg = xcd_chunk > 0 ? (int)((blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int wave_global = wg * 4 + (threadIdx.x >> 6);
  const int bins = PH * PW;
  if (wave_global >= K * bins) return;
  const int k = wave_global / bins;
  const int bin = wave_global - k * bins;
  const int ph = bin / PW, pw = bin - (bin / PW) * PW;

  const float* roi = rois + (size_t)k * 5;
  const int n = min(max((int)roi[0], 0), L.num_images - 1);
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  // map_roi_levels (single_level.py:69-72)
  const float scale = sqrtf((x2 - x1 + 1.f) * (y2 - y1 + 1.f));
  float lf = floorf(log2f(scale / L.finest_scale + 1e-6f));
  lf = fminf(fmaxf(lf, 0.f), (float)(L.num_levels - 1));
  const int lvl = (int)lf;
  if (lvl_out && bin == 0 && lane == 0) lvl_out[k] = lvl;
  const int H = L.H[lvl], W = L.W[lvl];
  const float ss = L.scale[lvl];
  const float* feat = L.feat[lvl] + (size_t)n * H * W * C;

  const float roi_start_w = x1 * ss, roi_start_h = y1 * ss;
  const float roi_end_w = (x2 + 1.f) * ss, roi_end_h = (y2 + 1.f) * ss;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
  const float bin_size_h = roi_height / (PH * POOL), bin_size_w = roi_width / (PW * POOL);
  float* o = out + ((size_t)k * bins + bin) * C;

  f32x4 total = {0.f, 0.f, 0.f, 0.f};     // forward, POOL > 1 (C <= 256: one quad per lane)
#pragma unroll 1
  for (int sub = 0; sub < POOL * POOL; ++sub) {
    const int fh = ph * POOL + sub / POOL, fw = pw * POOL + sub % POOL;   // fine-grid bin
    Tap taps[SAMPLES * SAMPLES];
#pragma unroll
    for (int iy = 0; iy < SAMPLES; ++iy) {
      const float y = roi_start_h + fh * bin_size_h + (iy + .5f) * bin_size_h / (float)SAMPLES;
#pragma unroll
      for (int ix = 0; ix < SAMPLES; ++ix) {
        const float x = roi_start_w + fw * bin_size_w + (ix + .5f) * bin_size_w / (float)SAMPLES;
        taps[iy * SAMPLES + ix] = make_tap(y, x, H, W);
      }
    }
    if (BWD) {
      float* dfeat = const_cast<float*>(feat);
      // consecutive lanes -> consecutive channels: one atomic wave-instruction covers 64
      // consecutive floats = two full 128-byte lines (a lane-owns-a-quad mapping touches eight
      // quarter-used lines per instruction: 2.67 -> measured below in profiles/r2p)
      for (int c = lane; c < C; c += 64) {
        const float g = o[c] / (float)(SAMPLES * SAMPLES * POOL * POOL);
#pragma unroll
        for (int s = 0; s < SAMPLES * SAMPLES; ++s) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float w = taps[s].w[q];
            if (w == 0.f) continue;      // out-of-bounds sample or a degenerate (clamped) tap
            unsafeAtomicAdd(dfeat + (size_t)taps[s].o[q] * C + c, g * w);
          }
        }
      }
      continue;
    }
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SAMPLES * SAMPLES; ++s) {
        // val = w1*lt + w2*rt + w3*lb + w4*rb, summed over the samples in the reference's order
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(feat + (size_t)taps[s].o[q] * C + c);
          v += taps[s].w[q] * d;
        }
        acc += v;
      }
      acc /= (float)(SAMPLES * SAMPLES);
      if (POOL == 1) {
        if (ACC) acc += *reinterpret_cast<const f32x4*>(o + c);
        *reinterpret_cast<f32x4*>(o + c) = acc;
      } else {
        total += acc;      // the launcher guarantees C <= 256 here: a single quad per lane
      }
    }
  }
  if (!BWD && POOL > 1) {
    const int c = lane * 4;
    if (c < C) {
      total /= (float)(POOL * POOL);
      if (ACC) total += *reinterpret_cast<const f32x4*>(o + c);
      *reinterpret_cast<f32x4*>(o + c) = total;
    }
  }
}


// ---- forward, sample_num = 2, with every distinct feature pixel of a bin loaded ONCE (round 5).
// The 2 x 2 sample points of a bin have 2 x 2 bilinear corners each; sample (iy, ix) uses rows {y_low[iy], y_high[iy]}
// and columns {x_low[ix], x_high[ix]}, so the 16 taps of a bin are a 4 x 4 grid rows x columns.  RoIs are mapped to the
// level where they measure 14 - 28 pixels, i.e. sample points lie 1 - 2 pixels apart: neighbouring samples share a row
// and / or a column of that grid more often than not (x_low[1] == x_high[0], or both samples in one pixel cell).  One wave
// owns one bin, so rows, columns and their equalities are wave-uniform scalars: a duplicate row / column is not loaded
// (scalar branch) and its registers are copied from the first occurrence afterwards.  The arithmetic — weights, products,
// summation order — is that of roi_align_nhwc_kernel: same bits (tests/test_gpu_det_ops.py).  What it saves is L1 / L2
// request traffic: 16 KB per bin and wave fall to 9 - 16 KB (the launch moves 0.82 GB through the L2s in ~100 us).
// An out-of-range sample has weight 0 on whatever finite pixel its clamped row / column names (the one-sample-at-a-time
// kernel reads pixel 0 for it): equal unless the feature map holds non-finite values.
struct AxisTap {
  int lo, hi;
  float l, h;
  bool ok;
};

__device__ __forceinline__ AxisTap make_axis_tap(float v, int n) {
  AxisTap t;
  t.ok = (v >= -1.0f && v <= (float)n);
  if (!t.ok) {
    t.lo = t.hi = 0;
    t.l = t.h = 0.f;
    return t;
  }
  if (v <= 0.f) v = 0.f;
  int lo = (int)v;
  if (lo >= n - 1) {
    t.hi = lo = n - 1;
    v = (float)lo;
  } else {
    t.hi = lo + 1;
  }
  t.lo = lo;
  t.l = v - lo;
  t.h = 1.f - t.l;
  return t;
}

template <int POOL, bool ACC>
__global__ __launch_bounds__(256) void roi_align_fwd_grid_kernel(RoiLevels L, const float* __restrict__ rois, int K,
                                                                 int C, int PH, int PW, float* __restrict__ out,
                                                                 int* __restrict__ lvl_out, int xcd_chunk) {
  const int lane = threadIdx.x & 63;
  const int wg = xcd_chunk > 0 ? (int)((blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int wave_global = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
  const int bins = PH * PW;
  if (wave_global >= K * bins) return;
  const int k = wave_global / bins;
  const int bin = wave_global - k * bins;
  const int ph = bin / PW, pw = bin - (bin / PW) * PW;

  const float* roi = rois + (size_t)k * 5;
  const int n = min(max((int)roi[0], 0), L.num_images - 1);
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float scale = sqrtf((x2 - x1 + 1.f) * (y2 - y1 + 1.f));
  float lf = floorf(log2f(scale / L.finest_scale + 1e-6f));
  lf = fminf(fmaxf(lf, 0.f), (float)(L.num_levels - 1));
  const int lvl = __builtin_amdgcn_readfirstlane((int)lf);
  if (lvl_out && bin == 0 && lane == 0) lvl_out[k] = lvl;
  const int H = L.H[lvl], W = L.W[lvl];
  const float ss = L.scale[lvl];
  const float* feat = L.feat[lvl] + (size_t)n * H * W * C;

  const float roi_start_w = x1 * ss, roi_start_h = y1 * ss;
  const float roi_end_w = (x2 + 1.f) * ss, roi_end_h = (y2 + 1.f) * ss;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
  const float bin_size_h = roi_height / (PH * POOL), bin_size_w = roi_width / (PW * POOL);
  float* o = out + ((size_t)k * bins + bin) * C;

  f32x4 total = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int sub = 0; sub < POOL * POOL; ++sub) {
    const int fh = ph * POOL + sub / POOL, fw = pw * POOL + sub % POOL;
    AxisTap ty[2], tx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ty[i] = make_axis_tap(roi_start_h + fh * bin_size_h + (i + .5f) * bin_size_h / 2.f, H);
      tx[i] = make_axis_tap(roi_start_w + fw * bin_size_w + (i + .5f) * bin_size_w / 2.f, W);
    }
    // the 4 rows / 4 columns of the tap grid and, per row / column, the first one with the same index
    int row[4] = {ty[0].lo, ty[0].hi, ty[1].lo, ty[1].hi};
    int col[4] = {tx[0].lo, tx[0].hi, tx[1].lo, tx[1].hi};
    int rsrc[4], csrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      row[i] = __builtin_amdgcn_readfirstlane(row[i]);
      col[i] = __builtin_amdgcn_readfirstlane(col[i]);
      rsrc[i] = csrc[i] = i;
#pragma unroll
      for (int j = i - 1; j >= 0; --j) {
        if (row[j] == row[i]) rsrc[i] = j;
        if (col[j] == col[i]) csrc[i] = j;
      }
    }
    // weights of sample (iy, ix), corner (a, b): as make_tap (hy*hx, hy*lx, ly*hx, ly*lx; all 0 for a sample out of range)
    float wy[2][2], wx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wy[i][0] = ty[i].h; wy[i][1] = ty[i].l;
      wx[i][0] = tx[i].h; wx[i][1] = tx[i].l;
    }
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 P[4][4];
      const float* base = feat + c;
      // phase 1: the loads (a scalar branch each; nothing in here reads a loaded register)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (rsrc[r] == r && csrc[q] == q)
            P[r][q] = *reinterpret_cast<const f32x4*>(base + (size_t)(row[r] * W + col[q]) * C);
      // phase 2: duplicate columns of the loaded rows, then duplicate rows
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (rsrc[r] == r && csrc[q] != q) {
#pragma unroll
            for (int j = 0; j < q; ++j)
              if (csrc[q] == j) P[r][q] = P[r][j];
          }
#pragma unroll
      for (int r = 1; r < 4; ++r)
        if (rsrc[r] != r) {
#pragma unroll
          for (int j = 0; j < r; ++j)
            if (rsrc[r] == j) {
#pragma unroll
              for (int q = 0; q < 4; ++q) P[r][q] = P[j][q];
            }
        }
      // phase 3: the reference's arithmetic, sample by sample
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          const bool ok = ty[iy].ok && tx[ix].ok;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const float w = ok ? wy[iy][a] * wx[ix][b] : 0.f;
              v += w * P[2 * iy + a][2 * ix + b];
            }
          acc += v;
        }
      acc /= 4.f;
      if (POOL == 1) {
        if (ACC) acc += *reinterpret_cast<const f32x4*>(o + c);
        *reinterpret_cast<f32x4*>(o + c) = acc;
      } else {
        total += acc;
      }
    }
  }
  if (POOL > 1) {
    const int c = lane * 4;
    if (c < C) {
      total /= (float)(POOL * POOL);
      if (ACC) total += *reinterpret_cast<const f32x4*>(o + c);
      *reinterpret_cast<f32x4*>(o + c) = total;
    }
  }
}


// ---- backward, sample_num = 2, on the same 4 x 4 tap grid: the weights of the taps that fall on one pixel are summed
// first (wave-uniform scalars), so a bin issues one atomic per DISTINCT pixel and channel instead of one per tap (9.1
// instead of 16 on the RoIs of a cfg[1] step).  fp32 atomics accumulate in arbitrary order anyway, so the result is
// the reference's up to the rounding of a sum — the same statement the one-atomic-per-tap kernel makes.
template <int POOL>
__global__ __launch_bounds__(256) void roi_align_bwd_grid_kernel(RoiLevels L, const float* __restrict__ rois, int K,
                                                                 int C, int PH, int PW, const float* __restrict__ dout) {
  const int lane = threadIdx.x & 63;
  const int wave_global = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  const int bins = PH * PW;
  if (wave_global >= K * bins) return;
  const int k = wave_global / bins;
  const int bin = wave_global - k * bins;
  const int ph = bin / PW, pw = bin - (bin / PW) * PW;
  const float* roi = rois + (size_t)k * 5;
  const int n = min(max((int)roi[0], 0), L.num_images - 1);
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float scale = sqrtf((x2 - x1 + 1.f) * (y2 - y1 + 1.f));
  float lf = floorf(log2f(scale / L.finest_scale + 1e-6f));
  lf = fminf(fmaxf(lf, 0.f), (float)(L.num_levels - 1));
  const int lvl = __builtin_amdgcn_readfirstlane((int)lf);
  const int H = L.H[lvl], W = L.W[lvl];
  const float ss = L.scale[lvl];
  float* dfeat = const_cast<float*>(L.feat[lvl]) + (size_t)n * H * W * C;
  const float roi_start_w = x1 * ss, roi_start_h = y1 * ss;
  const float roi_end_w = (x2 + 1.f) * ss, roi_end_h = (y2 + 1.f) * ss;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
  const float bin_size_h = roi_height / (PH * POOL), bin_size_w = roi_width / (PW * POOL);
  const float* o = dout + ((size_t)k * bins + bin) * C;
#pragma unroll 1
  for (int sub = 0; sub < POOL * POOL; ++sub) {
    const int fh = ph * POOL + sub / POOL, fw = pw * POOL + sub % POOL;
    AxisTap ty[2], tx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ty[i] = make_axis_tap(roi_start_h + fh * bin_size_h + (i + .5f) * bin_size_h / 2.f, H);
      tx[i] = make_axis_tap(roi_start_w + fw * bin_size_w + (i + .5f) * bin_size_w / 2.f, W);
    }
    int row[4] = {ty[0].lo, ty[0].hi, ty[1].lo, ty[1].hi};
    int col[4] = {tx[0].lo, tx[0].hi, tx[1].lo, tx[1].hi};
    int rsrc[4], csrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      row[i] = __builtin_amdgcn_readfirstlane(row[i]);
      col[i] = __builtin_amdgcn_readfirstlane(col[i]);
      rsrc[i] = csrc[i] = i;
#pragma unroll
      for (int j = i - 1; j >= 0; --j) {
        if (row[j] == row[i]) rsrc[i] = j;
        if (col[j] == col[i]) csrc[i] = j;
      }
    }
    // the 4 x 4 table of tap weights (make_tap's hy*hx, hy*lx, ly*hx, ly*lx per sample; 0 for a sample out of range)
    float wg[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) wg[r][q] = 0.f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        const bool ok = ty[iy].ok && tx[ix].ok;
        const float wy0 = ty[iy].h, wy1 = ty[iy].l, wx0 = tx[ix].h, wx1 = tx[ix].l;
        wg[2 * iy][2 * ix] = ok ? wy0 * wx0 : 0.f;
        wg[2 * iy][2 * ix + 1] = ok ? wy0 * wx1 : 0.f;
        wg[2 * iy + 1][2 * ix] = ok ? wy1 * wx0 : 0.f;
        wg[2 * iy + 1][2 * ix + 1] = ok ? wy1 * wx1 : 0.f;
      }
    // every entry (r, q) of the table is one tap of one sample (sample (r >> 1, q >> 1)); two entries name the same
    // pixel iff their rows and their columns are equal, so folding duplicate columns and then duplicate rows onto
    // their first occurrence sums exactly the taps that land on one pixel
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 3; q >= 1; --q)
#pragma unroll
        for (int j = 0; j < q; ++j)
          if (csrc[q] == j) {
            wg[r][j] += wg[r][q];
            wg[r][q] = 0.f;
          }
#pragma unroll
    for (int r = 3; r >= 1; --r)
#pragma unroll
      for (int j = 0; j < r; ++j)
        if (rsrc[r] == j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            wg[j][q] += wg[r][q];
            wg[r][q] = 0.f;
          }
        }
    for (int c = lane; c < C; c += 64) {
      const float g = o[c] / (float)(4 * POOL * POOL);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float w = wg[r][q];
          if (w == 0.f) continue;
          unsafeAtomicAdd(dfeat + (size_t)(row[r] * W + col[q]) * C + c, g * w);
        }
    }
  }
}


// ---- any sample_num (0 = the reference's adaptive grid: ceil(roi_size / pooled_size) samples per bin and axis,
// roi_align_kernel.cu:95-99) and fp16 tensors (AT_DISPATCH_FLOATING_TYPES_AND_HALF, roi_align_kernel.cu:136):
// the rest of the `roi_align_cuda` interface.  Same wave-per-bin mapping; the taps of a sample point are
// recomputed in the loop instead of held in a register array (their number is a per-RoI runtime value), features /
// output (forward) and the incoming gradient (backward) are T = float or _Float16 with fp32 arithmetic, the
// gradient maps are always fp32 (the caller adds them into its own-dtype bottom_grad: mmdet/ops/roi_align/
// roi_align.py:45-52).  A degenerate RoI (zero height or width under sample_num = 0) has an empty sample grid
// and produces 0 / 0 = NaN, as the reference kernel does.
template <typename T>
__device__ __forceinline__ f32x4 load4_as_f32(const T* p);
template <>
__device__ __forceinline__ f32x4 load4_as_f32<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <>
__device__ __forceinline__ f32x4 load4_as_f32<_Float16>(const _Float16* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(_Float16* p, f32x4 v) {
  *reinterpret_cast<f16x4*>(p) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roi_align_nhwc_generic_kernel(RoiLevels L, const float* __restrict__ rois,
                                                                     int K, int C, int PH, int PW, int sample_num,
                                                                     T* __restrict__ out,
                                                                     int* __restrict__ lvl_out) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int bins = PH * PW;
  if (wave_global >= K * bins) return;
  const int k = wave_global / bins;
  const int bin = wave_global - k * bins;
  const int ph = bin / PW, pw = bin - (bin / PW) * PW;
  const float* roi = rois + (size_t)k * 5;
  const int n = min(max((int)roi[0], 0), L.num_images - 1);
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float scale = sqrtf((x2 - x1 + 1.f) * (y2 - y1 + 1.f));
  float lf = floorf(log2f(scale / L.finest_scale + 1e-6f));
  lf = fminf(fmaxf(lf, 0.f), (float)(L.num_levels - 1));
  const int lvl = (int)lf;
  if (lvl_out && bin == 0 && lane == 0) lvl_out[k] = lvl;
  const int H = L.H[lvl], W = L.W[lvl];
  const float ss = L.scale[lvl];
  const float roi_start_w = x1 * ss, roi_start_h = y1 * ss;
  const float roi_end_w = (x2 + 1.f) * ss, roi_end_h = (y2 + 1.f) * ss;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
  const float roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
  const float bin_size_h = roi_height / PH, bin_size_w = roi_width / PW;
  const int sh = sample_num > 0 ? sample_num : (int)ceilf(roi_height / PH);
  const int sw = sample_num > 0 ? sample_num : (int)ceilf(roi_width / PW);
  const float count = (float)(sh * sw);
  T* o = out + ((size_t)k * bins + bin) * C;
  if (BWD) {
    float* dfeat = const_cast<float*>(L.feat[lvl]) + (size_t)n * H * W * C;
    for (int c = lane; c < C; c += 64) {
      const float g = (float)o[c];
      for (int iy = 0; iy < sh; ++iy) {
        const float y = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)sh;
        for (int ix = 0; ix < sw; ++ix) {
          const float x = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)sw;
          const Tap t = make_tap(y, x, H, W);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (t.w[q] == 0.f) continue;
            unsafeAtomicAdd(dfeat + (size_t)t.o[q] * C + c, g * t.w[q] / count);
          }
        }
      }
    }
    return;
  }
  const T* feat = reinterpret_cast<const T*>(L.feat[lvl]) + (size_t)n * H * W * C;
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < sh; ++iy) {
      const float y = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)sh;
      for (int ix = 0; ix < sw; ++ix) {
        const float x = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)sw;
        const Tap t = make_tap(y, x, H, W);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) v += t.w[q] * load4_as_f32<T>(feat + (size_t)t.o[q] * C + c);
        acc += v;
      }
    }
    acc /= count;
    store4(o + c, acc);
  }
}

}  // namespace

static int fill_levels(RoiLevels& L, const float* const* feats, const int* host_heights,
                       const int* host_widths, const float* host_scales, int num_levels,
                       int num_images, float finest_scale, bool need_align) {
  for (int i = 0; i < kMaxLevels; ++i) {
    L.feat[i] = nullptr;
    L.H[i] = L.W[i] = 1;
    L.scale[i] = 1.f;
  }
  for (int i = 0; i < num_levels; ++i) {
    if (!feats[i] || (need_align && (uintptr_t)feats[i] % 16 != 0)) return BGS_ERR_INVALID_ARG;
    L.feat[i] = feats[i];
    L.H[i] = host_heights[i];
    L.W[i] = host_widths[i];
    L.scale[i] = host_scales[i];
  }
  L.num_levels = num_levels;
  L.num_images = num_images;
  L.finest_scale = finest_scale;
  return BGS_OK;
}

extern "C" int bgs_roi_align_nhwc_fwd_ex(const float* const* host_feats, const int* host_heights,
                                         const int* host_widths, const float* host_scales,
                                         int num_levels, int num_images, float finest_scale,
                                         const float* rois, int K, int C, int pooled_h,
                                         int pooled_w, int sample_num, int pool, int accumulate,
                                         float* out, int* levels_out, bgs_stream_t stream) {
  if (num_levels <= 0 || num_levels > kMaxLevels || num_images <= 0 || K < 0 || C <= 0 || pooled_h <= 0 ||
      pooled_w <= 0)
    return BGS_ERR_INVALID_ARG;
  if (!host_feats || !host_heights || !host_widths || !host_scales) return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!rois || !out) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || (uintptr_t)out % 16 != 0) return BGS_ERR_UNSUPPORTED;
  if (sample_num < 0) return BGS_ERR_INVALID_ARG;
  if (pool != 1 && !(pool == 2 && C <= 256)) return BGS_ERR_UNSUPPORTED;
  if (sample_num != 2 && pool != 1) return BGS_ERR_UNSUPPORTED;   // the fused 2x2 pooling exists for sample_num = 2
  RoiLevels L;
  const int rc = fill_levels(L, host_feats, host_heights, host_widths, host_scales, num_levels,
                             num_images, finest_scale, true);
  if (rc != BGS_OK) return rc;
  const long long waves = (long long)K * pooled_h * pooled_w;
  unsigned grid = (unsigned)((waves + 3) / 4);
  // (default on: 86.1 -> 78.4 us for the 1024 RoIs of a cfg[1] step, 6.343 -> 6.318 ms per step, profiles/r9a / r9d;
  //  BGS_ROI_XCD=0 is the A/B arm, read at every call)
  const char* xcd_env = getenv("BGS_ROI_XCD");
  const int xcd_mode = xcd_env ? atoi(xcd_env) : 1;
  int xcd_chunk = 0;
  if (xcd_mode > 0 && sample_num == 2 && grid >= 64) {
    xcd_chunk = (int)((grid + 7) / 8);
    grid = (unsigned)(8 * xcd_chunk);
  }
  if (sample_num != 2) {      // every shipped config uses sample_num = 2; the rest of the interface runs the generic kernel
    if (accumulate) return BGS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((roi_align_nhwc_generic_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       L, rois, K, C, pooled_h, pooled_w, sample_num, out, levels_out);
    BGS_RETURN_LAUNCH_STATUS();
  }
  // BGS_ROI_DEDUP=0: the one-sample-at-a-time kernel (16 loads per bin and channel quad); default: the tap-grid kernel
  const char* dd_env = getenv("BGS_ROI_DEDUP");
  const bool dedup = !(dd_env && atoi(dd_env) == 0);
#define BGS_ROI_FWD(POOL_, ACC_)                                                                         \
  do {                                                                                                   \
    if (dedup)                                                                                           \
      hipLaunchKernelGGL((roi_align_fwd_grid_kernel<POOL_, ACC_>), dim3(grid), dim3(256), 0,             \
                         (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w, out, levels_out, xcd_chunk); \
    else                                                                                                 \
      hipLaunchKernelGGL((roi_align_nhwc_kernel<2, false, POOL_, ACC_>), dim3(grid), dim3(256), 0,       \
                         (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w, out, levels_out, xcd_chunk); \
  } while (0)
  if (pool == 1 && !accumulate) BGS_ROI_FWD(1, false);
  else if (pool == 1) BGS_ROI_FWD(1, true);
  else if (!accumulate) BGS_ROI_FWD(2, false);
  else BGS_ROI_FWD(2, true);
#undef BGS_ROI_FWD
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_roi_align_nhwc_fwd(const float* const* host_feats, const int* host_heights,
                                      const int* host_widths, const float* host_scales,
                                      int num_levels, int num_images, float finest_scale,
                                      const float* rois,
                                      int K, int C, int pooled_h, int pooled_w, int sample_num,
                                      float* out, int* levels_out, bgs_stream_t stream) {
  return bgs_roi_align_nhwc_fwd_ex(host_feats, host_heights, host_widths, host_scales, num_levels,
                                   num_images, finest_scale, rois, K, C, pooled_h, pooled_w,
                                   sample_num, 1, 0, out, levels_out, stream);
}

extern "C" int bgs_roi_align_nhwc_bwd_ex(float* const* host_dfeats, const int* host_heights,
                                         const int* host_widths, const float* host_scales,
                                         int num_levels, int num_images, float finest_scale,
                                         const float* rois, int K, int C, int pooled_h,
                                         int pooled_w, int sample_num, int pool, const float* dout,
                                         bgs_stream_t stream) {
  if (num_levels <= 0 || num_levels > kMaxLevels || num_images <= 0 || K < 0 || C <= 0 ||
      pooled_h <= 0 || pooled_w <= 0)
    return BGS_ERR_INVALID_ARG;
  if (!host_dfeats || !host_heights || !host_widths || !host_scales) return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!rois || !dout) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || (uintptr_t)dout % 16 != 0) return BGS_ERR_UNSUPPORTED;
  if (sample_num < 0) return BGS_ERR_INVALID_ARG;
  if ((pool != 1 && pool != 2) || (sample_num != 2 && pool != 1)) return BGS_ERR_UNSUPPORTED;
  RoiLevels L;
  const int rc = fill_levels(L, host_dfeats, host_heights, host_widths, host_scales, num_levels,
                             num_images, finest_scale, false);
  if (rc != BGS_OK) return rc;
  const long long waves = (long long)K * pooled_h * pooled_w;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (sample_num != 2) {
    hipLaunchKernelGGL((roi_align_nhwc_generic_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       L, rois, K, C, pooled_h, pooled_w, sample_num, const_cast<float*>(dout), nullptr);
    BGS_RETURN_LAUNCH_STATUS();
  }
  // BGS_ROI_DEDUP=0: one atomic per tap; default: one per distinct pixel of a bin (roi_align_bwd_grid_kernel)
  const char* dd_env = getenv("BGS_ROI_DEDUP");
  const bool dedup = !(dd_env && atoi(dd_env) == 0);
  if (dedup && pool == 1)
    hipLaunchKernelGGL((roi_align_bwd_grid_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, L, rois, K, C,
                       pooled_h, pooled_w, dout);
  else if (dedup)
    hipLaunchKernelGGL((roi_align_bwd_grid_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, L, rois, K, C,
                       pooled_h, pooled_w, dout);
  else if (pool == 1)
    hipLaunchKernelGGL((roi_align_nhwc_kernel<2, true, 1, false>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w,
                       const_cast<float*>(dout), nullptr, 0);
  else
    hipLaunchKernelGGL((roi_align_nhwc_kernel<2, true, 2, false>), dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w,
                       const_cast<float*>(dout), nullptr, 0);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_roi_align_nhwc_bwd(float* const* host_dfeats, const int* host_heights,
                                      const int* host_widths, const float* host_scales,
                                      int num_levels, int num_images, float finest_scale,
                                      const float* rois, int K, int C, int pooled_h, int pooled_w,
                                      int sample_num, const float* dout, bgs_stream_t stream) {
  return bgs_roi_align_nhwc_bwd_ex(host_dfeats, host_heights, host_widths, host_scales, num_levels,
                                   num_images, finest_scale, rois, K, C, pooled_h, pooled_w,
                                   sample_num, 1, dout, stream);
}

// fp16 tensors (the half instantiation of the reference's dispatch, roi_align_kernel.cu:136,281): features / output
// (forward) and the incoming gradient (backward) are IEEE half, RoIs fp32 (the caller widens them: exact), arithmetic
// and the gradient maps fp32.  Any sample_num >= 0 (0 = adaptive).
extern "C" int bgs_roi_align_nhwc_fwd_f16(const void* const* host_feats, const int* host_heights,
                                          const int* host_widths, const float* host_scales, int num_levels,
                                          int num_images, float finest_scale, const float* rois, int K, int C,
                                          int pooled_h, int pooled_w, int sample_num, void* out,
                                          int* levels_out, bgs_stream_t stream) {
  if (num_levels <= 0 || num_levels > kMaxLevels || num_images <= 0 || K < 0 || C <= 0 || pooled_h <= 0 ||
      pooled_w <= 0 || sample_num < 0)
    return BGS_ERR_INVALID_ARG;
  if (!host_feats || !host_heights || !host_widths || !host_scales) return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!rois || !out) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || (uintptr_t)out % 8 != 0) return BGS_ERR_UNSUPPORTED;
  RoiLevels L;
  for (int i = 0; i < num_levels; ++i)
    if (!host_feats[i] || (uintptr_t)host_feats[i] % 8 != 0) return BGS_ERR_INVALID_ARG;
  const int rc = fill_levels(L, reinterpret_cast<const float* const*>(host_feats), host_heights, host_widths,
                             host_scales, num_levels, num_images, finest_scale, false);
  if (rc != BGS_OK) return rc;
  const long long waves = (long long)K * pooled_h * pooled_w;
  hipLaunchKernelGGL((roi_align_nhwc_generic_kernel<_Float16, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w, sample_num,
                     reinterpret_cast<_Float16*>(out), levels_out);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_roi_align_nhwc_bwd_f16(float* const* host_dfeats, const int* host_heights,
                                          const int* host_widths, const float* host_scales, int num_levels,
                                          int num_images, float finest_scale, const float* rois, int K, int C,
                                          int pooled_h, int pooled_w, int sample_num, const void* dout,
                                          bgs_stream_t stream) {
  if (num_levels <= 0 || num_levels > kMaxLevels || num_images <= 0 || K < 0 || C <= 0 || pooled_h <= 0 ||
      pooled_w <= 0 || sample_num < 0)
    return BGS_ERR_INVALID_ARG;
  if (!host_dfeats || !host_heights || !host_widths || !host_scales) return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!rois || !dout) return BGS_ERR_INVALID_ARG;
  RoiLevels L;
  const int rc = fill_levels(L, host_dfeats, host_heights, host_widths, host_scales, num_levels, num_images,
                             finest_scale, false);
  if (rc != BGS_OK) return rc;
  const long long waves = (long long)K * pooled_h * pooled_w;
  hipLaunchKernelGGL((roi_align_nhwc_generic_kernel<_Float16, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, L, rois, K, C, pooled_h, pooled_w, sample_num,
                     const_cast<_Float16*>(reinterpret_cast<const _Float16*>(dout)), nullptr);
  BGS_RETURN_LAUNCH_STATUS();
}
