// Fused target-assignment / RPN-loss / proposal-decode / RoI-target kernels for gfx950.
//
// These replace the long chains of tiny tensor ops (and, in the reference, the host round
// trips) around the detector's two sampling stages:
//   bgs_iou_assign        MaxIoUAssigner.assign / assign_wrt_overlaps
//                         (mmdet/core/bbox/assigners/max_iou_assigner.py:47-180) incl.
//                         bbox_overlaps (mmdet/core/bbox/geometry.py:4-63): the reference builds the
//                         [G, A] IoU matrix (5.4 M floats per image for the RPN), moves to the CPU
//                         when G > 50 and loops over the GTs in Python.  Here: two passes over the
//                         boxes with the GTs in LDS, nothing materialised.
//   bgs_rpn_loss          anchor_target_single's target encoding (mmdet/core/anchor/
//                         anchor_target.py:118-152) + AnchorHead.loss_single
//                         (mmdet/models/anchor_heads/anchor_head.py:142-161): sigmoid BCE +
//                         SmoothL1 over the SAMPLED anchors only, per FPN level.
//   bgs_decode_proposals  RPNHead.get_bboxes_single's gather + delta2bbox + clamp
//                         (mmdet/models/anchor_heads/rpn_head.py:62-90, mmdet/core/bbox/
//                         transforms.py:34-111).
//   bgs_rcnn_targets      bbox2roi + bbox_target_single (mmdet/core/bbox/transforms.py:149-168,
//                         mmdet/core/bbox/bbox_target.py:35-61) on the fixed-size samples.
// All are tiny / latency-bound; the point is launch count (a few hundred launches fewer per
// iteration) and zero host involvement.
#include <math.h>
#include <string.h>

#include "bgs_common.h"

namespace {

constexpr int kMaxImgs = 16;
constexpr int kMaxLevels = 8;
constexpr int kGtChunk = 256;

struct ImgTable {
  int gt_off[kMaxImgs + 1];  // offsets into the concatenated gt array
  int img_h[kMaxImgs], img_w[kMaxImgs];
};

// legacy "+1" IoU, same operation order as geometry.py:36-63 (products and sums only, so the
// result is bitwise the same in both assignment passes and in the tensor-op restatement)
__device__ __forceinline__ float iou1(float ax1, float ay1, float ax2, float ay2, float aarea,
                                      float bx1, float by1, float bx2, float by2) {
#pragma clang fp contract(off)   // separately rounded mul/add, exactly like the tensor-op form
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1) + 1.f, 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1) + 1.f, 0.f);
  const float ov = w * h;
  const float barea = (bx2 - bx1 + 1.f) * (by2 - by1 + 1.f);
  return ov / (barea + aarea - ov);   // overlaps = gt x boxes: area1 = gt (b), area2 = box (a)
}

// w * h > 0 of iou1 (same expressions): false -> iou1 returns +0.0
__device__ __forceinline__ bool iou_overlaps(float ax1, float ay1, float ax2, float ay2,
                                             float bx1, float by1, float bx2, float by2) {
#pragma clang fp contract(off)
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1) + 1.f, 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1) + 1.f, 0.f);
  return w * h > 0.f;
}

// pass 1: per box max / argmax over the gts -> the threshold part of the assignment (written here);
// per gt the maximum over the (valid) boxes, globally (atomics) and per workgroup (`blk_max`, for pass 2).
// Only maxima that can matter travel to the global table: pass 2 ignores a gt whose maximum is below
// min_pos_iou, so a workgroup whose best IoU with a gt is below it keeps quiet (before: every workgroup sent
// one atomicMax per gt, IoU 0 included — 21,000 same-address atomics per image were 2/3 of the 28 us launch).
__global__ __launch_bounds__(256) void iou_gtmax_kernel(const float* __restrict__ boxes,
                                                        long long box_img_stride, int box_stride,
                                                        const uint8_t* __restrict__ valid,
                                                        const float* __restrict__ gt, ImgTable T,
                                                        int A, int gmax_stride, float pos_thr, float neg_lo,
                                                        float neg_hi, int send_bits,
                                                        float* __restrict__ box_max,
                                                        int* __restrict__ blk_max,
                                                        int* __restrict__ gt_max_bits,
                                                        int* __restrict__ assigned) {
  __shared__ float sgt[kGtChunk][4];
  __shared__ int smax[kGtChunk];     // per-gt maximum over this block's boxes (float bits)
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int g0 = T.gt_off[n], G = T.gt_off[n + 1] - g0;
  const bool live = i < A;
  const bool ok = live && (valid ? valid[(size_t)n * A + i] != 0 : true);
  float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
  if (live) {
    const float* b = boxes + (size_t)n * box_img_stride + (size_t)i * box_stride;
    x1 = b[0]; y1 = b[1]; x2 = b[2]; y2 = b[3];
  }
  const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
  int* bm = blk_max + ((size_t)n * gridDim.x + blockIdx.x) * gmax_stride;
  // Bounding box of the workgroup's valid boxes: 256 consecutive anchors are 85 neighbouring locations of one
  // pyramid level, so a gt that misses this box misses every one of them — min / max are exact and fp subtraction,
  // addition and the product of non-negatives are monotone, so "the bounding box does not overlap" implies
  // w * h == 0 for every lane (IoU exactly +0.0).  Such gts (~85 % of the (workgroup, gt) pairs on the fine levels)
  // are not visited at all: their effect on the outputs — best = 0 / argmax = 0 for a valid box that overlaps
  // nothing, a workgroup maximum of 0 — is what the initial values below already say.
  __shared__ float sbb[4][4];
  __shared__ int s_cnt[4];
  __shared__ unsigned char slist[kGtChunk];
  {
    const float inf = __builtin_inff();
    const float lx1 = bgs::wave_max(ok ? -x1 : -inf), ly1 = bgs::wave_max(ok ? -y1 : -inf);   // = -min
    const float lx2 = bgs::wave_max(ok ? x2 : -inf), ly2 = bgs::wave_max(ok ? y2 : -inf);
    if (lane == 0) {
      const int w = threadIdx.x >> 6;
      sbb[w][0] = lx1; sbb[w][1] = ly1; sbb[w][2] = lx2; sbb[w][3] = ly2;
    }
  }
  __syncthreads();
  const float bb_x1 = -fmaxf(fmaxf(sbb[0][0], sbb[1][0]), fmaxf(sbb[2][0], sbb[3][0]));
  const float bb_y1 = -fmaxf(fmaxf(sbb[0][1], sbb[1][1]), fmaxf(sbb[2][1], sbb[3][1]));
  const float bb_x2 = fmaxf(fmaxf(sbb[0][2], sbb[1][2]), fmaxf(sbb[2][2], sbb[3][2]));
  const float bb_y2 = fmaxf(fmaxf(sbb[0][3], sbb[1][3]), fmaxf(sbb[2][3], sbb[3][3]));
  const bool any_ok = bb_x2 > -__builtin_inff();           // workgroup-uniform
  // (= after a gt with IoU 0: first maximum, argmax 0.  An image WITHOUT gts inside a batch that has some keeps
  //  best = -1 -> every anchor "ignored" (-1), as before the bounding-box skip existed; the reference raises
  //  'No gt or bboxes' for such an image, max_iou_assigner.py:83-84, so no sampled negatives may come from it)
  float best = (ok && G > 0) ? 0.f : -1.f;
  int barg = 0;
  for (int c0 = 0; c0 < G; c0 += kGtChunk) {
    const int cn = min(kGtChunk, G - c0);
    __syncthreads();
    for (int t = threadIdx.x; t < cn * 4; t += 256) sgt[t >> 2][t & 3] = gt[(size_t)(g0 + c0) * 4 + t];
    for (int t = threadIdx.x; t < cn; t += 256) smax[t] = any_ok ? 0 : (int)0xBF800000;   // 0.0f / -1.0f
    __syncthreads();
    // the gts of this chunk that touch the bounding box, in ascending order (thread t tests gt t)
    {
      const int t = threadIdx.x;
      const bool touch = t < cn && iou_overlaps(bb_x1, bb_y1, bb_x2, bb_y2, sgt[t][0], sgt[t][1], sgt[t][2], sgt[t][3]);
      const unsigned long long m = __ballot(touch);
      if (lane == 0) s_cnt[t >> 6] = __popcll(m);
      __syncthreads();
      int off = 0;
      for (int w = 0; w < (t >> 6); ++w) off += s_cnt[w];
      if (touch) slist[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)t;
      __syncthreads();
    }
    const int nt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    for (int k = 0; k < nt; ++k) {
      const int g = slist[k];
      // most waves of anchors still miss the gt: overlap 0 -> IoU exactly +0.0 (0 / positive), and the
      // ~25 instructions of the exact division are skipped wave-uniformly
      float v = ok ? 0.f : -1.f;
      if (__ballot(ok && iou_overlaps(x1, y1, x2, y2, sgt[g][0], sgt[g][1], sgt[g][2], sgt[g][3])))
        v = ok ? iou1(x1, y1, x2, y2, area, sgt[g][0], sgt[g][1], sgt[g][2], sgt[g][3]) : -1.f;
      if (v > best) {   // first maximum wins (torch.max semantics on ties)
        best = v;
        barg = c0 + g;
      }
      // per-gt maximum: wave reduction, one LDS atomic per wave (no block barrier per gt).
      // signed-int order == float order for the values used here: -1.0f (negative int) < any
      // IoU >= 0 (non-negative ints, monotone in the float value)
      // (the six-step wave reduction only when some lane beats the maximum seen so far — a same-address LDS
      //  read is a broadcast and a racy LOWER bound of the running maximum, so skipping is exact)
      const float seen = __int_as_float(smax[g]);
      if (__ballot(v > seen)) {
        const float wm = bgs::wave_max(v);
        if (lane == 0) atomicMax(&smax[g], __float_as_int(wm));
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < cn; t += 256) {
      bm[c0 + t] = smax[t];
      if (smax[t] >= send_bits) atomicMax(&gt_max_bits[(size_t)n * gmax_stride + c0 + t], smax[t]);
    }
  }
  if (!live) return;
  if (box_max) box_max[(size_t)n * A + i] = best;
  int a = -1;
  if (ok) {
    if (best >= neg_lo && best < neg_hi) a = 0;
    if (best >= pos_thr) a = barg + 1;
  }
  assigned[(size_t)n * A + i] = a;
}

// pass 2: "every gt claims the boxes that attain its maximum" (later gts win).  A workgroup can only hold such
// a box for the gts whose workgroup maximum equals the global one: all the others (all but ~G of the 1050 per
// image) leave after comparing the two rows of maxima.
__global__ __launch_bounds__(256) void iou_assign_kernel(
    const float* __restrict__ boxes, long long box_img_stride, int box_stride,
    const uint8_t* __restrict__ valid,
    const float* __restrict__ gt, ImgTable T, int A, int gmax_stride,
    const int* __restrict__ blk_max, const int* __restrict__ gt_max_bits,
    float min_pos_iou, int* __restrict__ assigned) {
  __shared__ float sgt[kGtChunk][4];
  __shared__ float sgmax[kGtChunk];
  __shared__ int stie[kGtChunk];
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int g0 = T.gt_off[n], G = T.gt_off[n + 1] - g0;
  const int* bm = blk_max + ((size_t)n * gridDim.x + blockIdx.x) * gmax_stride;
  const bool live = i < A;
  const bool ok = live && (valid ? valid[(size_t)n * A + i] != 0 : true);
  float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
  bool have_box = false;
  int winner = 0;
  for (int c0 = 0; c0 < G; c0 += kGtChunk) {
    const int cn = min(kGtChunk, G - c0);
    int mine = 0;
    for (int t = threadIdx.x; t < cn; t += 256) {
      const int gm = gt_max_bits[(size_t)n * gmax_stride + c0 + t];
      const int tie = (gm == bm[c0 + t] && __int_as_float(gm) >= min_pos_iou) ? 1 : 0;
      stie[t] = tie;
      sgmax[t] = __int_as_float(gm);
      mine |= tie;
    }
    if (!__syncthreads_or(mine)) continue;       // (also orders the stie / sgmax writes)
    for (int t = threadIdx.x; t < cn * 4; t += 256) sgt[t >> 2][t & 3] = gt[(size_t)(g0 + c0) * 4 + t];
    if (!have_box && live) {
      const float* b = boxes + (size_t)n * box_img_stride + (size_t)i * box_stride;
      x1 = b[0]; y1 = b[1]; x2 = b[2]; y2 = b[3];
      area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
      have_box = true;
    }
    __syncthreads();
    if (ok) {
      for (int g = 0; g < cn; ++g) {
        if (!stie[g]) continue;
        const float v = iou1(x1, y1, x2, y2, area, sgt[g][0], sgt[g][1], sgt[g][2], sgt[g][3]);
        if (v == sgmax[g]) winner = c0 + g + 1;
      }
    }
    __syncthreads();
  }
  if (ok && winner > 0) assigned[(size_t)n * A + i] = winner;
}

__global__ void fill_i32_kernel(int* p, int v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    p[i] = v;
}

// ---------------------------------------------------------------------------------------------
struct LevelTable {
  const float* out[kMaxLevels];  // fused RPN head output of the level: [N, H*W, A + 4A]
  int start[kMaxLevels + 1];     // first global anchor index of the level; start[L] = A_total
  int hw[kMaxLevels];            // H*W of the level
  int num_levels;
  int num_anchors;               // A per location
};

struct Coding {
  float mean[4], stdv[4];
};

__device__ __forceinline__ void encode_delta(float px1, float py1, float px2, float py2, float gx1,
                                             float gy1, float gx2, float gy2, const Coding& c,
                                             float (&d)[4]) {
  const float px = (px1 + px2) * 0.5f, py = (py1 + py2) * 0.5f;
  const float pw = px2 - px1 + 1.f, ph = py2 - py1 + 1.f;
  const float gx = (gx1 + gx2) * 0.5f, gy = (gy1 + gy2) * 0.5f;
  const float gw = gx2 - gx1 + 1.f, gh = gy2 - gy1 + 1.f;
  d[0] = ((gx - px) / pw - c.mean[0]) / c.stdv[0];
  d[1] = ((gy - py) / ph - c.mean[1]) / c.stdv[1];
  d[2] = (logf(gw / pw) - c.mean[2]) / c.stdv[2];
  d[3] = (logf(gh / ph) - c.mean[3]) / c.stdv[3];
}

// partial [N][blocks][L][4] = {cls sum, bbox sum, n_pos, n_neg} of every block
__global__ __launch_bounds__(256) void rpn_loss_kernel(LevelTable Lv, ImgTable T,
                                                       const float* __restrict__ anchors,
                                                       const int* __restrict__ assigned,
                                                       const uint8_t* __restrict__ pos_mask,
                                                       const uint8_t* __restrict__ neg_mask,
                                                       const float* __restrict__ gt, Coding cod,
                                                       float beta, float pos_weight, int A,
                                                       float* __restrict__ partial) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int lvl = -1;
  if (i < A) {
    const bool ps = pos_mask[(size_t)n * A + i] != 0, ns = neg_mask[(size_t)n * A + i] != 0;
    if (ps || ns) {
      lvl = 0;
      while (lvl + 1 < Lv.num_levels && i >= Lv.start[lvl + 1]) ++lvl;
      const int p = i - Lv.start[lvl];
      const int na = Lv.num_anchors, ch = 5 * na;
      const int loc = p / na, a = p - loc * na;
      const float* o = Lv.out[lvl] + ((size_t)n * Lv.hw[lvl] + loc) * ch;
      const float x = o[a];
      const float t = ps ? 1.f : 0.f;
      const float w = ps ? pos_weight : 1.f;
      // F.binary_cross_entropy_with_logits: max(x,0) - x*t + log(1 + exp(-|x|))
      v[0] = w * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
      if (ps) {
        const int g = T.gt_off[n] + assigned[(size_t)n * A + i] - 1;
        const float* an = anchors + (size_t)i * 4;
        const float* gb = gt + (size_t)g * 4;
        float d[4];
        encode_delta(an[0], an[1], an[2], an[3], gb[0], gb[1], gb[2], gb[3], cod, d);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float df = fabsf(o[na + a * 4 + c] - d[c]);
          s += df < beta ? 0.5f * df * df / beta : df - 0.5f * beta;
        }
        v[1] = s;
        v[2] = 1.f;
      } else {
        v[3] = 1.f;
      }
    }
  }
  // deterministic block reduction, per level touched by this block
  __shared__ float red[4][4];
  const int first = blockIdx.x * 256, last = min(first + 255, A - 1);
  float* out = partial + (((size_t)n * gridDim.x + blockIdx.x) * Lv.num_levels) * 4;
  for (int l = 0; l < Lv.num_levels; ++l) {
    const bool touches = first < Lv.start[l + 1] && last >= Lv.start[l];   // block-uniform
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (touches) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float s = bgs::wave_sum(lvl == l ? v[q] : 0.f);
        if (lane == 0) red[wave][q] = s;
      }
      __syncthreads();
      if (threadIdx.x < 4) r[0] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                  red[3][threadIdx.x];
      __syncthreads();
    }
    if (threadIdx.x < 4) out[l * 4 + threadIdx.x] = r[0];
  }
}

// loss_cls[l], loss_bbox[l] = weight * sum over images / (sum_n max(n_pos,1) + max(n_neg,1))
__global__ __launch_bounds__(1024) void rpn_loss_finalize_kernel(const float* __restrict__ partial,
                                                                LevelTable Lv, int N, int blocks, int L,
                                                                float w_cls, float w_bbox,
                                                                float* __restrict__ loss_cls,
                                                                float* __restrict__ loss_bbox,
                                                                float* __restrict__ num_total_out) {
  __shared__ float tot[kMaxImgs][kMaxLevels][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // one wave per (image, level, quantity) triple, fixed summation order; 16 waves (the 40 triples
  // of cfg[1] on 4 waves were 33 us of dependent strided loads)
  // Only the blocks whose 256 anchors reach into level l hold anything but +0.0 for it (rpn_loss_kernel writes
  // exact zeros elsewhere), and x + 0.0 == x: every lane adds the same non-zero terms in the same order as a
  // walk over all blocks would — 13 + 4 + 1 + 1 + 1 trips for the five levels of cfg[1] instead of 5 x 17.
  const int jobs = N * L * 4;
  for (int j = wave; j < jobs; j += 16) {
    const int q = j & 3, l = (j >> 2) % L, n = (j >> 2) / L;
    const int b_lo = Lv.start[l] >> 8, b_hi = min(blocks - 1, (Lv.start[l + 1] - 1) >> 8);
    float s = 0.f;
    for (int b = (b_lo & ~63) + lane; b <= b_hi; b += 64)
      if (b >= b_lo) s += partial[(((size_t)n * blocks + b) * L + l) * 4 + q];
    s = bgs::wave_sum(s);
    if (lane == 0) tot[n][l][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float num = 0.f;
    for (int n = 0; n < N; ++n) {
      float np = 0.f, nn = 0.f;
      for (int l = 0; l < L; ++l) {
        np += tot[n][l][2];
        nn += tot[n][l][3];
      }
      num += fmaxf(np, 1.f) + fmaxf(nn, 1.f);   // anchor_target.py:62-63
    }
    for (int l = 0; l < L; ++l) {
      float c = 0.f, b = 0.f;
      for (int n = 0; n < N; ++n) {
        c += tot[n][l][0];
        b += tot[n][l][1];
      }
      loss_cls[l] = w_cls * c / num;
      loss_bbox[l] = w_bbox * b / num;
    }
    if (num_total_out) num_total_out[0] = num;
  }
}

// Gradient of the per-level RPN losses w.r.t. the fused head outputs (selectp = 0 path):
//   d cls logit = g_cls[l] * w_cls * w * (sigmoid(x) - t) / num      (sampled anchors)
//   d delta_c   = g_bbox[l] * w_bbox * sl1'(x_c - d_c) / num         (sampled positives)
// everything else 0 (the caller zero-fills dout).  One thread per anchor, distinct addresses.
struct GradTable {
  float* dout[kMaxLevels];
};

__global__ __launch_bounds__(256) void rpn_loss_grad_kernel(
    LevelTable Lv, GradTable G, ImgTable T, const float* __restrict__ anchors,
    const int* __restrict__ assigned, const uint8_t* __restrict__ pos_mask,
    const uint8_t* __restrict__ neg_mask, const float* __restrict__ gt, Coding cod, float beta,
    float pos_weight, int A, const float* __restrict__ num_total, float w_cls, float w_bbox,
    const float* __restrict__ g_cls, const float* __restrict__ g_bbox) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A) return;
  const bool ps = pos_mask[(size_t)n * A + i] != 0, ns = neg_mask[(size_t)n * A + i] != 0;
  if (!(ps || ns)) return;
  int lvl = 0;
  while (lvl + 1 < Lv.num_levels && i >= Lv.start[lvl + 1]) ++lvl;
  const int p = i - Lv.start[lvl];
  const int na = Lv.num_anchors, ch = 5 * na;
  const int loc = p / na, a = p - loc * na;
  const size_t row = ((size_t)n * Lv.hw[lvl] + loc) * ch;
  const float* o = Lv.out[lvl] + row;
  float* d_o = G.dout[lvl] + row;
  const float inv = 1.f / num_total[0];
  const float x = o[a];
  const float t = ps ? 1.f : 0.f;
  const float w = ps ? pos_weight : 1.f;
  const float sig = 1.f / (1.f + expf(-x));
  d_o[a] = g_cls[lvl] * w_cls * w * (sig - t) * inv;
  if (ps) {
    const int g = T.gt_off[n] + assigned[(size_t)n * A + i] - 1;
    const float* an = anchors + (size_t)i * 4;
    const float* gb = gt + (size_t)g * 4;
    float d[4];
    encode_delta(an[0], an[1], an[2], an[3], gb[0], gb[1], gb[2], gb[3], cod, d);
    const float sc = g_bbox[lvl] * w_bbox * inv;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float df = o[na + a * 4 + c] - d[c];
      const float ad = fabsf(df);
      const float gr = ad < beta ? df / beta : (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
      d_o[na + a * 4 + c] = sc * gr;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// boxes_out [N, L, nmax, 5] = decode(anchor[top_idx], delta) clamped to the image, score =
// sigmoid(top logit).  top_idx / top_logit: [N, L, nmax] (entries >= count[l] are ignored).
struct CountTable {
  int count[kMaxLevels];
};

__global__ __launch_bounds__(256) void decode_proposals_kernel(
    LevelTable Lv, ImgTable T, CountTable Ct, const float* __restrict__ anchors,
    const long long* __restrict__ top_idx, const float* __restrict__ top_logit, Coding cod,
    float max_ratio, int nmax, float* __restrict__ boxes_out) {
  const int n = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nmax) return;
  float* o = boxes_out + (((size_t)n * Lv.num_levels + l) * nmax + j) * 5;
  if (j >= Ct.count[l]) {
    o[0] = o[1] = o[2] = o[3] = o[4] = 0.f;
    return;
  }
  const size_t tj = ((size_t)n * Lv.num_levels + l) * nmax + j;
  long long p = top_idx[tj];
  const int na = Lv.num_anchors, ch = 5 * na;
  const long long cnt = (long long)Lv.hw[l] * na;
  p = p < 0 ? 0 : (p >= cnt ? cnt - 1 : p);
  const int loc = (int)(p / na), a = (int)(p - (long long)loc * na);
  const float* src = Lv.out[l] + ((size_t)n * Lv.hw[l] + loc) * ch + na + a * 4;
  const float* an = anchors + ((size_t)Lv.start[l] + p) * 4;
  const float dx = src[0] * cod.stdv[0] + cod.mean[0];
  const float dy = src[1] * cod.stdv[1] + cod.mean[1];
  float dw = src[2] * cod.stdv[2] + cod.mean[2];
  float dh = src[3] * cod.stdv[3] + cod.mean[3];
  dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
  dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
  const float px = (an[0] + an[2]) * 0.5f, py = (an[1] + an[3]) * 0.5f;
  const float pw = an[2] - an[0] + 1.f, ph = an[3] - an[1] + 1.f;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  const float wmax = (float)(T.img_w[n] - 1), hmax = (float)(T.img_h[n] - 1);
  o[0] = fminf(fmaxf(gx - gw * 0.5f + 0.5f, 0.f), wmax);
  o[1] = fminf(fmaxf(gy - gh * 0.5f + 0.5f, 0.f), hmax);
  o[2] = fminf(fmaxf(gx + gw * 0.5f - 0.5f, 0.f), wmax);
  o[3] = fminf(fmaxf(gy + gh * 0.5f - 0.5f, 0.f), hmax);
  o[4] = 1.f / (1.f + expf(-top_logit[tj]));
}

// Cascade stage hand-over in ONE launch (BBoxHead.refine_bboxes -> regress_by_class -> delta2bbox,
// mmdet/models/bbox_heads/bbox_head.py:169-239, mmdet/core/bbox/transforms.py:34-111): every RoI is
// re-regressed with the deltas of its own class (or the class-agnostic four) and clipped to its
// image.  The tensor-op form is ~35 element-wise launches per image and stage.  Same operation order
// as box_ops.delta2bbox with every product and sum rounded separately (fp contract off), so the
// boxes are those of the tensor form bit for bit wherever expf agrees.
__global__ __launch_bounds__(256) void refine_boxes_kernel(const float* __restrict__ rois,
                                                           const long long* __restrict__ labels,
                                                           const float* __restrict__ bbox_pred,
                                                           int K, int pred_cols, Coding cod,
                                                           ImgTable T, int n_img, float max_ratio,
                                                           float* __restrict__ out) {
#pragma clang fp contract(off)
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float* r = rois + (size_t)k * 5;
  int img = (int)r[0];
  img = img < 0 ? 0 : (img >= n_img ? n_img - 1 : img);
  long long c = 0;
  if (pred_cols > 4 && labels) {
    c = labels[k];
    const long long nc = pred_cols / 4;
    c = c < 0 ? 0 : (c >= nc ? nc - 1 : c);
  }
  const float* d = bbox_pred + (size_t)k * pred_cols + c * 4;
  const float dx = d[0] * cod.stdv[0] + cod.mean[0];
  const float dy = d[1] * cod.stdv[1] + cod.mean[1];
  float dw = d[2] * cod.stdv[2] + cod.mean[2];
  float dh = d[3] * cod.stdv[3] + cod.mean[3];
  dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
  dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
  const float pw = r[3] - r[1] + 1.0f, ph = r[4] - r[2] + 1.0f;
  const float px = (r[1] + r[3]) * 0.5f, py = (r[2] + r[4]) * 0.5f;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  const float wmax = (float)(T.img_w[img] - 1), hmax = (float)(T.img_h[img] - 1);
  float* o = out + (size_t)k * 4;
  o[0] = fminf(fmaxf(gx - gw * 0.5f + 0.5f, 0.f), wmax);
  o[1] = fminf(fmaxf(gy - gh * 0.5f + 0.5f, 0.f), hmax);
  o[2] = fminf(fmaxf(gx + gw * 0.5f - 0.5f, 0.f), wmax);
  o[3] = fminf(fmaxf(gy + gh * 0.5f - 0.5f, 0.f), hmax);
}

// ---------------------------------------------------------------------------------------------
struct PtrTable {
  const float* boxes[kMaxImgs];      // candidate boxes of the image [cand_n, >=4] (row stride below)
  const int* assigned[kMaxImgs];        // [cand_n] int32: -1 / 0 / gt index + 1
  const long long* inds[kMaxImgs];      // [num] sampled candidate indices
  const uint8_t* valid[kMaxImgs];       // [num] or null
  const long long* gt_labels[kMaxImgs];  // [G_n]
  int box_stride[kMaxImgs];
};

__global__ __launch_bounds__(256) void rcnn_targets_kernel(PtrTable P, ImgTable T,
                                                           const float* __restrict__ gt, Coding cod,
                                                           int num, float pos_weight,
                                                           float* __restrict__ rois,
                                                           long long* __restrict__ labels,
                                                           float* __restrict__ label_weights,
                                                           float* __restrict__ bbox_targets,
                                                           float* __restrict__ bbox_weights) {
  const int n = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= num) return;
  const size_t r = (size_t)n * num + j;
  const long long ci = P.inds[n][j];
  const float* b = P.boxes[n] + (size_t)ci * P.box_stride[n];
  const int a = P.assigned[n][ci];
  const bool ok = P.valid[n] ? P.valid[n][j] != 0 : true;
  const bool pos = ok && a > 0;
  rois[r * 5 + 0] = (float)n;
  rois[r * 5 + 1] = b[0];
  rois[r * 5 + 2] = b[1];
  rois[r * 5 + 3] = b[2];
  rois[r * 5 + 4] = b[3];
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  long long lab = 0;
  if (pos) {
    const int g = T.gt_off[n] + (int)a - 1;
    const float* gb = gt + (size_t)g * 4;
    encode_delta(b[0], b[1], b[2], b[3], gb[0], gb[1], gb[2], gb[3], cod, d);
    lab = P.gt_labels[n][a - 1];
  }
  labels[r] = lab;
  label_weights[r] = pos ? pos_weight : (ok ? 1.f : 0.f);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bbox_targets[r * 4 + c] = d[c];
    bbox_weights[r * 4 + c] = pos ? 1.f : 0.f;
  }
}

int fill_img_table(ImgTable* T, const int* host_gt_offsets, const int* host_img_hw, int N) {
  if (N <= 0 || N > kMaxImgs || !host_gt_offsets) return BGS_ERR_INVALID_ARG;
  for (int i = 0; i <= kMaxImgs; ++i) T->gt_off[i] = 0;
  for (int i = 0; i <= N; ++i) T->gt_off[i] = host_gt_offsets[i];
  for (int i = 0; i < kMaxImgs; ++i) {
    T->img_h[i] = host_img_hw ? host_img_hw[2 * (i < N ? i : 0)] : 1;
    T->img_w[i] = host_img_hw ? host_img_hw[2 * (i < N ? i : 0) + 1] : 1;
  }
  for (int i = 0; i < N; ++i)
    if (T->gt_off[i + 1] < T->gt_off[i]) return BGS_ERR_INVALID_ARG;
  return BGS_OK;
}

int fill_level_table(LevelTable* Lv, const float* const* host_outs, const int* host_hw, int L,
                     int num_anchors) {
  if (L <= 0 || L > kMaxLevels || !host_outs || !host_hw || num_anchors <= 0)
    return BGS_ERR_INVALID_ARG;
  int start = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    Lv->out[l] = l < L ? host_outs[l] : nullptr;
    Lv->hw[l] = l < L ? host_hw[l] : 0;
    Lv->start[l] = start;
    if (l < L) {
      if (!host_outs[l] || host_hw[l] <= 0) return BGS_ERR_INVALID_ARG;
      start += host_hw[l] * num_anchors;
    }
  }
  Lv->start[kMaxLevels] = start;
  for (int l = L; l <= kMaxLevels; ++l) Lv->start[l] = start;
  Lv->num_levels = L;
  Lv->num_anchors = num_anchors;
  return BGS_OK;
}

}  // namespace

extern "C" size_t bgs_iou_assign_workspace_bytes(int N, int A, int G_total) {
  if (N <= 0 || A <= 0 || G_total < 0) return 0;
  // box_max [N,A] f32 + gt_max [N, G_total] i32 + per-workgroup gt maxima [N, ceil(A / 256), G_total] i32
  const size_t g = (size_t)(G_total > 0 ? G_total : 1);
  return (size_t)N * A * 4 + (size_t)N * g * 4 + (size_t)N * ((A + 255) / 256) * g * 4;
}

extern "C" int bgs_iou_assign(const float* boxes, long long box_img_stride, int box_stride,
                              const uint8_t* valid,
                              const float* gt, const int* host_gt_offsets, int N, int A,
                              float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
                              float min_pos_iou, int* assigned, float* max_overlaps_out,
                              void* workspace, bgs_stream_t stream) {
  if (N <= 0 || A <= 0 || !boxes || !gt || !assigned || !workspace || box_stride < 4)
    return BGS_ERR_INVALID_ARG;
  ImgTable T;
  const int rc = fill_img_table(&T, host_gt_offsets, nullptr, N);
  if (rc != BGS_OK) return rc;
  const int Gt = T.gt_off[N];
  if (Gt <= 0) return BGS_ERR_INVALID_ARG;   // the reference raises ValueError('No gt or bboxes')
  hipStream_t st = (hipStream_t)stream;
  float* box_max = max_overlaps_out;          // only materialised when the caller wants it
  int* gt_max = (int*)((char*)workspace + (size_t)N * A * 4);
  int* blk_max = gt_max + (size_t)N * Gt;
  const size_t ng = (size_t)N * Gt;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, gt_max,
                     (int)0xBF800000 /* -1.0f */, ng);
  // gt_max rows are indexed by the gt's position inside its image: row stride = Gt is enough
  dim3 grid((unsigned)((A + 255) / 256), (unsigned)N);
  // maxima below min_pos_iou never matter to pass 2 (IoUs are >= 0: positive floats order like their bits)
  const float send = min_pos_iou > 0.f ? min_pos_iou : 0.f;
  int send_bits;
  memcpy(&send_bits, &send, sizeof(int));
  hipLaunchKernelGGL(iou_gtmax_kernel, grid, dim3(256), 0, st, boxes, box_img_stride, box_stride,
                     valid, gt, T, A, Gt, pos_iou_thr, neg_iou_lo, neg_iou_hi, send_bits, box_max, blk_max,
                     gt_max, assigned);
  hipLaunchKernelGGL(iou_assign_kernel, grid, dim3(256), 0, st, boxes, box_img_stride, box_stride,
                     valid, gt, T, A, Gt, blk_max, gt_max, min_pos_iou, assigned);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" size_t bgs_rpn_loss_workspace_bytes(int N, int A_total, int L) {
  if (N <= 0 || A_total <= 0 || L <= 0) return 0;
  return (size_t)N * ((A_total + 255) / 256) * L * 4 * sizeof(float);
}

extern "C" int bgs_rpn_loss(const float* const* host_level_outs, const int* host_level_hw, int L,
                            int num_anchors, const float* anchors, const int* assigned,
                            const uint8_t* pos_mask, const uint8_t* neg_mask, const float* gt,
                            const int* host_gt_offsets, int N, const float* host_means,
                            const float* host_stds, float beta, float pos_weight,
                            float loss_weight_cls, float loss_weight_bbox, float* loss_cls_out,
                            float* loss_bbox_out, float* num_total_out, void* workspace,
                            bgs_stream_t stream) {
  if (!anchors || !assigned || !pos_mask || !neg_mask || !gt || !loss_cls_out || !loss_bbox_out ||
      !workspace || !host_means || !host_stds || !(beta > 0.f))
    return BGS_ERR_INVALID_ARG;
  LevelTable Lv;
  int rc = fill_level_table(&Lv, host_level_outs, host_level_hw, L, num_anchors);
  if (rc != BGS_OK) return rc;
  ImgTable T;
  rc = fill_img_table(&T, host_gt_offsets, nullptr, N);
  if (rc != BGS_OK) return rc;
  Coding cod;
  for (int c = 0; c < 4; ++c) {
    cod.mean[c] = host_means[c];
    cod.stdv[c] = host_stds[c];
  }
  const int A = Lv.start[kMaxLevels];
  const int blocks = (A + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(blocks, N), dim3(256), 0, st, Lv, T, anchors, assigned,
                     pos_mask, neg_mask, gt, cod, beta, pos_weight <= 0.f ? 1.f : pos_weight, A,
                     (float*)workspace);
  hipLaunchKernelGGL(rpn_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)workspace,
                     Lv, N, blocks, L, loss_weight_cls, loss_weight_bbox, loss_cls_out, loss_bbox_out,
                     num_total_out);
  BGS_RETURN_LAUNCH_STATUS();
}

// Counter-based 62-bit sampling keys: out[i] = splitmix64(seed, draw, i) >> 2.  `draw` is a device
// counter the caller bumps with a tensor op per call, so the kernel arguments stay constant under
// hipGraph replay while every replay draws fresh keys (RandomSampler's shuffle,
// mmdet/core/bbox/samplers/random_sampler.py:19-33, as a top-k over random keys: assign.py).
__global__ __launch_bounds__(256) void random_keys_kernel(uint64_t seed,
                                                          const long long* __restrict__ draw, int n,
                                                          long long* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t d = draw ? (uint64_t)draw[0] : 0ull;
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)i + 1ull) + 0xD1B54A32D192ED03ull * (d + 1ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  out[i] = (long long)(x >> 2);
}

extern "C" int bgs_random_keys(uint64_t seed, const long long* draw_counter, int n, long long* out,
                               bgs_stream_t stream) {
  if (n < 0) return BGS_ERR_INVALID_ARG;
  if (n == 0) return BGS_OK;
  if (!out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(random_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     seed, draw_counter, n, out);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_rpn_loss_grad(const float* const* host_level_outs,
                                 float* const* host_level_douts, const int* host_level_hw, int L,
                                 int num_anchors, const float* anchors, const int* assigned,
                                 const uint8_t* pos_mask, const uint8_t* neg_mask, const float* gt,
                                 const int* host_gt_offsets, int N, const float* host_means,
                                 const float* host_stds, float beta, float pos_weight,
                                 float loss_weight_cls, float loss_weight_bbox,
                                 const float* num_total, const float* grad_loss_cls,
                                 const float* grad_loss_bbox, bgs_stream_t stream) {
  if (!anchors || !assigned || !pos_mask || !neg_mask || !gt || !num_total || !grad_loss_cls ||
      !grad_loss_bbox || !host_level_douts || !host_means || !host_stds || !(beta > 0.f))
    return BGS_ERR_INVALID_ARG;
  LevelTable Lv;
  int rc = fill_level_table(&Lv, host_level_outs, host_level_hw, L, num_anchors);
  if (rc != BGS_OK) return rc;
  GradTable G;
  for (int l = 0; l < kMaxLevels; ++l) G.dout[l] = nullptr;
  for (int l = 0; l < L; ++l) {
    if (!host_level_douts[l]) return BGS_ERR_INVALID_ARG;
    G.dout[l] = host_level_douts[l];
  }
  ImgTable T;
  rc = fill_img_table(&T, host_gt_offsets, nullptr, N);
  if (rc != BGS_OK) return rc;
  Coding cod;
  for (int c = 0; c < 4; ++c) {
    cod.mean[c] = host_means[c];
    cod.stdv[c] = host_stds[c];
  }
  const int A = Lv.start[kMaxLevels];
  const int blocks = (A + 255) / 256;
  hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, Lv,
                     G, T, anchors, assigned, pos_mask, neg_mask, gt, cod, beta,
                     pos_weight <= 0.f ? 1.f : pos_weight, A, num_total, loss_weight_cls,
                     loss_weight_bbox, grad_loss_cls, grad_loss_bbox);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_decode_proposals(const float* const* host_level_outs, const int* host_level_hw,
                                    const int* host_level_counts, int L, int num_anchors,
                                    const float* anchors, const long long* top_idx,
                                    const float* top_logit, int N, const int* host_img_hw,
                                    const float* host_means, const float* host_stds,
                                    float wh_ratio_clip, int nmax, float* boxes_out,
                                    bgs_stream_t stream) {
  if (!anchors || !top_idx || !top_logit || !boxes_out || !host_img_hw || !host_means ||
      !host_stds || !host_level_counts || nmax <= 0 || !(wh_ratio_clip > 0.f))
    return BGS_ERR_INVALID_ARG;
  LevelTable Lv;
  int rc = fill_level_table(&Lv, host_level_outs, host_level_hw, L, num_anchors);
  if (rc != BGS_OK) return rc;
  ImgTable T;
  int zero_off[kMaxImgs + 1] = {0};
  rc = fill_img_table(&T, zero_off, host_img_hw, N);
  if (rc != BGS_OK) return rc;
  Coding cod;
  CountTable Ct;
  for (int c = 0; c < 4; ++c) {
    cod.mean[c] = host_means[c];
    cod.stdv[c] = host_stds[c];
  }
  for (int l = 0; l < kMaxLevels; ++l) Ct.count[l] = l < L ? host_level_counts[l] : 0;
  hipLaunchKernelGGL(decode_proposals_kernel, dim3((nmax + 255) / 256, L, N), dim3(256), 0,
                     (hipStream_t)stream, Lv, T, Ct, anchors, top_idx, top_logit, cod,
                     fabsf(logf(wh_ratio_clip)), nmax, boxes_out);
  BGS_RETURN_LAUNCH_STATUS();
}

// rois [K,5] (image index, x1, y1, x2, y2), labels [K] int64 or NULL (class-agnostic), bbox_pred
// [K, pred_cols] (pred_cols = 4 or 4 * classes) -> out [K,4].  host_img_hw = {h0, w0, h1, w1, ..}.
extern "C" int bgs_refine_boxes(const float* rois, const long long* labels, const float* bbox_pred,
                                int K, int pred_cols, const int* host_img_hw, int N,
                                const float* host_means, const float* host_stds, float wh_ratio_clip,
                                float* out, bgs_stream_t stream) {
  if (K < 0 || N <= 0 || N > kMaxImgs || pred_cols < 4 || pred_cols % 4 != 0) return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!rois || !bbox_pred || !out || !host_img_hw || !host_means || !host_stds || wh_ratio_clip <= 0.f)
    return BGS_ERR_INVALID_ARG;
  if (pred_cols > 4 && !labels) return BGS_ERR_INVALID_ARG;
  ImgTable T;
  for (int i = 0; i < kMaxImgs; ++i) {
    const int s = i < N ? i : 0;
    T.gt_off[i] = 0;
    T.img_h[i] = host_img_hw[2 * s];
    T.img_w[i] = host_img_hw[2 * s + 1];
  }
  T.gt_off[kMaxImgs] = 0;
  Coding cod;
  for (int i = 0; i < 4; ++i) {
    cod.mean[i] = host_means[i];
    cod.stdv[i] = host_stds[i];
  }
  hipLaunchKernelGGL(refine_boxes_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, rois,
                     labels, bbox_pred, K, pred_cols, cod, T, N, fabsf(logf(wh_ratio_clip)), out);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_rcnn_targets(const float* const* host_boxes, const int* host_box_strides,
                                const int* const* host_assigned,
                                const long long* const* host_inds,
                                const uint8_t* const* host_valid,
                                const long long* const* host_gt_labels, const float* gt,
                                const int* host_gt_offsets, int N, int num, const float* host_means,
                                const float* host_stds, float pos_weight, float* rois,
                                long long* labels, float* label_weights, float* bbox_targets,
                                float* bbox_weights, bgs_stream_t stream) {
  if (!host_boxes || !host_box_strides || !host_assigned || !host_inds || !host_gt_labels || !gt ||
      !rois || !labels || !label_weights || !bbox_targets || !bbox_weights || num <= 0 ||
      !host_means || !host_stds)
    return BGS_ERR_INVALID_ARG;
  ImgTable T;
  const int rc = fill_img_table(&T, host_gt_offsets, nullptr, N);
  if (rc != BGS_OK) return rc;
  PtrTable P;
  for (int i = 0; i < kMaxImgs; ++i) {
    const int s = i < N ? i : 0;
    P.boxes[i] = host_boxes[s];
    P.assigned[i] = host_assigned[s];
    P.inds[i] = host_inds[s];
    P.valid[i] = host_valid ? host_valid[s] : nullptr;
    P.gt_labels[i] = host_gt_labels[s];
    P.box_stride[i] = host_box_strides[s];
    if (i < N && (!P.boxes[i] || !P.assigned[i] || !P.inds[i] || !P.gt_labels[i] ||
                  P.box_stride[i] < 4))
      return BGS_ERR_INVALID_ARG;
  }
  Coding cod;
  for (int c = 0; c < 4; ++c) {
    cod.mean[c] = host_means[c];
    cod.stdv[c] = host_stds[c];
  }
  hipLaunchKernelGGL(rcnn_targets_kernel, dim3((num + 255) / 256, N), dim3(256), 0,
                     (hipStream_t)stream, P, T, gt, cod, num, pos_weight <= 0.f ? 1.f : pos_weight,
                     rois, labels, label_weights, bbox_targets, bbox_weights);
  BGS_RETURN_LAUNCH_STATUS();
}
