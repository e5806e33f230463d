// Mask branch of the BAGS Mask R-CNN (cfg 4 = gs_mask_rcnn_r50_fpn_1x_lvis):
//
//   bgs_mask_target     mask_target_single (mmdet/core/mask/mask_target.py:16-38): crop the
//                       assigned GT bitmap with the integer-truncated positive proposal and resize it
//                       to mask_size x mask_size with mmcv.imresize == cv2.resize(INTER_LINEAR) on
//                       uint8.  The reference does this per RoI on the HOST (D2H of the proposals,
//                       a Python loop, OpenCV, H2D of the targets); here: one workgroup per RoI on
//                       bitmaps that stay in HBM.
//   bgs_mask_gt_logits  FCNMaskHead.conv_logits (mmdet/models/mask_heads/fcn_mask_head.py:82,101)
//                       evaluated ONLY for the channel the loss reads, pred[i, label_i]
//                       (mask_cross_entropy, mmdet/models/losses/cross_entropy_loss.py:54-61): the
//                       reference computes all 1231 channels (247 MMAC/RoI, a 988 MB tensor for 256
//                       RoIs) and then gathers one of them.
//   bgs_mask_bce        mask_cross_entropy fused with that single-channel 1x1 conv and its
//                       backward (d features, d conv_logits.weight rows, d bias).
//
// cv2.resize(INTER_LINEAR) for 8-bit images is fixed-point (modules/imgproc/src/resize.cpp:
// INTER_RESIZE_COEF_BITS = 11; horizontal pass keeps 11 fractional bits, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2).  OpenCV is not available in this image, so
// this restatement is "parity unpinned" (DESIGN.md §6); oracle/mask_oracle.py states the same
// arithmetic in numpy and the two are tested bit-exactly against each other.
#include <math.h>

#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxImgs = 16;

struct MaskTable {
  const uint8_t* masks[kMaxImgs];   // [G_n, Hm, Wm] uint8 bitmaps of image n
  int num_gt[kMaxImgs];
};

// cv2's cvRound on the default (round-half-to-even) mode, then saturate to short
__device__ __forceinline__ int coef(float v) {
  int r = (int)rintf(v * 2048.f);
  return min(max(r, -32768), 32767);
}

// one workgroup per RoI, one thread per output pixel (strided)
__global__ __launch_bounds__(256) void mask_target_kernel(MaskTable T, int Hm, int Wm,
                                                          const float* __restrict__ rois,
                                                          int roi_stride,
                                                          const int* __restrict__ gt_inds,
                                                          const uint8_t* __restrict__ valid,
                                                          int S, float* __restrict__ out) {
  const int p = blockIdx.x;
  float* o = out + (size_t)p * S * S;
  const float* r = rois + (size_t)p * roi_stride;
  const int n = (int)r[0];
  const int g = gt_inds[p];
  const bool ok = (!valid || valid[p]) && n >= 0 && n < kMaxImgs && g >= 0 && g < T.num_gt[n];
  if (!ok) {
    for (int i = threadIdx.x; i < S * S; i += 256) o[i] = 0.f;
    return;
  }
  // bbox = proposals_np[i, :].astype(np.int32): truncation toward zero
  const int x1 = (int)r[1], y1 = (int)r[2], x2 = (int)r[3], y2 = (int)r[4];
  int w = max(x2 - x1 + 1, 1), h = max(y2 - y1 + 1, 1);
  // numpy slicing gt_mask[y1:y1+h, x1:x1+w] truncates at the bitmap border
  const int cx0 = min(max(x1, 0), Wm), cy0 = min(max(y1, 0), Hm);
  w = min(x1 + w, Wm) - cx0;
  h = min(y1 + h, Hm) - cy0;
  if (w <= 0 || h <= 0) {     // empty crop: cv2 would raise; the reference never gets here
    for (int i = threadIdx.x; i < S * S; i += 256) o[i] = 0.f;
    return;
  }
  const uint8_t* src = T.masks[n] + ((size_t)g * Hm + cy0) * Wm + cx0;
  const double scale_x = (double)w / S, scale_y = (double)h / S;
  for (int i = threadIdx.x; i < S * S; i += 256) {
    const int dy = i / S, dx = i - dy * S;
    int v;
    if (w == S && h == S) {
      v = src[(size_t)dy * Wm + dx] ? 1 : 0;   // same size: cv2 copies
    } else {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) {
        fx = 0.f;
        sx = 0;
      }
      if (sx >= w - 1) {
        fx = 0.f;
        sx = w - 1;
      }
      const int a0 = coef(1.f - fx), a1 = coef(fx);
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      const int sy = (int)floorf(fy);
      fy -= sy;
      const int b0 = coef(1.f - fy), b1 = coef(fy);
      const int r0 = min(max(sy, 0), h - 1), r1 = min(max(sy + 1, 0), h - 1);
      const int sx1 = min(sx + 1, w - 1);
      const uint8_t* p0 = src + (size_t)r0 * Wm;
      const uint8_t* p1 = src + (size_t)r1 * Wm;
      const int S0 = (p0[sx] ? 1 : 0) * a0 + (p0[sx1] ? 1 : 0) * a1;
      const int S1 = (p1[sx] ? 1 : 0) * a0 + (p1[sx1] ? 1 : 0) * a1;
      v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    }
    o[i] = (float)v;
  }
}

// logit[p, pix] = <feat[p, pix, :], W[label_p, :]> + b[label_p]; one wave per pixel.
// MODE 0: write the logits.  MODE 1: BCE-with-logits against target, partial sums per
// workgroup, optional gradients (dfeat dense; dW / db rows through fp32 atomics: RoIs of the same
// class share a row).
template <int MODE>
__global__ __launch_bounds__(256) void mask_gt_channel_kernel(
    const float* __restrict__ feat, const float* __restrict__ W, const float* __restrict__ bias,
    const long long* __restrict__ labels, const float* __restrict__ target,
    const uint8_t* __restrict__ valid, const float* __restrict__ norm, int P, int pix, int C,
    int K, int chunks, float* __restrict__ out, float* __restrict__ dfeat, float* __restrict__ dW,
    float* __restrict__ db) {
  const int p = blockIdx.x / chunks, chunk = blockIdx.x - p * chunks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long lab = labels[p];
  lab = lab < 0 ? 0 : (lab >= K ? K - 1 : lab);
  const bool ok = !valid || valid[p] != 0;
  const int per = (pix + chunks - 1) / chunks;
  const int i0 = chunk * per, i1 = min(pix, i0 + per);
  // C == 256: one 16-byte quad per lane (C % 256 == 0 handled by the loop)
  const float* wrow = W + (size_t)lab * C;
  const float b = bias ? bias[lab] : 0.f;
  float lsum = 0.f, gsum = 0.f;
  const float scale = MODE == 1 ? norm[0] : 0.f;
  f32x4 dw = {0.f, 0.f, 0.f, 0.f};     // only C == 256 keeps the row gradient in registers
  for (int i = i0 + wave; i < i1; i += 4) {
    const float* x = feat + ((size_t)p * pix + i) * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + c);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + c);
      acc += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
    }
    const float z = bgs::wave_sum(acc) + b;
    if (MODE == 0) {
      if (lane == 0) out[(size_t)p * pix + i] = z;
      continue;
    }
    const float t = target[(size_t)p * pix + i];
    if (ok) lsum += fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
    if (dfeat || dW) {
      const float g = ok ? (1.f / (1.f + expf(-z)) - t) * scale : 0.f;
      gsum += g;
      for (int c = lane * 4; c < C; c += 256) {
        if (dfeat) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + c);
          *reinterpret_cast<f32x4*>(dfeat + ((size_t)p * pix + i) * C + c) = g * wv;
        }
        if (dW) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + c);
          if (C == 256) {
            dw += g * xv;
          } else {
            float* d = dW + (size_t)lab * C + c;
            unsafeAtomicAdd(d + 0, g * xv[0]);
            unsafeAtomicAdd(d + 1, g * xv[1]);
            unsafeAtomicAdd(d + 2, g * xv[2]);
            unsafeAtomicAdd(d + 3, g * xv[3]);
          }
        }
      }
    }
  }
  if (MODE == 0) return;
  __shared__ float red[4];
  __shared__ f32x4 redw[4][64];
  if (lane == 0) red[wave] = lsum;
  if (dW && C == 256) redw[wave][lane] = dw;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  if (dW && ok) {
    if (C == 256 && wave == 0) {
      const f32x4 s = (redw[0][lane] + redw[1][lane]) + (redw[2][lane] + redw[3][lane]);
      float* d = dW + (size_t)lab * C + lane * 4;
      unsafeAtomicAdd(d + 0, s[0]);
      unsafeAtomicAdd(d + 1, s[1]);
      unsafeAtomicAdd(d + 2, s[2]);
      unsafeAtomicAdd(d + 3, s[3]);
    }
    if (db && lane == 0) unsafeAtomicAdd(db + lab, gsum);   // gsum is wave-uniform
  }
}


// ---- test-time mask paste: FCNMaskHead.get_seg_masks (mmdet/models/mask_heads/fcn_mask_head.py:125-181) without the
// RLE step.  Per detection k: bbox = (int32)(box / scale_factor) (truncation, :164), w = max(x2 - x1 + 1, 1),
// h likewise; bbox_mask = mmcv.imresize(prob [S, S] float32, (w, h)) = cv2.resize(..., INTER_LINEAR) on float32
// (OpenCV resize.cpp, float path: src coordinate fx = (float)((dx + 0.5) * scale - 0.5) with scale = 1 / (w / S) in
// double, sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= S - 1 -> (S - 1, 0); rows: sy = floor(fy), the two
// source rows clamped to [0, S - 1] with fy kept; horizontal pass first (S[sx] * (1 - fx) + S[sx + 1] * fx, or S[sx]
// alone where sx + 1 leaves the row), then vertical (row0 * (1 - fy) + row1 * fy), all in float32, each product and sum
// rounded separately); (bbox_mask > thr) as uint8 goes to im_mask[y1 : y1 + h, x1 : x1 + w] of a zero [img_h, img_w]
// image.  Here the whole uint8 [K, img_h, img_w] tensor is produced in one launch — a thread owns one aligned 4-byte
// word of it (zeros outside the boxes: no separate fill) — and the part of a box that leaves the image is clipped
// (numpy's slice assignment raises there; boxes are clipped to the image upstream, bbox_head.py:136-139).
__device__ __forceinline__ void paste_axis(int d, double scale, int S, int& s0, int& s1, float& f, bool rows) {
  float fx = (float)(((double)d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (rows) {                                   // (resizeGeneric_Invoker: row indices clipped, weight kept)
    s0 = min(max(sx, 0), S - 1);
    s1 = min(max(sx + 1, 0), S - 1);
    f = fx;
    return;
  }
  if (sx < 0) {
    fx = 0.f;
    sx = 0;
  }
  if (sx >= S - 1) {
    fx = 0.f;
    sx = S - 1;
  }
  s0 = sx;
  s1 = sx + 1 < S ? sx + 1 : -1;                // -1: the "D[dx] = S[sx] * ONE" tail of HResizeLinear
  f = fx;
}

__global__ __launch_bounds__(256) void mask_paste_kernel(const float* __restrict__ probs,
                                                         const float* __restrict__ boxes, int box_stride,
                                                         int K, int S, float scale_factor, float thr, int img_h,
                                                         int img_w, unsigned* __restrict__ out_words,
                                                         long long total_bytes) {
  const long long word = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b0 = word * 4;
  if (b0 >= total_bytes) return;
  const long long plane = (long long)img_h * img_w;
  unsigned packed = 0;
  int k = -1, x1 = 0, y1 = 0, w = 1, h = 1;
  double sx_scale = 1.0, sy_scale = 1.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long long b = b0 + t;
    if (b >= total_bytes) break;
    const int kk = (int)(b / plane);
    const int rem = (int)(b - (long long)kk * plane);
    const int y = rem / img_w, x = rem - y * img_w;
    if (kk != k) {                               // (a word crosses a detection at most once)
      k = kk;
      const float* bx = boxes + (size_t)k * box_stride;
      x1 = (int)(bx[0] / scale_factor);
      y1 = (int)(bx[1] / scale_factor);
      const int x2 = (int)(bx[2] / scale_factor), y2 = (int)(bx[3] / scale_factor);
      w = max(x2 - x1 + 1, 1);
      h = max(y2 - y1 + 1, 1);
      sx_scale = 1.0 / ((double)w / (double)S);
      sy_scale = 1.0 / ((double)h / (double)S);
    }
    const int dx = x - x1, dy = y - y1;
    if (dx < 0 || dx >= w || dy < 0 || dy >= h) continue;
    const float* pm = probs + (size_t)k * S * S;
    float v;
    if (w == S && h == S) {                      // cv2.resize returns the source when the size is unchanged
      v = pm[dy * S + dx];
    } else {
      int c0, c1, r0, r1;
      float fx, fy;
      paste_axis(dx, sx_scale, S, c0, c1, fx, false);
      paste_axis(dy, sy_scale, S, r0, r1, fy, true);
      const float a0 = 1.f - fx, a1 = fx, bt0 = 1.f - fy, bt1 = fy;
      float h0, h1;
      if (c1 >= 0) {
        h0 = __fadd_rn(__fmul_rn(pm[r0 * S + c0], a0), __fmul_rn(pm[r0 * S + c1], a1));
        h1 = __fadd_rn(__fmul_rn(pm[r1 * S + c0], a0), __fmul_rn(pm[r1 * S + c1], a1));
      } else {
        h0 = pm[r0 * S + c0];
        h1 = pm[r1 * S + c0];
      }
      v = __fadd_rn(__fmul_rn(h0, bt0), __fmul_rn(h1, bt1));
    }
    if (v > thr) packed |= 1u << (8 * t);
  }
  if (b0 + 4 <= total_bytes) {
    out_words[word] = packed;
  } else {                                       // the last, partial word
    unsigned char* ob = reinterpret_cast<unsigned char*>(out_words) + b0;
    for (int t = 0; b0 + t < total_bytes; ++t) ob[t] = (unsigned char)((packed >> (8 * t)) & 0xffu);
  }
}
}  // namespace

extern "C" int bgs_mask_target(const uint8_t* const* host_masks, const int* host_num_gt,
                               int num_images, int mask_h, int mask_w, const float* rois,
                               int roi_stride, const int* gt_inds, const uint8_t* valid, int P,
                               int mask_size, float* out, bgs_stream_t stream) {
  if (num_images <= 0 || num_images > kMaxImgs || !host_masks || !host_num_gt || mask_h <= 0 ||
      mask_w <= 0 || P < 0 || mask_size <= 0 || roi_stride < 5)
    return BGS_ERR_INVALID_ARG;
  if (P == 0) return BGS_OK;
  if (!rois || !gt_inds || !out) return BGS_ERR_INVALID_ARG;
  MaskTable T;
  for (int i = 0; i < kMaxImgs; ++i) {
    T.masks[i] = nullptr;
    T.num_gt[i] = 0;
  }
  for (int i = 0; i < num_images; ++i) {
    if (host_num_gt[i] < 0 || (host_num_gt[i] > 0 && !host_masks[i])) return BGS_ERR_INVALID_ARG;
    T.masks[i] = host_masks[i];
    T.num_gt[i] = host_num_gt[i];
  }
  hipLaunchKernelGGL(mask_target_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, T, mask_h,
                     mask_w, rois, roi_stride, gt_inds, valid, mask_size, out);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_mask_paste_u8(const float* probs, const float* boxes, int box_stride, int K, int S,
                                 float scale_factor, float thr, int img_h, int img_w, unsigned char* out,
                                 bgs_stream_t stream) {
  if (K < 0 || S <= 0 || img_h <= 0 || img_w <= 0 || box_stride < 4 || !(scale_factor > 0.f))
    return BGS_ERR_INVALID_ARG;
  if (K == 0) return BGS_OK;
  if (!probs || !boxes || !out) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)out % 4 != 0) return BGS_ERR_UNSUPPORTED;
  const long long total = (long long)K * img_h * img_w;
  const long long words = (total + 3) / 4;
  if (words > 0x7fffffffLL * 256LL || (long long)img_h * img_w > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mask_paste_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     probs, boxes, box_stride, K, S, scale_factor, thr, img_h, img_w,
                     reinterpret_cast<unsigned*>(out), total);
  BGS_RETURN_LAUNCH_STATUS();
}

static int mask_chunks(int P) {
  // >= ~1024 workgroups in flight for the usual P = 256
  int c = 1;
  while (P * c < 1024 && c < 8) c *= 2;
  return c;
}

extern "C" int bgs_mask_gt_logits(const float* feat, const float* weight, const float* bias,
                                  const long long* labels, int P, int pixels, int C,
                                  int num_classes, float* logits_out, bgs_stream_t stream) {
  if (P < 0 || pixels <= 0 || C <= 0 || num_classes <= 0) return BGS_ERR_INVALID_ARG;
  if (P == 0) return BGS_OK;
  if (!feat || !weight || !labels || !logits_out) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || ((uintptr_t)feat | (uintptr_t)weight) % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int chunks = mask_chunks(P);
  hipLaunchKernelGGL((mask_gt_channel_kernel<0>), dim3(P * chunks), dim3(256), 0,
                     (hipStream_t)stream, feat, weight, bias, labels, nullptr, nullptr, nullptr, P,
                     pixels, C, num_classes, chunks, logits_out, nullptr, nullptr, nullptr);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_mask_bce_partials(int P) { return P * mask_chunks(P); }

extern "C" int bgs_mask_bce(const float* feat, const float* weight, const float* bias,
                            const long long* labels, const float* target, const uint8_t* valid,
                            const float* norm, int P, int pixels, int C, int num_classes,
                            float* partial_out, float* dfeat, float* dweight, float* dbias,
                            bgs_stream_t stream) {
  if (P < 0 || pixels <= 0 || C <= 0 || num_classes <= 0) return BGS_ERR_INVALID_ARG;
  if (P == 0) return BGS_OK;
  if (!feat || !weight || !labels || !target || !norm || !partial_out) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || ((uintptr_t)feat | (uintptr_t)weight | (uintptr_t)dfeat) % 16 != 0)
    return BGS_ERR_UNSUPPORTED;
  const int chunks = mask_chunks(P);
  hipLaunchKernelGGL((mask_gt_channel_kernel<1>), dim3(P * chunks), dim3(256), 0,
                     (hipStream_t)stream, feat, weight, bias, labels, target, valid, norm, P, pixels,
                     C, num_classes, chunks, partial_out, dfeat, dweight, dbias);
  BGS_RETURN_LAUNCH_STATUS();
}
