// Bilinear resize (align_corners = True) of NHWC feature maps for gfx950 (MI355X), forward +
// backward.
//
// Replaces `F.interpolate(feat, size=fused_size, mode='bilinear', align_corners=True)` in the HTC
// semantic branch (mmdet/models/mask_heads/fused_semantic_head.py:88-93): the five FPN levels
// are brought to the fusion level's size before their lateral 1x1 convs.
//
// Arithmetic follows torch's upsample_bilinear2d (the op the reference calls):
//   scale = (in - 1) / (out - 1)  (0 when out == 1);  src = scale * dst;  i0 = (int)src;
//   i1 = i0 + (i0 < in - 1);  l1 = src - i0;  l0 = 1 - l1;
//   y = l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11).
//
// HBM-bound: one thread per (output pixel, channel quad), consecutive lanes on consecutive quads
// of the same pixel, so each of the four taps is a contiguous C*4-byte row read with 16-byte
// loads.  Algorithmic bytes per output pixel: C*4 written + at most 4*C*4 read (the taps of
// neighbouring pixels overlap and hit L2; the unique input is (H*W)/(Ho*Wo) of that).
// Backward: the same mapping scatters the four weighted contributions with hardware fp32
// atomics into a zero-initialised dx (2x upsampling: every input pixel receives <= 9 adds).
#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Axis {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Axis source_index(int dst, int in_size, int out_size, float scale) {
#pragma clang fp contract(off)
  Axis a;
  if (in_size == out_size) {
    a.i0 = a.i1 = dst;
    a.l0 = 1.f;
    a.l1 = 0.f;
    return a;
  }
  // rounded product (no FMA contraction into `src - i0`): torch's CPU kernel rounds it, and at
  // src ~ 40 the fused form moves lambda by 4e-6 (1e-5 in the output)
  const float src = scale * (float)dst;
  a.i0 = min((int)src, in_size - 1);
  a.i1 = a.i0 + (a.i0 < in_size - 1 ? 1 : 0);
  a.l1 = fminf(fmaxf(src - (float)a.i0, 0.f), 1.f);
  a.l0 = 1.f - a.l1;
  return a;
}

template <bool BWD>
__global__ __launch_bounds__(256) void resize_bilinear_nhwc_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C, int Ho,
    int Wo, float scale_h, float scale_w) {
  // forward: src = x [N,H,W,C], dst = y [N,Ho,Wo,C];  backward: src = dy, dst = dx (atomics)
  const int quads = C >> 2;
  const long long total = (long long)N * Ho * Wo * quads;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(t % quads);
    long long p = t / quads;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const Axis ay = source_index(oy, H, Ho, scale_h);
    const Axis ax = source_index(ox, W, Wo, scale_w);
    const size_t base = (size_t)n * H * W;
    const size_t o00 = ((base + (size_t)ay.i0 * W + ax.i0) * C) + q * 4;
    const size_t o01 = ((base + (size_t)ay.i0 * W + ax.i1) * C) + q * 4;
    const size_t o10 = ((base + (size_t)ay.i1 * W + ax.i0) * C) + q * 4;
    const size_t o11 = ((base + (size_t)ay.i1 * W + ax.i1) * C) + q * 4;
    const size_t oo = ((((size_t)n * Ho + oy) * Wo + ox) * C) + q * 4;
    if (!BWD) {
      const f32x4 v00 = *reinterpret_cast<const f32x4*>(src + o00);
      const f32x4 v01 = *reinterpret_cast<const f32x4*>(src + o01);
      const f32x4 v10 = *reinterpret_cast<const f32x4*>(src + o10);
      const f32x4 v11 = *reinterpret_cast<const f32x4*>(src + o11);
      const f32x4 r = ay.l0 * (ax.l0 * v00 + ax.l1 * v01) + ay.l1 * (ax.l0 * v10 + ax.l1 * v11);
      *reinterpret_cast<f32x4*>(dst + oo) = r;
    } else {
      const f32x4 g = *reinterpret_cast<const f32x4*>(src + oo);
      const float w00 = ay.l0 * ax.l0, w01 = ay.l0 * ax.l1, w10 = ay.l1 * ax.l0,
                  w11 = ay.l1 * ax.l1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (w00 != 0.f) unsafeAtomicAdd(dst + o00 + j, w00 * g[j]);
        if (w01 != 0.f) unsafeAtomicAdd(dst + o01 + j, w01 * g[j]);
        if (w10 != 0.f) unsafeAtomicAdd(dst + o10 + j, w10 * g[j]);
        if (w11 != 0.f) unsafeAtomicAdd(dst + o11 + j, w11 * g[j]);
      }
    }
  }
}

int launch(bool bwd, const float* src, float* dst, int N, int H, int W, int C, int Ho, int Wo,
           int align_corners, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return BGS_ERR_INVALID_ARG;
  if (!src || !dst) return BGS_ERR_INVALID_ARG;
  if (!align_corners) return BGS_ERR_UNSUPPORTED;      // the HTC semantic head's only mode
  if (C % 4 != 0 || (uintptr_t)src % 16 != 0 || (uintptr_t)dst % 16 != 0)
    return BGS_ERR_UNSUPPORTED;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;          // grid-stride beyond 64 blocks per CU
  if (bwd)
    hipLaunchKernelGGL((resize_bilinear_nhwc_kernel<true>), dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, H, W, C, Ho, Wo, sh, sw);
  else
    hipLaunchKernelGGL((resize_bilinear_nhwc_kernel<false>), dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, H, W, C, Ho, Wo, sh, sw);
  BGS_RETURN_LAUNCH_STATUS();
}

}  // namespace

extern "C" int bgs_resize_bilinear_nhwc_f32(const float* x, float* y, int N, int H, int W, int C,
                                            int Ho, int Wo, int align_corners,
                                            bgs_stream_t stream) {
  return launch(false, x, y, N, H, W, C, Ho, Wo, align_corners, stream);
}

extern "C" int bgs_resize_bilinear_nhwc_bwd_f32(const float* dy, float* dx, int N, int H, int W,
                                                int C, int Ho, int Wo, int align_corners,
                                                bgs_stream_t stream) {
  return launch(true, dy, dx, N, H, W, C, Ho, Wo, align_corners, stream);
}
