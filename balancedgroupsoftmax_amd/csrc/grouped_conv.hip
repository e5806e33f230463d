// Grouped 3x3 convolution, NHWC fp32 — conv2 of the ResNeXt bottleneck (cfg 5: ResNeXt-101 64x4d,
// mmdet/models/backbones/resnext.py:12-92: groups = 64, 4/8/16/32 channels per group in
// layer1..4), with the folded eval-mode BN bias and ReLU in the epilogue.
//
// A group is a tiny dense conv (cg x cg x 9 MACs per pixel: 144 .. 9216), far below an MFMA tile
// for cg = 4/8 and < 3 % of the network's flops in total, so this kernel stays on the vector
// ALUs: one thread = one output pixel x 4 consecutive output channels (always inside one group:
// cg % 4 == 0); a wave covers 64 consecutive channel quads of a pixel, so the input loads of a
// group are shared by its cg/4 threads through the L1 and the weight rows [co][r][s][0..cg) are
// contiguous 16-byte loads that stay cache-resident across pixels.
#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CG>
__global__ __launch_bounds__(256) void grouped_conv3x3_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, int stride, int relu) {
  const int quads = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * quads;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int q = (int)(e % quads);
    size_t t = e / quads;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int co = q * 4;
    const int g0 = (co / CG) * CG;            // first input channel of this group
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * stride - 1 + r;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * stride - 1 + s;
        if (wi < 0 || wi >= W) continue;
        const float* xp = x + (((size_t)n * H + hi) * W + wi) * C + g0;
        const float* wp = w + ((size_t)co * 9 + r * 3 + s) * CG;
#pragma unroll
        for (int c = 0; c < CG; c += 4) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)k * 9 * CG + c);
            acc[k] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
          }
        }
      }
    }
    if (bias) acc += *reinterpret_cast<const f32x4*>(bias + co);
    if (relu) {
      acc[0] = fmaxf(acc[0], 0.f);
      acc[1] = fmaxf(acc[1], 0.f);
      acc[2] = fmaxf(acc[2], 0.f);
      acc[3] = fmaxf(acc[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + e * 4) = acc;
  }
}

}  // namespace

extern "C" int bgs_grouped_conv3x3_nhwc_f32(const float* x, const float* w, const float* bias,
                                            float* y, int N, int H, int W, int C, int groups,
                                            int stride, int relu, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || (stride != 1 && stride != 2))
    return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (C % groups != 0) return BGS_ERR_INVALID_ARG;
  const int cg = C / groups;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias) % 16 != 0)
    return BGS_ERR_INVALID_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  size_t grid = (total + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipStream_t st = (hipStream_t)stream;
#define BGS_GC_LAUNCH(CG_)                                                                       \
  hipLaunchKernelGGL((grouped_conv3x3_kernel<CG_>), dim3((unsigned)grid), dim3(256), 0, st, x, w, \
                     bias, y, N, H, W, C, Ho, Wo, stride, relu)
  if (cg == 4) BGS_GC_LAUNCH(4);
  else if (cg == 8) BGS_GC_LAUNCH(8);
  else if (cg == 16) BGS_GC_LAUNCH(16);
  else if (cg == 32) BGS_GC_LAUNCH(32);
  else return BGS_ERR_UNSUPPORTED;   // ResNeXt 32x4d / 64x4d use 4..32 channels per group
#undef BGS_GC_LAUNCH
  BGS_RETURN_LAUNCH_STATUS();
}
