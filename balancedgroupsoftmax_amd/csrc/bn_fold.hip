// Eval-mode BatchNorm folded into the convolution filter, and the backward of the fold (gfx950).
//
// Every conv of the reference trunk is Conv2d -> BatchNorm2d in eval mode (`norm_eval=True`,
// mmdet/models/backbones/resnet.py:535-542), so  bn(conv(x, w)) == conv(x, w * s) + (beta - mu * s)
// with s = gamma / sqrt(var + eps).  The conv kernels consume the folded filter in [Cout, R, S, Cin]
// (KRSC) layout; the parameters stay in the reference's [Cout, Cin, R, S] layout (checkpoints).
//
// With a trainable trunk (`selectp = 0`, tools/train.py:49-57) the fold has to sit on the autograd
// tape: written as tensor ops it is ~20 elementwise / permute launches per conv forward + backward
// — 1,200 launches and 6.3 ms of a 30 ms step (profiles/r5c_sp0_prof_summary.md).  Here it is ONE
// launch each way, one workgroup per output channel:
//   forward   wf[co][r][s][ci] = w[co][ci][r][s] * s[co]   (LDS transpose: both sides coalesced)
//             bf[co]           = beta - mu * s (+ conv_bias * s);   without BN: s = 1, bf = conv_bias
//   backward  dw[co][ci][r][s] = dwf[co][r][s][ci] * s[co]
//             ds               = sum_{r,s,ci} dwf * w  -  mu * dbf  (+ conv_bias * dbf)
//             dgamma = ds / sqrt(var + eps),  dbeta = dbf,  dconv_bias = dbf * s
#include "bgs_common.h"

namespace {

constexpr int kFoldBlock = 256;
constexpr int kFoldMaxK = 8192;        // R * S * CinP floats staged in LDS (32 KB): 3x3x512 = 4608

__global__ __launch_bounds__(kFoldBlock) void fold_fwd_kernel(
    const float* __restrict__ w, const float* __restrict__ cbias, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
    float eps, int Cin, int RS, int CinP, float* __restrict__ wf, float* __restrict__ bf) {
  __shared__ float t[kFoldMaxK];
  const int co = blockIdx.x, tid = threadIdx.x;
  const int K = Cin * RS, KP = CinP * RS;
  float s = 1.f;
  if (gamma) s = gamma[co] / sqrtf(var[co] + eps);
  const float* wr = w + (size_t)co * K;
  for (int i = tid; i < K; i += kFoldBlock) t[i] = wr[i] * s;       // [ci][rs], coalesced
  __syncthreads();
  float* o = wf + (size_t)co * KP;
  for (int i = tid; i < KP; i += kFoldBlock) {                       // [rs][ci], coalesced
    const int rs = i / CinP, ci = i - rs * CinP;
    o[i] = ci < Cin ? t[ci * RS + rs] : 0.f;
  }
  if (tid == 0) {
    float b = cbias ? cbias[co] : 0.f;
    if (gamma) b = (beta[co] - mean[co] * s) + b * s;               // shift (+ conv.bias * scale)
    bf[co] = b;
  }
}

__global__ __launch_bounds__(kFoldBlock) void fold_bwd_kernel(
    const float* __restrict__ dwf, const float* __restrict__ dbf, const float* __restrict__ w,
    const float* __restrict__ cbias, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, int Cin, int RS, int CinP, float* __restrict__ dw,
    float* __restrict__ dcbias, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float t[kFoldMaxK];
  __shared__ float red[kFoldBlock / BGS_WAVE];
  const int co = blockIdx.x, tid = threadIdx.x;
  const int K = Cin * RS, KP = CinP * RS;
  float s = 1.f, inv = 1.f;
  if (gamma) {
    inv = 1.f / sqrtf(var[co] + eps);
    s = gamma[co] * inv;
  }
  const float* g = dwf + (size_t)co * KP;
  for (int i = tid; i < KP; i += kFoldBlock) t[i] = g[i];            // [rs][ci], coalesced
  __syncthreads();
  const float* wr = w + (size_t)co * K;
  float* dwr = dw ? dw + (size_t)co * K : nullptr;
  float acc = 0.f;
  for (int i = tid; i < K; i += kFoldBlock) {                        // [ci][rs], coalesced
    const int ci = i / RS, rs = i - ci * RS;
    const float gv = t[rs * CinP + ci];
    acc += gv * wr[i];
    if (dwr) dwr[i] = gv * s;
  }
  acc = bgs::wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float ds = 0.f;
    for (int v = 0; v < kFoldBlock / BGS_WAVE; ++v) ds += red[v];
    const float db = dbf ? dbf[co] : 0.f;
    if (gamma) {
      ds -= mean[co] * db;
      if (cbias) ds += cbias[co] * db;
      if (dgamma) dgamma[co] = ds * inv;
      if (dbeta) dbeta[co] = db;
    }
    if (dcbias) dcbias[co] = db * s;
  }
}

}  // namespace

extern "C" int bgs_fold_conv_bn_fwd(const float* w, const float* conv_bias, const float* gamma,
                                    const float* beta, const float* mean, const float* var, float eps,
                                    int Cout, int Cin, int R, int S, int cin_padded, float* wf,
                                    float* bf, bgs_stream_t stream) {
  if (!w || !wf || !bf || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || cin_padded < Cin)
    return BGS_ERR_INVALID_ARG;
  if (gamma && (!beta || !mean || !var)) return BGS_ERR_INVALID_ARG;
  if ((long long)R * S * cin_padded > kFoldMaxK) return BGS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fold_fwd_kernel, dim3((unsigned)Cout), dim3(kFoldBlock), 0, (hipStream_t)stream, w,
                     conv_bias, gamma, beta, mean, var, eps, Cin, R * S, cin_padded, wf, bf);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_fold_conv_bn_bwd(const float* dwf, const float* dbf, const float* w,
                                    const float* conv_bias, const float* gamma, const float* mean,
                                    const float* var, float eps, int Cout, int Cin, int R, int S,
                                    int cin_padded, float* dw, float* dconv_bias, float* dgamma,
                                    float* dbeta, bgs_stream_t stream) {
  if (!dwf || !w || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || cin_padded < Cin)
    return BGS_ERR_INVALID_ARG;
  if (gamma && (!mean || !var)) return BGS_ERR_INVALID_ARG;
  if ((long long)R * S * cin_padded > kFoldMaxK) return BGS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fold_bwd_kernel, dim3((unsigned)Cout), dim3(kFoldBlock), 0, (hipStream_t)stream, dwf,
                     dbf, w, conv_bias, gamma, mean, var, eps, Cin, R * S, cin_padded, dw, dconv_bias,
                     dgamma, dbeta);
  BGS_RETURN_LAUNCH_STATUS();
}
