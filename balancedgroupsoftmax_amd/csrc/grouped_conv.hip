// Grouped 3x3 convolution, NHWC fp32 — conv2 of the ResNeXt bottleneck (cfg 5: ResNeXt-101 64x4d,
// mmdet/models/backbones/resnext.py:12-92: groups = 64, 4/8/16/32 channels per group in
// layer1..4), with the folded eval-mode BN bias and ReLU in the epilogue.
//
// v_mfma_f32_16x16x4_f32 per (64 output pixels x 16 output channels) wave tile: a 16-channel
// output tile is one group (cg = 16), half a group (cg = 32: two K halves) or several whole
// groups (cg = 4 / 8: the B operand is block-diagonal — lanes whose input quad belongs to
// another group feed zeros; 4x / 2x redundant MACs on layers that are < 1 % of the flops).
// Per filter tap every lane makes ONE 16-byte load of the input (pixel i, channel quad j) per
// 16-pixel sub-tile and ONE 16-byte load of the weights (output channel n, quad j); the four
// floats are the A / B operands of four consecutive MFMAs (k index = lane group j):
//     acc[a] += x[pix(a,i)+tap][16*slab + 4j + u] * w[n][tap][.. 4j + u ..],  u = 0..3
// A workgroup = 4 waves = the same 64 pixels x 4 neighbouring channel tiles (input reuse in L1).
// Backward (`selectp = 0` on the X101 configs): the data gradient is the SAME kernel on dy with
// the per-group transposed, flipped filter (`up = 2` reads dy as if zero-upsampled for the
// stride-2 blocks); the weight gradient is grouped_wgrad3x3_kernel below.
// (First version: one thread per pixel x 4 channels on the vector ALUs re-read its 4 x 9 x cg
//  weights per pixel: 0.69 ms per layer3 conv = 3.6 TFLOP/s, 47 % of the X101 step.)
#include <stdlib.h>

#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CG>
__global__ __launch_bounds__(256) void grouped_conv3x3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, int stride, int relu,
    int up) {
  constexpr int KH = CG >= 16 ? CG / 16 : 1;       // 16-channel K slabs per tap
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, j = lane >> 4;
  const int M = N * Ho * Wo;
  const int m0 = blockIdx.x * 64;
  const int ct = blockIdx.y * 4 + wave;            // 16-channel output tile of this wave
  if (ct * 16 >= C) return;
  const int n_out = ct * 16 + i;                   // this lane's B column (output channel)
  // input channel slab read by this tile: the group(s) of its 16 output channels
  const int grp_first = (ct * 16) / CG;            // first group touched
  const int in0 = (CG >= 16) ? grp_first * CG : ct * 16;   // first input channel of the slab(s)
  // block-diagonal mask for CG < 16: lane's input quad j belongs to group (in0 + 4j) / CG
  bool w_live = true;
  int w_off = 4 * j;                               // offset of the quad inside the group's cg inputs
  if (CG < 16) {
    const int g_in = (in0 + 4 * j) / CG, g_out = n_out / CG;
    w_live = g_in == g_out;
    w_off = (in0 + 4 * j) - g_in * CG;
  }
  // geometry of the 4 pixels this lane feeds as A rows (sub-tile a, row i)
  int hi0[4], wi0[4];
  const float* xb[4];
  bool ok[4];
  const int hw = Ho * Wo;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int m = m0 + a * 16 + i;
    ok[a] = m < M;
    const int mm = ok[a] ? m : 0;
    const int n = mm / hw, rem = mm - n * hw;
    const int ho = rem / Wo, wo = rem - ho * Wo;
    hi0[a] = ho * stride - 1;
    wi0[a] = wo * stride - 1;
    xb[a] = x + (size_t)n * H * W * C;
  }
  f32x4 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wrow = w + (size_t)n_out * 9 * CG;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (w_live) bv = *reinterpret_cast<const f32x4*>(wrow + (r * 3 + s) * CG + kh * 16 + w_off);
        f32x4 av[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          int hi = hi0[a] + r, wi = wi0[a] + s;
          bool in = ok[a] && hi >= 0 && wi >= 0;
          if (up == 2) {          // input read as if zero-upsampled by 2 (stride-2 data gradient)
            in = in && !((hi | wi) & 1);
            hi >>= 1;
            wi >>= 1;
          }
          av[a] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (in && hi < H && wi < W)
            av[a] = *reinterpret_cast<const f32x4*>(xb[a] + ((size_t)hi * W + wi) * C + in0 +
                                                    kh * 16 + 4 * j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < 4; ++a)
            // the TRANSPOSED product (filter rows x pixel columns: the two operand registers
            // swapped): in the D layout — column = lane & 15, rows 4 (lane >> 4) + t — a lane then
            // owns four CONSECUTIVE output channels of one pixel: one 16-byte store per sub-tile
            // instead of four 4-byte ones (the vector-memory instruction count bounds these kernels)
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[u], av[a][u], acc[a], 0, 0, 0);
      }
    }
  }
  const int c_out = ct * 16 + 4 * j;               // first of this lane's four output channels
  f32x4 bsv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bsv = *reinterpret_cast<const f32x4*>(bias + c_out);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int m = m0 + a * 16 + i;
    if (m >= M) continue;
    f32x4 v = acc[a] + bsv;
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + (size_t)m * C + c_out) = v;
  }
}

// Stride-1 layers (30 of the 33 grouped convs of an X101 forward, and their data gradients) with an
// LDS-RESIDENT input patch.  A workgroup owns an 8 x 16 pixel tile x 64 channels (whole groups); the
// 10 x 18 pixel patch of those 64 channels (45 KB) reaches LDS ONCE by `global_load_lds_dwordx4` —
// for a grouped conv the 64-channel slab is the ENTIRE reduction of its groups, there is no K loop —
// and the nine taps read shifted rows (the kernel above re-fetches every tap from L1 / L2: 45
// global loads per wave; 62 us per layer3 conv against 13 us of HBM traffic).  Wave w computes the
// 128 pixels x 16 output channels 16 w .. 16 w + 15 of the slab as eight 16-pixel sub-tiles (tile
// rows); its 9 (x 2 for cg = 32) filter fragments live in registers for the whole kernel.
//   * LDS layout: pixel P at 256 P bytes; the 16-byte channel quad q sits in slot q ^ (P & 15): the
//     DMA writes lane-linearly, so the swizzle is applied to the SOURCE quad each lane fetches, and a
//     fragment read (16 consecutive pixels of a tile row, same quad) hits 16 distinct slots;
//   * transposed product as above: a lane owns four consecutive output channels of one pixel.
constexpr int GTH = 8, GTW = 16, GPH = GTH + 2, GPW = GTW + 2, GPIX = GPH * GPW;   // 180 patch pixels
constexpr int GDMA = (GPIX * 16 + 63) / 64;                                       // 45 DMA pieces

__device__ __forceinline__ void gc_glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __attribute__((aligned(16))) unsigned g_gc_zero_page[4];

// BF: the bf16 mode of cfg[4] — both operands rounded to bf16 (RNE) on their way into
// v_mfma_f32_16x16x16_bf16 (one MFMA per tap and sub-tile instead of four fp32 ones), fp32 accumulate.
typedef short bf16x4_bits __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4_bits gc_pack_bf16(const f32x4 v) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t lo = __builtin_convertvector(f32x2_t{v[0], v[1]}, bf16x2_t);
  const bf16x2_t hi = __builtin_convertvector(f32x2_t{v[2], v[3]}, bf16x2_t);
  return __builtin_bit_cast(bf16x4_bits, u32x2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)});
}

template <int CG, bool BF = false>
__global__ __launch_bounds__(256, CG == 32 ? 2 : 3) void grouped_conv3x3_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int N, int H, int W, int C, int tiles_y, int tiles_x, int relu) {
  constexpr int KH = CG >= 16 ? CG / 16 : 1;       // 16-channel K slabs per tap
  __shared__ __attribute__((aligned(1024))) unsigned char lds[GDMA * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, j = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int c0 = blockIdx.y * 64;                  // first channel of the slab
  const int h0 = ty * GTH - 1, w0 = tx * GTW - 1;  // input coords of patch (0, 0)

  // ---- patch DMA: piece d (1 KB) = patch pixels 4 d .. 4 d + 3; lane -> (pixel, slot); wave w issues
  //      pieces w, w + 4, ..
  const float* xn = x + (size_t)n * H * W * C + c0;
  for (int d = wave; d < GDMA; d += 4) {
    const int P = 4 * d + (lane >> 4);
    const int q = (lane & 15) ^ (P & 15);          // source quad of this slot
    const int pr = P / GPW, pc = P - pr * GPW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = P < GPIX && hi >= 0 && hi < H && wi >= 0 && wi < W;
    const float* src = in ? xn + ((size_t)hi * W + wi) * C + 4 * q
                          : reinterpret_cast<const float*>(g_gc_zero_page);
    gc_glds16(src, lds + d * 1024);
  }

  // ---- this wave's filter fragments (registers) while the patch is in flight
  const int ct16 = c0 + wave * 16;                 // first output channel of this wave
  const int n_out = ct16 + i;                      // B' row of this lane (output channel)
  const int grp_first = ct16 / CG;
  const int in0 = (CG >= 16) ? grp_first * CG : ct16;      // first input channel of the K slab(s)
  bool w_live = true;
  int w_off = 4 * j;
  if (CG < 16) {
    const int g_in = (in0 + 4 * j) / CG, g_out = n_out / CG;
    w_live = g_in == g_out;
    w_off = (in0 + 4 * j) - g_in * CG;
  }
  f32x4 bv[9][KH];
  const float* wrow = w + (size_t)n_out * 9 * CG;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      bv[tap][kh] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (w_live) bv[tap][kh] = *reinterpret_cast<const f32x4*>(wrow + tap * CG + kh * 16 + w_off);
    }
  f32x4 acc[GTH];
#pragma unroll
  for (int a = 0; a < GTH; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int qbase = (in0 - c0) / 4 + j;            // quad (within the slab) this lane feeds per K slab
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
      for (int a = 0; a < GTH; ++a) {
        const int P = (a + r) * GPW + i + s2;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          const int q = qbase + kh * 4;
          const f32x4 av = *reinterpret_cast<const f32x4*>(lds + P * 256 + ((q ^ (P & 15)) << 4));
          if (BF) {
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gc_pack_bf16(bv[r * 3 + s2][kh]),
                                                               gc_pack_bf16(av), acc[a], 0, 0, 0);
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[r * 3 + s2][kh][u], av[u], acc[a], 0, 0, 0);
          }
        }
      }
    }
  }
  const int c_out = ct16 + 4 * j;
  f32x4 bsv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bsv = *reinterpret_cast<const f32x4*>(bias + c_out);
  const int wo = tx * GTW + i;
#pragma unroll
  for (int a = 0; a < GTH; ++a) {
    const int ho = ty * GTH + a;
    if (ho >= H || wo >= W) continue;
    f32x4 v = acc[a] + bsv;
    if (relu) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) v[tt] = fmaxf(v[tt], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + (((size_t)n * H + ho) * W + wo) * C + c_out) = v;
  }
}

// stride-1 launch (forward, or the data gradient with the per-group transposed filter)
int launch_grouped_s1(const float* x, const float* w, const float* bias, float* y, int N, int H, int W,
                      int C, int cg, int relu, hipStream_t st, bool bf = false) {
  const int tiles_y = (H + GTH - 1) / GTH, tiles_x = (W + GTW - 1) / GTW;
  dim3 grid((unsigned)(N * tiles_y * tiles_x), (unsigned)(C / 64));
  bgs_internal_census_bump(BGS_CENSUS_GROUPED_LDS);
#define BGS_GL_LAUNCH(CG_)                                                                              \
  do {                                                                                                  \
    if (bf)                                                                                             \
      hipLaunchKernelGGL((grouped_conv3x3_lds_kernel<CG_, true>), grid, dim3(256), 0, st, x, w, bias, y, \
                         N, H, W, C, tiles_y, tiles_x, relu);                                           \
    else                                                                                                \
      hipLaunchKernelGGL((grouped_conv3x3_lds_kernel<CG_, false>), grid, dim3(256), 0, st, x, w, bias, y, \
                         N, H, W, C, tiles_y, tiles_x, relu);                                           \
  } while (0)
  if (cg == 4) BGS_GL_LAUNCH(4);
  else if (cg == 8) BGS_GL_LAUNCH(8);
  else if (cg == 16) BGS_GL_LAUNCH(16);
  else if (cg == 32) BGS_GL_LAUNCH(32);
  else return BGS_ERR_UNSUPPORTED;
#undef BGS_GL_LAUNCH
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}

int g_grouped_lds = -1;      // BGS_GROUPED_LDS=0: the direct-load kernel everywhere (A/B, tests)
bool grouped_lds_enabled() {
  if (g_grouped_lds < 0) {
    const char* e = getenv("BGS_GROUPED_LDS");
    g_grouped_lds = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_grouped_lds != 0;
}

}  // namespace

extern "C" int bgs_grouped_conv3x3_nhwc_f32(const float* x, const float* w, const float* bias,
                                            float* y, int N, int H, int W, int C, int groups,
                                            int stride, int relu, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || (stride != 1 && stride != 2))
    return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (C % groups != 0 || C % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int cg = C / groups;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias) % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((C / 16 + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && C % 64 == 0 && grouped_lds_enabled())
    return launch_grouped_s1(x, w, bias, y, N, H, W, C, cg, relu, st);
#define BGS_GC_LAUNCH(CG_)                                                                        \
  hipLaunchKernelGGL((grouped_conv3x3_mfma_kernel<CG_>), grid, dim3(256), 0, st, x, w, bias, y, N, \
                     H, W, C, Ho, Wo, stride, relu, 1)
  if (cg == 4) BGS_GC_LAUNCH(4);
  else if (cg == 8) BGS_GC_LAUNCH(8);
  else if (cg == 16) BGS_GC_LAUNCH(16);
  else if (cg == 32) BGS_GC_LAUNCH(32);
  else return BGS_ERR_UNSUPPORTED;   // ResNeXt 32x4d / 64x4d use 4..32 channels per group
#undef BGS_GC_LAUNCH
  BGS_RETURN_LAUNCH_STATUS();
}


namespace {

// Weight gradient of the grouped 3x3 conv: dw[co][r][s][cl] = sum_m dy[m][co] * x[pix(m)+tap][g*cg+cl].
// Thread = one (co, cl) pair with nine accumulators; a workgroup covers 256 / CG output channels
// and one chunk of `chunk` output pixels; partial sums per chunk go to `part[chunk][C*9*CG]` and
// are added in a fixed order by grouped_wgrad_reduce_kernel (bitwise reproducible, like
// conv_wgrad.hip).  This layer family is < 1 % of the X101 flops: a plain VALU kernel.
template <int CG>
__global__ __launch_bounds__(256) void grouped_wgrad3x3_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int N,
    int H, int W, int C, int Ho, int Wo, int stride, int chunk) {
  const int pair = blockIdx.y * 256 + threadIdx.x;         // (co, cl)
  const int co = pair / CG, cl = pair - co * CG;
  if (co >= C) return;
  const int ci = (co / CG) * CG + cl;
  const int M = N * Ho * Wo;
  const int m_begin = blockIdx.x * chunk, m_end = min(M, m_begin + chunk);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const int hw = Ho * Wo;
  for (int m = m_begin; m < m_end; ++m) {
    const float g = dy[(size_t)m * C + co];
    const int n = m / hw, rem = m - n * hw;
    const int ho = rem / Wo, wo = rem - ho * Wo;
    const float* xb = x + (size_t)n * H * W * C + ci;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * stride - 1 + r;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * stride - 1 + s;
        if (wi < 0 || wi >= W) continue;
        acc[r * 3 + s] = fmaf(g, xb[((size_t)hi * W + wi) * C], acc[r * 3 + s]);
      }
    }
  }
  float* o = part + (size_t)blockIdx.x * C * 9 * CG + (size_t)co * 9 * CG + cl;
#pragma unroll
  for (int t = 0; t < 9; ++t) o[t * CG] = acc[t];
}

__global__ __launch_bounds__(256) void grouped_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ dw,
                                                                   float* __restrict__ db,
                                                                   const float* __restrict__ dy,
                                                                   int total, int chunks, int C,
                                                                   long long M, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < total) {
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += part[(size_t)c * total + e];
    dw[e] = accumulate ? dw[e] + v : v;
  }
  if (db && e < C) {                     // bias gradient: column sums of dy (fixed order)
    float v = 0.f;
    for (long long m = 0; m < M; ++m) v += dy[m * C + e];
    db[e] = accumulate ? db[e] + v : v;
  }
}

}  // namespace

// bgs_grouped_conv3x3_nhwc_f32 with both operands rounded to bf16 (fp32 tensors, fp32 accumulate):
// the bf16 mode of cfg[4] (the reference's fp16 autocast, mmdet/core/fp16/decorators.py:9-160).
// Stride 1 and C % 64 == 0 only (the LDS-resident kernel); BGS_ERR_UNSUPPORTED otherwise — the caller
// then uses the fp32 entry point.
extern "C" int bgs_grouped_conv3x3_nhwc_bf16ops(const float* x, const float* w, const float* bias,
                                                float* y, int N, int H, int W, int C, int groups,
                                                int stride, int relu, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0) return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (stride != 1 || C % groups != 0 || C % 64 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias) % 16 != 0) return BGS_ERR_INVALID_ARG;
  if ((long long)N * H * W > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  return launch_grouped_s1(x, w, bias, y, N, H, W, C, C / groups, relu, (hipStream_t)stream, true);
}

// Data gradient of bgs_grouped_conv3x3_nhwc_f32: dy [N,Ho,Wo,C] -> dx [N,H,W,C].  wt = the filter
// transposed inside each group and flipped: wt[g*cg+cl][2-r][2-s][co_local] = w[g*cg+co_local][r][s][cl].
extern "C" int bgs_grouped_conv3x3_dgrad_nhwc_f32(const float* dy, const float* wt, float* dx, int N,
                                                  int H, int W, int C, int groups, int stride,
                                                  bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || (stride != 1 && stride != 2))
    return BGS_ERR_INVALID_ARG;
  if (!dy || !wt || !dx) return BGS_ERR_INVALID_ARG;
  if (C % groups != 0 || C % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int cg = C / groups;
  if (((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)dx) % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;   // dy dims
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((C / 16 + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  const float* nobias = nullptr;
  if (stride == 1 && C % 64 == 0 && grouped_lds_enabled())      // dy and dx have the same size
    return launch_grouped_s1(dy, wt, nobias, dx, N, H, W, C, cg, 0, st);
  // "input" = dy (Ho x Wo, zero-upsampled when stride 2), "output" = dx (H x W), unit stride, pad 1
#define BGS_GD_LAUNCH(CG_)                                                                         \
  hipLaunchKernelGGL((grouped_conv3x3_mfma_kernel<CG_>), grid, dim3(256), 0, st, dy, wt, nobias, dx, \
                     N, Ho, Wo, C, H, W, 1, 0, stride)
  if (cg == 4) BGS_GD_LAUNCH(4);
  else if (cg == 8) BGS_GD_LAUNCH(8);
  else if (cg == 16) BGS_GD_LAUNCH(16);
  else if (cg == 32) BGS_GD_LAUNCH(32);
  else return BGS_ERR_UNSUPPORTED;
#undef BGS_GD_LAUNCH
  BGS_RETURN_LAUNCH_STATUS();
}

static int grouped_wgrad_chunks(long long M, int* chunk) {
  *chunk = 1024;
  return (int)((M + *chunk - 1) / *chunk);
}

extern "C" size_t bgs_grouped_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int C, int groups,
                                                            int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  int chunk;
  const int chunks = grouped_wgrad_chunks((long long)N * Ho * Wo, &chunk);
  return (size_t)chunks * C * 9 * (C / groups) * sizeof(float);
}

// dw [C,3,3,C/groups] (+)= ..., db [C] (+)= column sums of dy when db != NULL.
extern "C" int bgs_grouped_conv3x3_wgrad_nhwc_f32(const float* x, const float* dy, float* dw,
                                                  float* db, int N, int H, int W, int C, int groups,
                                                  int stride, int accumulate, void* workspace,
                                                  bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || (stride != 1 && stride != 2))
    return BGS_ERR_INVALID_ARG;
  if (!x || !dy || !dw || !workspace) return BGS_ERR_INVALID_ARG;
  if (C % groups != 0) return BGS_ERR_UNSUPPORTED;
  const int cg = C / groups;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  int chunk;
  const int chunks = grouped_wgrad_chunks(M, &chunk);
  float* part = reinterpret_cast<float*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)chunks, (unsigned)((C * cg + 255) / 256));
#define BGS_GW_LAUNCH(CG_)                                                                       \
  hipLaunchKernelGGL((grouped_wgrad3x3_kernel<CG_>), grid, dim3(256), 0, st, x, dy, part, N, H, W, \
                     C, Ho, Wo, stride, chunk)
  if (cg == 4) BGS_GW_LAUNCH(4);
  else if (cg == 8) BGS_GW_LAUNCH(8);
  else if (cg == 16) BGS_GW_LAUNCH(16);
  else if (cg == 32) BGS_GW_LAUNCH(32);
  else return BGS_ERR_UNSUPPORTED;
#undef BGS_GW_LAUNCH
  if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
  const int total = C * 9 * cg;
  hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, part, dw, db, dy, total, chunks, C, M, accumulate);
  BGS_RETURN_LAUNCH_STATUS();
}
