// Implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950 (MI355X).
//
// Replaces, for the BAGS detector's forward pass, what the reference delegates to
// cuDNN / cuBLAS through nn.Conv2d + (frozen, eval-mode) BatchNorm2d + ReLU and nn.Linear:
//   ResNet bottlenecks     mmdet/models/backbones/resnet.py:220-266  (norm_eval=True :535-542)
//   FPN laterals / outputs mmdet/models/necks/fpn.py:101-141
//   RPN head               mmdet/models/anchor_heads/rpn_head.py:30-35
//   RoI head FCs           mmdet/models/bbox_heads/convfc_bbox_head.py:132-168
//
// One kernel:  y[m, j] = act( sum_k A[m, k] * Wt[j, k] + bias[j] (+ residual) )
//   m = (n, ho, wo) output pixel, j = output channel, k = (r, s, c) filter tap x input channel.
//   * activations are NHWC fp32 (channels contiguous == the GEMM K axis contiguous), weights
//     are [Cout][R][S][Cin] (K-major rows): both operands stream with 16-byte loads;
//   * v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bit-exact fma chain (no TF32 on gfx950)
//     -> the 1e-4 parity budget of the detector is met by construction;
//   * workgroup = 4 waves (2 x 2), each wave owns MB x NB tiles of 32 x 32 (64 accumulator
//     registers for the 128 x 128 tile), BK = 16, LDS rows hold the even-k then the odd-k
//     values so that a lane's 8 fragment operands of a K tile are two ds_read_b128,
//     double-buffered LDS + register prefetch of the next K tile: one barrier per K tile;
//   * epilogue fuses the folded-BN bias, the residual add (same-shape for bottlenecks,
//     nearest-2x-upsampled for the FPN top-down path) and ReLU, and writes NHWC directly.
#include <stdlib.h>

#include "conv_args.h"

using namespace bgs_conv;

namespace {

// UP = 1: plain convolution.  UP = 2: the input is read as if it had been zero-upsampled by 2
// (x_virtual[2h, 2w] = x[h, w], zeros elsewhere) — the data gradient of a stride-2 convolution.
template <int MB, int NB, int BK, int UP>
__global__ __launch_bounds__(kThreads) void conv_igemm_f32_kernel(ConvArgs p) {
  constexpr int BM = 64 * MB, BN = 64 * NB;
  constexpr int LDK = BK + 4;            // LDS row stride in floats
  constexpr int KQ = BK / 4;             // 16-byte quads per row of a K tile
  constexpr int RP = kThreads / KQ;      // rows staged per pass
  constexpr int PA = BM / RP, PB = BN / RP;
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;          // workgroup-uniform
  const int m0 = (vtile / p.tiles_n) * BM, n0 = (vtile % p.tiles_n) * BN;

  // ---- staging role of this thread: 4 consecutive k (one 16-byte load) of rows srow + 64*pass
  const int kq = tid % KQ;
  const int srow = tid / KQ;
  int a_hi0[PA], a_wi0[PA];
  const float* a_base[PA];
  bool a_ok[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + srow + RP * q;
    a_ok[q] = m < p.M;
    const int mm = a_ok[q] ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_hi0[q] = ho * p.stride - p.pad;
    a_wi0[q] = wo * p.stride - p.pad;
    a_base[q] = p.x + (size_t)n * p.H * p.W * p.Cin;
  }
  const float* b_base[PB];
  bool b_ok[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int j = n0 + srow + RP * q;
    b_ok[q] = j < p.Cout;
    b_base[q] = p.w + (size_t)(b_ok[q] ? j : 0) * p.K;
  }
  // (r, s, c) of this thread's k-quad, advanced incrementally from tile to tile
  const int nk_all = (p.K + BK - 1) / BK;
  const int kt_begin = p.partial ? blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.partial ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
  int kg = kt_begin * BK + kq * 4;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }

  f32x4 ra[PA], rb[PB];
  auto load_tile = [&]() {
    const bool kok = kg < p.K;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      int hi = a_hi0[q] + kr, wi = a_wi0[q] + ks;
      bool ok = a_ok[q] && kok && hi >= 0 && wi >= 0;
      if (UP == 2) {
        ok = ok && !((hi | wi) & 1);
        hi >>= 1;
        wi >>= 1;
      }
      ok = ok && hi < p.H && wi < p.W;
      ra[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok)
        ra[q] = *reinterpret_cast<const f32x4*>(a_base[q] + ((size_t)hi * p.W + wi) * p.Cin + kc);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      rb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b_ok[q] && kok) rb[q] = *reinterpret_cast<const f32x4*>(b_base[q] + kg);
    }
    // advance to the next K tile
    kg += BK;
    kc += BK;
    while (kc >= p.Cin) {
      kc -= p.Cin;
      if (++ks == p.S) {
        ks = 0;
        ++kr;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    // k = 4*kq + {0,1,2,3}: evens (k = 4kq, 4kq+2) -> slots 2kq, 2kq+1; odds -> BK/2 + 2kq, +1
    for (int q = 0; q < PA; ++q) {
      float* d = &As[buf][(srow + RP * q) * LDK + kq * 2];
      *reinterpret_cast<f32x2*>(d) = f32x2{ra[q][0], ra[q][2]};
      *reinterpret_cast<f32x2*>(d + BK / 2) = f32x2{ra[q][1], ra[q][3]};
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      float* d = &Bs[buf][(srow + RP * q) * LDK + kq * 2];
      *reinterpret_cast<f32x2*>(d) = f32x2{rb[q][0], rb[q][2]};
      *reinterpret_cast<f32x2*>(d + BK / 2) = f32x2{rb[q][1], rb[q][3]};
    }
  };

  // (Tried: two accumulators per wave for the single-tile configuration, to break the dependent
  //  MFMA chain — no gain: with 4 workgroups per CU the other waves already fill those bubbles.)
  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = kt_end - kt_begin;
  // Software pipeline (one barrier per K tile, placed MID-tile): the fragments are read one
  // half K tile ahead of the MFMAs that consume them, so the LDS latency, the global->LDS
  // staging of the next tile and the barrier all sit behind 16+ in-flight MFMAs instead of in
  // front of them:
  //     [read frags (t, half 1)] [MFMA half 0] [stage tile t+1 -> LDS] [barrier]
  //     [read frags (t+1, half 0)] [MFMA half 1] ...
  // (measured before: ds_read x8 -> wait -> 32 MFMAs -> vmcnt wait -> ds_write -> barrier, all
  //  serial inside a wave: 0.71 of the sustained MFMA rate.)
  constexpr int NV = BK / 8;      // f32x4 fragments per tile row per K tile
  constexpr int HV = NV / 2;      // ... per half
  const int frow = lane & 31, fk = lane >> 5;
  f32x4 fa[2][MB][HV], fb[2][NB][HV];
  auto read_half = [&](int buf, int h, int slot) {
    const float* Ab = &As[buf][(wm * 32 * MB + frow) * LDK + fk * (BK / 2) + 4 * h * HV];
    const float* Bb = &Bs[buf][(wn * 32 * NB + frow) * LDK + fk * (BK / 2) + 4 * h * HV];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int v = 0; v < HV; ++v)
        fa[slot][a][v] = *reinterpret_cast<const f32x4*>(Ab + a * 32 * LDK + 4 * v);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < HV; ++v)
        fb[slot][b][v] = *reinterpret_cast<const f32x4*>(Bb + b * 32 * LDK + 4 * v);
  };
  auto mfma_half = [&](int slot) {
#pragma unroll
    for (int v = 0; v < HV; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][v][j], fb[slot][b][v][j],
                                                              acc[a][b], 0, 0, 0);
  };
  load_tile();
  store_tile(0);
  __syncthreads();
  read_half(0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tile();            // global loads of tile kt+1 in flight under the MFMAs
    read_half(buf, 1, 1);
    mfma_half(0);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    if (more) read_half(buf ^ 1, 0, 0);
    mfma_half(1);
  }

  conv_store_tile<MB, NB>(p, acc, m0, n0, wm, wn, lane);
}

// y[m][j] = epilogue( sum_z partial[z][m][j] ): fixed summation order (bitwise reproducible).
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(ConvArgs p, int splits) {
  const size_t total = (size_t)p.M * p.Cout;
  const int hw = p.Ho * p.Wo;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (size_t)gridDim.x * 256) {
    const int m = (int)(e / p.Cout);
    const int j = (int)(e - (size_t)m * p.Cout);
    float v = p.partial[e];
    for (int z = 1; z < splits; ++z) v += p.partial[(size_t)z * total + e];
    if (p.bias) v += p.bias[j];
    if (p.res_mode == 1) {
      v += p.res[e];
    } else if (p.res_mode == 2 || p.res_mode == 3) {
      const int n = m / hw;
      const int rem = m - n * hw;
      const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
      if (p.res_mode == 2) {
        v += p.res[(((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + j];
      } else {
        const size_t r0 = (((size_t)n * (p.Ho * 2) + ho * 2) * (p.Wo * 2) + wo * 2) * p.Cout + j;
        const size_t down = (size_t)p.Wo * 2 * p.Cout;
        v += (p.res[r0] + p.res[r0 + p.Cout]) + (p.res[r0 + down] + p.res[r0 + down + p.Cout]);
      }
    }
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.mask) v = p.mask[e] > 0.f ? v : 0.f;
    p.y[e] = v;
  }
}

// Same reduction, four consecutive channels per thread with 16-byte accesses (Cout % 4 == 0 and
// 16-byte aligned buffers: every layer but the fused 15-channel RPN head): identical arithmetic per
// element, a quarter of the instructions.
__global__ __launch_bounds__(256) void conv_splitk_epilogue4_kernel(ConvArgs p, int splits) {
  const int c4n = p.Cout >> 2;
  const size_t total4 = (size_t)p.M * c4n;
  const size_t total = (size_t)p.M * p.Cout;
  const int hw = p.Ho * p.Wo;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total4;
       q += (size_t)gridDim.x * 256) {
    const int m = (int)(q / c4n);
    const int j = (int)(q - (size_t)m * c4n) * 4;
    const size_t e = (size_t)m * p.Cout + j;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.partial + e);
    for (int z = 1; z < splits; ++z)
      v += *reinterpret_cast<const f32x4*>(p.partial + (size_t)z * total + e);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + j);
    if (p.res_mode == 1) {
      v += *reinterpret_cast<const f32x4*>(p.res + e);
    } else if (p.res_mode == 2 || p.res_mode == 3) {
      const int n = m / hw;
      const int rem = m - n * hw;
      const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
      if (p.res_mode == 2) {
        v += *reinterpret_cast<const f32x4*>(
            p.res + (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + j);
      } else {
        const size_t r0 = (((size_t)n * (p.Ho * 2) + ho * 2) * (p.Wo * 2) + wo * 2) * p.Cout + j;
        const size_t down = (size_t)p.Wo * 2 * p.Cout;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.res + r0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.res + r0 + p.Cout);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p.res + r0 + down);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p.res + r0 + down + p.Cout);
        v += (a + b) + (c + d);
      }
    }
    if (p.relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    if (p.mask) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + e);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
    }
    *reinterpret_cast<f32x4*>(p.y + e) = v;
  }
}

// 3x3 / stride-2 / pad-1 max pooling, NHWC (ResNet stem, resnet.py:452), 4 channels per thread.
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x,
                                                                float* __restrict__ y, int N,
                                                                int H, int W, int C, int Ho,
                                                                int Wo) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    size_t t = i / c4n;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int hi = ho * 2 - 1 + dy;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int wi = wo * 2 - 1 + dx;
        if (wi < 0 || wi >= W) continue;
        const f32x4 v =
            *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + hi) * W + wi) * C + c4 * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

// Backward of maxpool3x3s2_nhwc_kernel: one thread per INPUT (pixel, channel quad) gathers from the
// <= 4 output windows that contain it; an input element receives a window's gradient iff it is that
// window's FIRST maximum in (row, column) scan order — the element torch's max_pool2d backward
// (the op the reference calls, resnet.py:452) routes to.  No atomics: deterministic.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_nhwc_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int N,
                                                                    int H, int W, int C, int Ho,
                                                                    int Wo) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * H * W * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    size_t t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const f32x4 mine = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    // windows (ho, wo) with ho*2-1 <= h <= ho*2+1
    for (int ho = (h >> 1); ho <= ((h + 1) >> 1); ++ho) {
      if (ho < 0 || ho >= Ho) continue;
      for (int wo = (w >> 1); wo <= ((w + 1) >> 1); ++wo) {
        if (wo < 0 || wo >= Wo) continue;
        const f32x4 gy = *reinterpret_cast<const f32x4*>(dy + ((((size_t)n * Ho + ho) * Wo + wo) * C + c4 * 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // is (h, w) the first maximum of this window for channel u?
          bool first = true;
          for (int dyy = 0; dyy < 3 && first; ++dyy) {
            const int hi = ho * 2 - 1 + dyy;
            if (hi < 0 || hi >= H) continue;
            for (int dxx = 0; dxx < 3; ++dxx) {
              const int wi = wo * 2 - 1 + dxx;
              if (wi < 0 || wi >= W) continue;
              if (hi == h && wi == w) continue;
              const float v = x[(((size_t)n * H + hi) * W + wi) * C + c4 * 4 + u];
              const bool before = hi < h || (hi == h && wi < w);
              if (v > mine[u] || (before && v == mine[u]) || v != v) {   // NaN propagates like torch
                first = false;
                break;
              }
            }
          }
          if (first) g[u] += gy[u];
        }
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = g;
  }
}

}  // namespace

// y = epilogue(sum of the split-K partial slabs): shared with conv_bfx.hip.
int bgs_internal_conv_splitk_epilogue(ConvArgs& p, int splits, hipStream_t st) {
  const size_t total = (size_t)p.M * p.Cout;
  auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
  if (p.Cout % 4 == 0 && al16(p.partial) && al16(p.y) && al16(p.bias) && al16(p.res) &&
      al16(p.mask)) {
    size_t g = (total / 4 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(conv_splitk_epilogue4_kernel, dim3((unsigned)g), dim3(256), 0, st, p,
                       splits);
  } else {
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)g), dim3(256), 0, st, p,
                       splits);
  }
  BGS_RETURN_LAUNCH_STATUS();
}

// Tuning knobs are read from the environment ONCE (first launch), not per launch.
struct ConvKnobs {
  int tile = 0, bk = 0, splitk = 0, noswizzle = 0;
  bool has_splitk = false;
  ConvKnobs() {
    if (const char* e = getenv("BGS_CONV_TILE")) tile = atoi(e);
    if (const char* e = getenv("BGS_CONV_BK")) bk = atoi(e);
    if (const char* e = getenv("BGS_CONV_SPLITK")) { splitk = atoi(e); has_splitk = true; }
    noswizzle = getenv("BGS_CONV_NOSWIZZLE") != nullptr;
  }
};
static ConvKnobs& conv_knobs() {
  static ConvKnobs k;
  return k;
}
static int g_last_tile = 0, g_last_bk = 0, g_last_up = 0, g_last_splits = 0;

// Shared launcher: tile / K-tile choice from the per-layer sweeps of the R50-FPN shapes
// (tools/conv_sweep.py, profiles/r1i_conv_sweep.txt, r1q_conv_sweep_bk*.txt): the 128x128 tile
// only pays for the two huge-M, deep-K 3x3 convs on the stride-4 maps; everywhere else the 64x64
// tile wins or ties because it exposes 4x more workgroups; thin outputs (Cout <= 64) with a huge
// M use 128x64.  K tile: 32 for deep reductions with the 64x64 tile (+5-15 % on K >= 512).
static int launch_conv(ConvArgs& p, int up, hipStream_t st, void* workspace = nullptr,
                       size_t workspace_bytes = 0) {
  const long long M = p.M;
  const ConvKnobs& knobs = conv_knobs();
  const int force = knobs.tile;  // tuning hook: BGS_CONV_TILE=22|21|11 forces a tile configuration
  int tile = 11;
  if (p.Cout > 64 && p.K >= 1152 && M >= 100000) tile = 22;
  else if (p.Cout <= 64 && M >= 400000) tile = 21;
  if (force == 22 || force == 21 || force == 11) tile = force;
  int bk = (p.K >= 512 && tile == 11) ? 32 : 16;
  if (knobs.bk == 32 || knobs.bk == 16) bk = knobs.bk;
  // split-K: only for the 64x64 tile when the grid cannot fill the chip (< ~2 workgroups per CU)
  // and the reduction is deep enough to share out (>= 8 K tiles per slice)
  int splits = 1;
  p.partial = nullptr;
  p.kt_per_split = 0;
  // (BGS_CONV_TILE + BGS_CONV_SPLITK together also split the larger tiles: tuning sweeps)
  const bool force_split = force != 0 && knobs.has_splitk;
  if ((tile == 11 || force_split) && workspace) {
    const long long wgs = ((M + 63) / 64) * ((p.Cout + 63) / 64);
    const int nk = (p.K + bk - 1) / bk;
    // measured on the cfg[1] shapes (profiles/r1z_splitk_sweep*.txt): < 600 workgroups: aim at
    // ~1100 (l4.c2 49 -> 80, fpn.out3 34 -> 56, fc1 70 -> 93 TFLOP/s); 600..1500: 2 slices (+5 %)
    int want = 1;
    if (wgs < 600) want = (int)((1100 + wgs - 1) / wgs);
    else if (wgs < 1500) want = 2;
    if (want > nk / 8) want = nk / 8;
    if (want > 8) want = 8;
    if (knobs.has_splitk) {
      const int f = knobs.splitk;
      if (f >= 1 && f <= 16) want = f < nk ? f : nk;
    }
    while (want > 1 && (size_t)want * (size_t)M * p.Cout * sizeof(float) > workspace_bytes) --want;
    if (want > 1) {
      p.kt_per_split = (nk + want - 1) / want;
      splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
      p.partial = reinterpret_cast<float*>(workspace);
    }
  }
  g_last_tile = tile; g_last_bk = bk; g_last_up = up; g_last_splits = splits;
#define BGS_CONV_LAUNCH2(MB_, NB_, BK_, UP_)                                                     \
  hipLaunchKernelGGL((conv_igemm_f32_kernel<MB_, NB_, BK_, UP_>), grid, dim3(kThreads), 0, st, p)
#define BGS_CONV_LAUNCH(MB_, NB_, BM_, BN_)                                                      \
  do {                                                                                           \
    p.tiles_m = (int)((M + BM_ - 1) / BM_);                                                      \
    p.tiles_n = (p.Cout + BN_ - 1) / BN_;                                                        \
    p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;                                                   \
    if (knobs.noswizzle) p.chunk = 0;                                                            \
    dim3 grid((unsigned)(p.chunk ? 8 * p.chunk : p.tiles_m * p.tiles_n), 1u, (unsigned)splits);   \
    if (up == 2) {                                                                               \
      if (bk == 32) BGS_CONV_LAUNCH2(MB_, NB_, 32, 2);                                           \
      else BGS_CONV_LAUNCH2(MB_, NB_, 16, 2);                                                    \
    } else {                                                                                     \
      if (bk == 32) BGS_CONV_LAUNCH2(MB_, NB_, 32, 1);                                           \
      else BGS_CONV_LAUNCH2(MB_, NB_, 16, 1);                                                    \
    }                                                                                            \
  } while (0)
  if (tile == 22) BGS_CONV_LAUNCH(2, 2, 128, 128);
  else if (tile == 21) BGS_CONV_LAUNCH(2, 1, 128, 64);
  else BGS_CONV_LAUNCH(1, 1, 64, 64);
#undef BGS_CONV_LAUNCH
#undef BGS_CONV_LAUNCH2
  if (splits > 1) {
    if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
    return bgs_internal_conv_splitk_epilogue(p, splits, st);
  }
  BGS_RETURN_LAUNCH_STATUS();
}

// Tuning / test hooks (process-wide; not for concurrent use).  tile 0 = auto | 11 | 21 | 22;
// bk 0 = auto | 16 | 32; splitk 0 = auto | 1..16 (with a forced tile also splits the larger tiles).
extern "C" void bgs_conv_tuning(int tile, int bk, int splitk, int noswizzle) {
  ConvKnobs& k = conv_knobs();
  k.tile = tile;
  k.bk = bk;
  k.splitk = splitk;
  k.has_splitk = splitk >= 1;
  k.noswizzle = noswizzle;
}

extern "C" int bgs_conv_last_launch(int* tile, int* bk, int* up, int* splits) {
  if (tile) *tile = g_last_tile;
  if (bk) *bk = g_last_bk;
  if (up) *up = g_last_up;
  if (splits) *splits = g_last_splits;
  return BGS_OK;
}

// Scratch for the split-K path of the two entry points below (0 is always legal: no split).
extern "C" size_t bgs_conv2d_workspace_bytes(long long M, int Cout) {
  if (M <= 0 || Cout <= 0) return 0;
  const long long wgs = ((M + 63) / 64) * ((Cout + 63) / 64);
  if (wgs >= 1500) return 0;
  if (conv_knobs().has_splitk) return (size_t)16 * (size_t)M * Cout * sizeof(float);   // sweeps
  return (size_t)(wgs < 600 ? 8 : 2) * (size_t)M * Cout * sizeof(float);
}

extern "C" int bgs_conv2d_nhwc_f32_ws(const float* x, const float* w, const float* bias,
                                      const float* residual, float* y, int N, int H, int W,
                                      int Cin, int Cout, int R, int S, int stride, int pad,
                                      int relu, int residual_mode, void* workspace,
                                      size_t workspace_bytes, bgs_stream_t stream);

extern "C" int bgs_conv2d_nhwc_f32(const float* x, const float* w, const float* bias,
                                   const float* residual, float* y, int N, int H, int W, int Cin,
                                   int Cout, int R, int S, int stride, int pad, int relu,
                                   int residual_mode, bgs_stream_t stream) {
  return bgs_conv2d_nhwc_f32_ws(x, w, bias, residual, y, N, H, W, Cin, Cout, R, S, stride, pad,
                                relu, residual_mode, nullptr, 0, stream);
}

extern "C" int bgs_conv2d_nhwc_f32_ws(const float* x, const float* w, const float* bias,
                                      const float* residual, float* y, int N, int H, int W,
                                      int Cin, int Cout, int R, int S, int stride, int pad,
                                      int relu, int residual_mode, void* workspace,
                                      size_t workspace_bytes, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (Cin % 4 != 0) return BGS_ERR_UNSUPPORTED;  // pad the input channels to a multiple of 4
  if (((uintptr_t)x | (uintptr_t)w) % 16 != 0) return BGS_ERR_INVALID_ARG;
  if (residual_mode < 0 || residual_mode > 2 || (residual_mode != 0 && !residual))
    return BGS_ERR_INVALID_ARG;
  ConvArgs p;
  p.x = x; p.w = w; p.bias = bias; p.res = residual; p.mask = nullptr; p.y = y;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return BGS_ERR_INVALID_ARG;
  if (residual_mode == 2 && ((p.Ho & 1) || (p.Wo & 1))) return BGS_ERR_INVALID_ARG;
  const long long M = (long long)N * p.Ho * p.Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  p.M = (int)M;
  p.K = R * S * Cin;
  p.relu = relu;
  p.res_mode = residual_mode;
  return launch_conv(p, 1, (hipStream_t)stream, workspace, workspace_bytes);
}

// Data gradient of bgs_conv2d_nhwc_f32: dx[n,h,w,ci] = sum_{r,s,co} dy[n,ho,wo,co] W[co,r,s,ci]
// over the (ho, wo) with ho*stride - pad + r == h.  Same implicit-GEMM kernel with the roles of
// the channel axes swapped: "input" = dy [N,Ho,Wo,Cout] (virtually zero-upsampled by the
// stride), filter = wt[ci][R-1-r][S-1-s][co] (the caller passes this re-laid-out copy), padding
// R-1-pad, unit stride, output H x W x Cin.  Epilogue: + residual (mode 1 same shape; mode 3 =
// 2x2 sum-pool of a twice-as-large map: the backward of the FPN nearest-2x top-down add), then
// the ReLU-backward mask of the tensor that fed the forward conv.
extern "C" int bgs_conv2d_dgrad_nhwc_f32_ws(const float* dy, const float* wt,
                                            const float* residual, const float* mask, float* dx,
                                            int N, int H, int W, int Cin, int Cout, int R, int S,
                                            int stride, int pad, int residual_mode, void* workspace,
                                            size_t workspace_bytes, bgs_stream_t stream);

extern "C" int bgs_conv2d_dgrad_nhwc_f32(const float* dy, const float* wt, const float* residual,
                                         const float* mask, float* dx, int N, int H, int W,
                                         int Cin, int Cout, int R, int S, int stride, int pad,
                                         int residual_mode, bgs_stream_t stream) {
  return bgs_conv2d_dgrad_nhwc_f32_ws(dy, wt, residual, mask, dx, N, H, W, Cin, Cout, R, S, stride,
                                      pad, residual_mode, nullptr, 0, stream);
}

extern "C" int bgs_conv2d_dgrad_nhwc_f32_ws(const float* dy, const float* wt,
                                            const float* residual, const float* mask, float* dx,
                                            int N, int H, int W, int Cin, int Cout, int R, int S,
                                            int stride, int pad, int residual_mode, void* workspace,
                                            size_t workspace_bytes, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!dy || !wt || !dx) return BGS_ERR_INVALID_ARG;
  if (stride != 1 && stride != 2) return BGS_ERR_UNSUPPORTED;
  if (Cout % 4 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)dy | (uintptr_t)wt) % 16 != 0) return BGS_ERR_INVALID_ARG;
  if (!(residual_mode == 0 || residual_mode == 1 || residual_mode == 3) ||
      (residual_mode != 0 && !residual))
    return BGS_ERR_INVALID_ARG;
  if (R - 1 - pad < 0 || S - 1 - pad < 0) return BGS_ERR_UNSUPPORTED;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return BGS_ERR_INVALID_ARG;
  ConvArgs p;
  p.x = dy; p.w = wt; p.bias = nullptr; p.res = residual; p.mask = mask; p.y = dx;
  p.N = N; p.H = Ho; p.W = Wo; p.Cin = Cout; p.Cout = Cin; p.R = R; p.S = S;
  p.stride = 1; p.pad = R - 1 - pad;   // square filters on this path: R == S, same padding
  if (R != S) return BGS_ERR_UNSUPPORTED;
  p.Ho = H; p.Wo = W;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  p.M = (int)M;
  p.K = R * S * Cout;
  p.relu = 0;
  p.res_mode = residual_mode;
  return launch_conv(p, stride, (hipStream_t)stream, workspace, workspace_bytes);
}

// [N, C <= 4, H, W] image batch -> [N, H, W, 4] (channels zero-padded to 16-byte pixels): the stem conv's
// input layout.  As tensor ops: permute + pad + contiguous = a fill and a strided copy.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int C, size_t hw, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / hw, pix = i - n * hw;
    const float* src = x + n * C * hw + pix;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    v[0] = src[0];
    if (C > 1) v[1] = src[hw];
    if (C > 2) v[2] = src[2 * hw];
    if (C > 3) v[3] = src[3 * hw];
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

extern "C" int bgs_nchw_to_nhwc4_f32(const float* x, float* y, int N, int C, int H, int W,
                                     bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || !x || !y) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)y % 16 != 0) return BGS_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W, total = (size_t)N * hw;
  size_t grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, C,
                     hw, total);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C,
                                         bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !x || !y) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || ((uintptr_t)x | (uintptr_t)y) % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  size_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
  BGS_RETURN_LAUNCH_STATUS();
}

// Backward of bgs_maxpool3x3s2_nhwc_f32 (ResNet stem when `frozen_stages < 1`): x [N,H,W,C] (the
// pool's input), dy [N,Ho,Wo,C] -> dx [N,H,W,C] (overwritten).
extern "C" int bgs_maxpool3x3s2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H,
                                             int W, int C, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !x || !dy || !dx) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * H * W * (C / 4);
  size_t grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_nhwc_kernel, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, x, dy, dx, N, H, W, C, Ho, Wo);
  BGS_RETURN_LAUNCH_STATUS();
}
