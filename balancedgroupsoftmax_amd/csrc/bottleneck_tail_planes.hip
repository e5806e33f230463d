// The second half of a frozen ResNet-50 layer1 bottleneck in one launch, planes form (round 6): conv2 (3x3 / stride 1,
// 64 -> 64, folded BN, ReLU) -> conv3 (1x1, 64 -> 256, folded BN) + residual + ReLU
// (mmdet/models/backbones/resnet.py:239-266).  Same contract, arithmetic and results (BIT-IDENTICAL) as
// conv3x3_c3_fused_bfx_kernel of conv_bfx.hip and as the two launches it fuses; what changes is the shape of the work:
//   * a workgroup owns 8 x 8 output pixels (that kernel: 8 x 16): 2100 tiles of the stride-4 map on 768 slots are 2.73
//     rounds instead of the 1.37 rounds of double-size tiles that execute as two;
//   * conv2's input has 64 channels in all, so its WHOLE 10 x 10 x 64 patch is loaded once (7 buffer loads per thread;
//     outside the image: an offset past the descriptor = 0), split once into three bf16 planes in LDS
//     ([plane][16-channel k step][patch row of 12 slots][32 B], k halves swapped on odd rows: conv3x3_planes.hip's
//     conflict-free layout) and read by all 36 k steps (4 chunks x 9 taps, the halo kernels' order) behind ONE barrier;
//     wave (wm, wn) owns 32 pixels x 32 channels, its filter fragments straight from L2 one step ahead;
//   * the conv2 tile (+ bias2, ReLU) goes through an fp32 LDS transpose into conv3's A planes ([plane][k step][pixel][32
//     B], conv1x1_planes.hip's layout) — the same split3 of the same fp32 values the unfused conv2 stores;
//   * conv3: wave w owns 64 pixels x channels 64 w .. (acc[2][2]), four k steps, filter fragments from L2;
//   * epilogue in two halves of 32 pixels through the LDS transpose: bias3, then the residual, then the clamp.
// LDS 46 KB (the patch planes; every later stage overlays them): three workgroups per CU.
#include <stdlib.h>

#include "conv_args.h"
#include "bfx_split.h"

using namespace bgs_conv;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}

// x (4 consecutive k) -> three planes of 4 packed bf16 each: conv_bfx.hip's split3, verbatim
__device__ __forceinline__ void split3p(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
  const f32x4 r = {bfx_resid_lo(hi[0], v[0]), bfx_resid_hi(hi[0], v[1]), bfx_resid_lo(hi[1], v[2]),
                   bfx_resid_hi(hi[1], v[3])};
  mid = u32x2{pk_bf16(r[0], r[1]), pk_bf16(r[2], r[3])};
  const f32x4 r2 = {bfx_resid_lo(mid[0], r[0]), bfx_resid_hi(mid[0], r[1]), bfx_resid_lo(mid[1], r[2]),
                    bfx_resid_hi(mid[1], r[3])};
  lo = u32x2{pk_bf16(r2[0], r2[1]), pk_bf16(r2[2], r2[3])};
}

struct TailArgs {
  const float* x;        // [N,H,W,64]
  const __bf16* ws2;     // split conv2 filter [3][36][64][16]: k step = tap * 4 + 16-channel chunk
  const float* bias2;    // [64] or null
  const __bf16* ws3;     // split conv3 filter [3][4][256][16]
  const float* bias3;    // [256] or null
  const float* res;      // [N,H,W,256] or null
  float* y;              // [N,H,W,256]
  int N, H, W, relu3;
  int tiles_y, tiles_x, tiles, chunk;
};

constexpr int kOob = 0x7f000000;   // byte offset past every descriptor this kernel builds: the load returns 0

__global__ __launch_bounds__(kThreads, 3) void bottleneck_tail_planes_kernel(TailArgs g) {
  constexpr int NS = 3, CM = 64, CO3 = 256, KC2 = 36, KC3 = 4;
  constexpr int PW = 10, PS = 12, PSLOTS = PW * PS;                 // 10 x 10 patch in rows of 12 slots
  constexpr int AQ = PW * PW * 16, AQT = (AQ + kThreads - 1) / kThreads;   // 1600 fp32 quads: 7 per thread (the 7th: 64 threads)
  constexpr int KS = PSLOTS * 32, PL = 4 * KS;                      // 3840 B per k step, 15,360 per plane
  constexpr int LDS_BYTES = NS * PL;                                // 46,080
  constexpr int LD2 = CM + 4, S2_BYTES = 64 * LD2 * 4;              // conv2 transpose tile: 17,408 at offset 0
  constexpr int Y_OFF = 18 * 1024;                                  // conv3's A planes behind it: 18,432 .. 43,008
  constexpr int P3_CH = 64 * 32, P3_PL = KC3 * P3_CH;               // 2 KB per k step, 8 KB per plane
  constexpr int LD4 = CO3 + 8;                                      // epilogue half tile: 32 x 264 x 4 = 33,792 (overlays both)
  static_assert(S2_BYTES <= Y_OFF && Y_OFF + NS * P3_PL <= LDS_BYTES && 32 * LD4 * 4 <= LDS_BYTES, "overlays");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fk = lane >> 5;
  const int vtile = (int)((blockIdx.x & 7) * g.chunk + (blockIdx.x >> 3));
  if (vtile >= g.tiles) return;                                     // workgroup-uniform
  const int per_img = g.tiles_y * g.tiles_x;
  const int n = vtile / per_img, trem = vtile - n * per_img;
  const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
  const int h0 = ty * 8 - 1, w0 = tx * 8 - 1;

  // ---- phase 0: the whole patch, 16 quads per patch pixel
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.x), 0, (int)((size_t)g.N * g.H * g.W * CM * 4), 0x00020000);
  f32x4 ra[AQT];
  int a_dst[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int q = tid + kThreads * i;
    const bool use = q < AQ;
    const int pix = use ? q >> 4 : 0, kq = q & 15;
    const int ppy = pix / PW, ppx = pix - ppy * PW;
    const int hi = h0 + ppy, wi = w0 + ppx;
    const bool in = use && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
    const int off = in ? (((n * g.H + hi) * g.W + wi) * CM + kq * 4) * 4 : kOob;
    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
    const int quad = kq & 3;
    a_dst[i] = use ? (kq >> 2) * KS + (ppy * PS + ppx) * 32 + (((quad >> 1) ^ (ppy & 1)) << 4) + (quad & 1) * 8 : -1;
  }

  // conv2 filter fragments: wave (wm, wn) reads rows 32 wn .. of the 64
  const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws2), 0, NS * KC2 * CM * 32, 0x00020000);
  const int b2_lane = ((wn * 32 + frow) * 16 + fk * 8) * 2;         // bytes
  auto load_b2 = [&](int kc, bf16x8 (&dst)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b2_rsrc, b2_lane, (s * KC2 + kc) * CM * 32, 0));
  };
  bf16x8 fbA[NS], fbB[NS];
  load_b2(0, fbA);                                                  // (chunk 0, tap 0)

#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    if (a_dst[i] < 0) continue;
    u32x2 hh, mm, ll;
    split3p(ra[i], hh, mm, ll);
    unsigned char* d = lds + a_dst[i];
    *reinterpret_cast<u32x2*>(d) = hh;
    *reinterpret_cast<u32x2*>(d + PL) = mm;
    *reinterpret_cast<u32x2*>(d + 2 * PL) = ll;
  }
  __syncthreads();

  // ---- phase 1: conv2, 36 k steps (16-channel chunks ascending, nine taps per chunk); wave = 32 pixels x 32 channels
  int a_frag[2];                                                    // [parity of dy]
  {
    const int m = wm * 32 + frow;
    const int py = m >> 3, px = m & 7;
#pragma unroll
    for (int par = 0; par < 2; ++par) a_frag[par] = (py * PS + px) * 32 + ((fk ^ ((py + par) & 1)) << 4);
  }
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
  for (int t = 0; t < KC2; ++t) {
    const int c16 = t / 9, tap = t % 9, dy = tap / 3, dx = tap % 3;
    const int t1 = t + 1;
    const int kc_next = t1 < KC2 ? (t1 % 9) * 4 + t1 / 9 : 0;       // (the last prefetch: unused)
    if (t & 1) load_b2(kc_next, fbA);
    else load_b2(kc_next, fbB);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&fb)[NS] = (t & 1) ? fbB : fbA;
    bf16x8 fa[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
      fa[s] = *reinterpret_cast<const bf16x8*>(lds + s * PL + c16 * KS + a_frag[dy & 1] + (dy * PS + dx) * 32);
#pragma unroll
    for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
      for (int i = 0; i <= tt; ++i) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[tt - i], acc2, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // conv3 filter fragments: wave w reads rows 64 w ..; the first k step travels under phase 2
  const __amdgpu_buffer_rsrc_t b3_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws3), 0, NS * KC3 * CO3 * 32, 0x00020000);
  const int b3_lane = ((wave * 64 + frow) * 16 + fk * 8) * 2;
  auto load_b3 = [&](int kc, bf16x8 (&dst)[NS][2]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        dst[s][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b3_rsrc, b3_lane, ((s * KC3 + kc) * CO3 + 32 * b) * 32, 0));
  };
  bf16x8 f3a[NS][2], f3b[NS][2];
  load_b3(0, f3a);

  // ---- phase 2: conv2 tile -> fp32 transpose -> bias2, ReLU, split -> conv3's A planes
  __syncthreads();                                                  // every wave is done with the patch planes
  float* scratch = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    scratch[i * LD2 + wn * 32 + (lane & 31)] = acc2[r];
  }
  __syncthreads();
  {
    const int c4 = (tid & 15) * 4, r0 = tid >> 4;                   // 16 threads per pixel, 16 pixels per pass
    f32x4 bias2 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias2) bias2 = *reinterpret_cast<const f32x4*>(g.bias2 + c4);
    const int kc2 = c4 >> 4, kq2 = (c4 & 15) >> 2;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int i = r0 + ps * 16;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD2 + c4);
      v += bias2;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      u32x2 hh, mm, ll;
      split3p(v, hh, mm, ll);
      unsigned char* d = lds + Y_OFF + kc2 * P3_CH + i * 32 + (((kq2 >> 1) ^ ((i >> 3) & 1)) << 4) + (kq2 & 1) * 8;
      *reinterpret_cast<u32x2*>(d) = hh;
      *reinterpret_cast<u32x2*>(d + P3_PL) = mm;
      *reinterpret_cast<u32x2*>(d + 2 * P3_PL) = ll;
    }
  }
  __syncthreads();

  // ---- phase 3: conv3, wave w = 64 pixels x channels 64 w ..
  int a3_frag[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = a * 32 + frow;
    a3_frag[a] = Y_OFF + m * 32 + ((fk ^ ((m >> 3) & 1)) << 4);
  }
  f32x16 acc3[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[a][b][r] = 0.f;
#pragma unroll
  for (int kc = 0; kc < KC3; ++kc) {
    if (kc + 1 < KC3) {
      if (kc & 1) load_b3(kc + 1, f3a);
      else load_b3(kc + 1, f3b);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&fb3)[NS][2] = (kc & 1) ? f3b : f3a;
    bf16x8 fa3[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a) fa3[s][a] = *reinterpret_cast<const bf16x8*>(lds + a3_frag[a] + kc * P3_CH + s * P3_PL);
#pragma unroll
    for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
      for (int i = 0; i <= tt; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc3[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[i][a], fb3[tt - i][b], acc3[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                                  // every wave is done with conv3's planes

  // ---- phase 4: two halves of 32 pixels; bias3, then the residual, then the clamp; 16-byte loads / stores
  const int e4 = (tid & 63) * 4, er0 = tid >> 6;                    // 64 threads per pixel, 4 pixels per pass
  f32x4 bias3 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias3) bias3 = *reinterpret_cast<const f32x4*>(g.bias3 + e4);
  const int out_bytes = (int)((size_t)g.N * g.H * g.W * CO3 * 4);
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.res ? g.res : g.x), 0, g.res ? out_bytes : 0, 0x00020000);     // 0 bytes: every read is 0
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.y, 0, out_bytes, 0x00020000);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    f32x4 rs[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int m = a * 32 + er0 + ps * 4;
      const int ho = min(ty * 8 + (m >> 3), g.H - 1), wo = min(tx * 8 + (m & 7), g.W - 1);
      rs[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, (((n * g.H + ho) * g.W + wo) * CO3 + e4) * 4, 0, 0));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[i * LD4 + wave * 64 + b * 32 + (lane & 31)] = acc3[a][b][r];
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int i = er0 + ps * 4;
      const int m = a * 32 + i;
      const int ho = ty * 8 + (m >> 3), wo = tx * 8 + (m & 7);
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD4 + e4);
      v += bias3;
      v += rs[ps];
      if (g.relu3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      if (ho < g.H && wo < g.W)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, (((n * g.H + ho) * g.W + wo) * CO3 + e4) * 4, 0, 0);
    }
    __syncthreads();
  }
}

}  // namespace

// -1: not eligible (the caller runs conv3x3_c3_fused_bfx_kernel)
int bgs_internal_bottleneck_tail_planes(const float* x, const void* w2split, const float* bias2, const void* w3split,
                                        const float* bias3, const float* residual, float* y, int N, int H, int W,
                                        int relu3, hipStream_t st) {
  const long long lim = kOob;
  if ((long long)N * H * W * 256 * 4 >= lim) return -1;
  TailArgs g;
  g.x = x;
  g.ws2 = reinterpret_cast<const __bf16*>(w2split);
  g.bias2 = bias2;
  g.ws3 = reinterpret_cast<const __bf16*>(w3split);
  g.bias3 = bias3;
  g.res = residual;
  g.y = y;
  g.N = N; g.H = H; g.W = W; g.relu3 = relu3;
  g.tiles_y = (H + 7) / 8;
  g.tiles_x = (W + 7) / 8;
  g.tiles = N * g.tiles_y * g.tiles_x;
  g.chunk = (g.tiles + 7) / 8;
  hipLaunchKernelGGL(bottleneck_tail_planes_kernel, dim3((unsigned)(8 * g.chunk)), dim3(kThreads), 0, st, g);
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}
