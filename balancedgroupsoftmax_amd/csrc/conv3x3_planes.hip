// 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores for the SMALL maps (round 6): the recipe of
// conv1x1_planes.hip applied to the halo convolution.  Same contract and arithmetic as conv3x3_halo_bfx4_kernel
// (mmdet/models/backbones/resnet.py:239-252 conv2, necks/fpn.py:129-141 fpn_convs, anchor_heads/rpn_head.py:30-35
// rpn_conv): fp32 NHWC in / out, every product from the exact three-way bf16 split of both operands, fp32 accumulate,
// bias / ReLU / the ReLU-backward mask of the data-gradient form in the epilogue.
//
// Why a second 3x3 kernel: it was written for the small maps — on the stride-16 / -32 maps (and the 128-channel stride-8
// layers) the halo kernel's 128-pixel x 128-channel workgroups number 70 - 570, so its plan slices K over gridDim.z to
// fill the chip (36 - 43 MB of partial sums per layer + a reduction launch), and each of its k steps ends in `vmcnt(0)`
// + a workgroup barrier: a lone workgroup takes 0.68 us per step where its MFMAs need 0.37
// (tools/halo_small_map_probe.py) — and turned out ahead of the halo kernels on EVERY map down to 50 x 84, the P2 level
// included (0.756 -> 0.685 ms: no filter hop through LDS, one barrier per 18 k steps instead of one per step).  Here:
//   * a workgroup owns 8 x 8 output pixels x 256 (or 128) channels and the WHOLE reduction: twice the pixel tiles, no
//     K slices, no partial sums, no second launch;
//   * per 32-channel chunk the 10 x 10 input patch is loaded ONCE (buffer loads; outside the image: an offset past the
//     descriptor, which reads 0), split ONCE into three bf16 planes in LDS ([plane][k step][patch row of 12 slots][32 B],
//     the 16-byte k halves swapped on odd patch rows: the A-fragment ds_read_b128 of every tap is conflict-free), double
//     buffered: ONE barrier per chunk = per 18 k steps (2 k steps x 9 taps);
//   * the MFMA phase is fragments + MFMAs: A fragments of tap (dy, dx) = the patch shifted by a compile-time offset,
//     filter fragments straight from L2 into registers one k step ahead (ping-pong register sets, unconditional
//     prefetches, the next chunk's patch loads issued BEHIND the first filter prefetch of the chunk, sched_barriers —
//     conv1x1_planes.hip explains each);
//   * epilogue through the LDS transpose in two halves of 32 pixels (4 rows of the tile), 16-byte stores.
// Order of accumulation = the halo kernel's WITHOUT K slices: 16-channel chunks ascending, the nine taps inside a chunk,
// the six plane products of a step in the ring kernel's order: BIT-IDENTICAL to conv3x3_halo_bfx4_kernel with one slice
// (tests/test_gpu_det_ops.py); against the sliced default it differs by the summation order of the slices (fp32
// rounding).  LDS 45 KB, three workgroups per CU.
#include <stdlib.h>

#ifndef BGS_P3_A_AUX
#define BGS_P3_A_AUX 0
#endif
#ifndef BGS_P3_Y_AUX
#define BGS_P3_Y_AUX 0
#endif
#ifndef BGS_P3_PRIO
#define BGS_P3_PRIO 0            // s_setprio 1 around the MFMA cluster of a k step (build variant p3prio): 2 % SLOWER on the P2 layer (0.735 vs 0.721 ms)
#endif
#ifndef BGS_P3_ABL
#define BGS_P3_ABL 0             // timing-only ablations (build variants p3abl1 / p3abl2 / p3abl3; results are WRONG)
#endif
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

struct Planes3Args {
  ConvArgs c;
  const __bf16* ws;      // split weights [3][KC][Cout][16], KC index = tap * (Cin / 16) + 16-channel chunk
  int KC;                // 9 * Cin / 16
  int tiles_y, tiles_x;  // 8 x 8 pixel tiles per image
};

constexpr int kOob = 0x7f000000;   // byte offset past every descriptor this kernel builds: the load returns 0

template <int NB>
__global__ __launch_bounds__(kThreads, 3) void conv3x3_planes_bfx_kernel(Planes3Args g) {
  const ConvArgs& p = g.c;
  constexpr int NS = 3, KCH = 2;                                    // k steps (16 channels) per 32-channel chunk
  constexpr int TW = 8, PW = 10, PS = 12, PSLOTS = PW * PS;        // 10 x 10 patch in rows of 12 slots
  constexpr int AQ = PW * PW * 8;                                   // 800 fp32 quads per chunk: 4 per thread (the 4th: 32 threads)
  constexpr int KS = PSLOTS * 32, PL = KCH * KS, BUF = NS * PL;     // 3840 B per k step, 7680 per plane, 23,040 per buffer
  constexpr int CO = 128 * NB, LD4 = CO + 8;
  constexpr int LDS_BYTES = 2 * BUF;                                // 46,080
  static_assert(PL % 512 == 0 && 3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");
  static_assert(32 * LD4 * 4 <= LDS_BYTES, "epilogue tile overlays the operand buffers");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int frow = lane & 31, fk = lane >> 5;
  const int vtile = (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3));
  if (vtile >= p.tiles_m * p.tiles_n) return;                       // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int per_img = g.tiles_y * g.tiles_x;
  const int n = tm / per_img, trem = tm - n * per_img;
  const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
  const int h0 = ty * 8 - 1, w0 = tx * 8 - 1;                       // input coordinates of patch (0, 0)
  const int n0 = tn * CO;
  const int cch16 = p.Cin >> 4, nchunks = p.Cin >> 5;

  // ---- patch loader: quad q = tid + 256 i: patch pixel q / 8, channels 4 (q % 8) .. of the chunk
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
  int a_off[4], a_dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + kThreads * i;
    const bool use = q < AQ;
    const int pix = use ? q >> 3 : 0, kq = q & 7;
    const int ppy = pix / PW, ppx = pix - ppy * PW;
    const int hi = h0 + ppy, wi = w0 + ppx;
    const bool in = use && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_off[i] = in ? (((n * p.H + hi) * p.W + wi) * p.Cin + kq * 4) * 4 : kOob;
    const int quad = kq & 3;
    a_dst[i] = use ? (kq >> 2) * KS + (ppy * PS + ppx) * 32 + (((quad >> 1) ^ (ppy & 1)) << 4) + (quad & 1) * 8 : -1;
  }
  f32x4 ra[4];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, a_off[i], chunk * 128, BGS_P3_A_AUX));
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a_dst[i] < 0) continue;
      u32x2 hh, mm, ll;
      split3p(ra[i], hh, mm, ll);
      unsigned char* d = lds + buf * BUF + a_dst[i];
      *reinterpret_cast<u32x2*>(d) = hh;
      *reinterpret_cast<u32x2*>(d + PL) = mm;
      *reinterpret_cast<u32x2*>(d + 2 * PL) = ll;
    }
  };

  // ---- filter fragments: buffer loads (descriptor in SGPRs, one 32-bit lane offset, the per-load constant as scalar offset)
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws), 0, (int)((size_t)NS * g.KC * p.Cout * 32), 0x00020000);
  const int b_lane = ((n0 + wave * 32 * NB + frow) * 16 + fk * 8) * 2;   // bytes
  const int b_plane = g.KC * p.Cout * 32;                                // bytes per plane
  auto load_b = [&](int kc, bf16x8 (&dst)[NS][NB]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        dst[s][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   b_rsrc, b_lane, s * b_plane + (kc * p.Cout + 32 * b) * 32, 0));
  };

  // ---- A fragments: lane frow of sub-tile a owns output pixel m = 32 a + frow = (m / 8, m % 8) of the tile; tap (dy, dx)
  //      reads patch pixel (m / 8 + dy, m % 8 + dx).  Two patch rows 12 slots = 384 B apart share a 128-byte half of the
  //      256-byte bank row; the k halves swap on odd rows, so every 16-lane group of the ds_read_b128 covers the 16
  //      16-byte slots exactly once, for every tap (a shift by dx slots or dy rows moves all lanes alike).
  int a_frag[2][2];                                                 // [sub-tile][parity of dy]
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = a * 32 + frow;
    const int py = m >> 3, px = m & 7;
#pragma unroll
    for (int par = 0; par < 2; ++par) a_frag[a][par] = (py * PS + px) * 32 + ((fk ^ ((py + par) & 1)) << 4);
  }
  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  bf16x8 fb0[NS][NB], fb1[NS][NB];
  load_a(0);
  load_b(0, fb0);                                                   // (tap 0, 16-channel chunk 0)
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    store_a(buf);                                                   // (waits for this chunk's patch loads)
    // one barrier per chunk: buffer `buf` was last read in chunk - 2, and every wave has passed the barrier of chunk - 1
    // (behind its chunk - 2 reads) before any wave writes it again
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2 * 9; ++t) {                               // k step t of the chunk: 16-channel half t / 9, tap t % 9
      const int kcs = t / 9, tap = t % 9, dy = tap / 3, dx = tap % 3;
      const int t1 = t + 1;
      const int kc_next = t1 < 18 ? (t1 % 9) * cch16 + chunk * 2 + t1 / 9 : (chunk + 1) * 2;
#if BGS_P3_ABL & 1                                               // timing only: filter fragments loaded twice per chunk
      if (t < 2) {
#endif
      if (t & 1) load_b(kc_next, fb0);
      else load_b(kc_next, fb1);
#if BGS_P3_ABL & 1
      }
#endif
#if BGS_P3_ABL & 2                                               // timing only: the patch loaded once
      if (t == 0 && chunk == 0) load_a(chunk + 1);
#else
      if (t == 0) load_a(chunk + 1);                                // in flight under this chunk's MFMAs (past the end: 0)
#endif
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 (&fbu)[NS][NB] = (t & 1) ? fb1 : fb0;
      bf16x8 fa[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + buf * BUF + s * PL + kcs * KS + a_frag[a][dy & 1] +
                                                      (dy * PS + dx) * 32);
#if BGS_P3_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
        for (int i = 0; i <= tt; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fbu[tt - i][b], acc[a][b], 0, 0, 0);
#if BGS_P3_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                  // every wave is done with the operand buffers

  // ---- epilogue: two halves of 32 pixels (4 tile rows) through the LDS transpose; bias, clamp, 16-byte stores
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int TPR = CO / 4, RPP = kThreads / TPR, EP = 32 / RPP;  // threads per pixel, pixels per pass, passes per half
  const int e4 = (tid % TPR) * 4, er0 = tid / TPR;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n0 + e4);
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
  // ReLU-backward mask of the data-gradient form (y = mask > 0 ? y : 0, [N,H,W,Cout] like y): prefetched per half
  const __amdgpu_buffer_rsrc_t mk_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.mask ? p.mask : p.x), 0, p.mask ? (int)((size_t)p.M * p.Cout * 4) : 0, 0x00020000);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    f32x4 mk[EP];
    if (p.mask) {
#pragma unroll
      for (int ps = 0; ps < EP; ++ps) {
        const int m = a * 32 + er0 + ps * RPP;
        const int ho = min(ty * 8 + (m >> 3), p.H - 1), wo = min(tx * TW + (m & 7), p.W - 1);
        mk[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               mk_rsrc, (((n * p.H + ho) * p.W + wo) * p.Cout + n0 + e4) * 4, 0, 0));
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[i * LD4 + wave * 32 * NB + b * 32 + (lane & 31)] = acc[a][b][r];
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < EP; ++ps) {
      const int i = er0 + ps * RPP;
      const int m = a * 32 + i;
      const int ho = ty * 8 + (m >> 3), wo = tx * TW + (m & 7);
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD4 + e4);
      v += bias;
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      if (p.mask) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = mk[ps][t] > 0.f ? v[t] : 0.f;
      }
      if (ho < p.H && wo < p.W)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc,
                                               (((n * p.H + ho) * p.W + wo) * p.Cout + n0 + e4) * 4, 0, BGS_P3_Y_AUX);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The stride-2 sibling (the three `layerN.0.conv2` launches of a ResNet: mmdet/models/backbones/resnet.py:239-252 with
// conv2_stride = 2; on the 64 x 64 operand ring before): 8 x 8 OUTPUT pixels x 128 / 256 channels per workgroup, the
// whole reduction in the workgroup.  The 17 x 17 input patch of a 16-channel chunk is stored as FOUR parity sub-grids
// ((row & 1, column & 1): 9 rows of 12 slots each), so that tap (dy, dx) reads sub-grid (dy & 1, dx & 1) at unit stride
// — the access pattern, and with it the conflict-free layout (k halves swapped on odd sub-grid rows), of the stride-1
// kernel.  One chunk = 9 k steps (the taps) = 41.5 KB of planes: single-buffered (three workgroups per CU), the next
// chunk's patch travels in registers under the MFMAs, two barriers per chunk.  Accumulation order: 16-channel chunks
// ascending, nine taps inside a chunk (the halo kernels' order; the operand ring sums tap-major — same products,
// another fp32 summation order).
template <int NB>
__global__ __launch_bounds__(kThreads, 3) void conv3x3s2_planes_bfx_kernel(Planes3Args g) {
  const ConvArgs& p = g.c;
  constexpr int NS = 3;
  constexpr int PW = 17, PS = 12, SG = 9 * PS * 32;                 // 17 x 17 patch; a sub-grid: 9 rows x 12 slots x 32 B = 3456 B
  constexpr int AQ = PW * PW * 4, AQT = (AQ + kThreads - 1) / kThreads;   // 1156 fp32 quads per chunk: 5 per thread
  constexpr int PL = 4 * SG, BUF = NS * PL;                         // 13,824 B per plane, 41,472 per chunk
  constexpr int CO = 128 * NB, LD4 = CO + 8;
  static_assert(3 * BUF <= 160 * 1024, "three workgroups per CU");
  static_assert(32 * LD4 * 4 <= BUF, "epilogue tile overlays the operand buffer");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int frow = lane & 31, fk = lane >> 5;
  const int vtile = (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3));
  if (vtile >= p.tiles_m * p.tiles_n) return;                       // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int per_img = g.tiles_y * g.tiles_x;
  const int n = tm / per_img, trem = tm - n * per_img;
  const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
  const int h0 = ty * 16 - 1, w0 = tx * 16 - 1;                     // input coordinates of patch (0, 0)
  const int n0 = tn * CO;
  const int cch16 = p.Cin >> 4;

  // ---- patch loader: quad q = tid + 256 i: patch pixel q / 4, channels 4 (q % 4) .. of the chunk
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
  int a_off[AQT], a_dst[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int q = tid + kThreads * i;
    const bool use = q < AQ;
    const int pix = use ? q >> 2 : 0, quad = q & 3;
    const int ppy = pix / PW, ppx = pix - ppy * PW;
    const int hi = h0 + ppy, wi = w0 + ppx;
    const bool in = use && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_off[i] = in ? (((n * p.H + hi) * p.W + wi) * p.Cin + quad * 4) * 4 : kOob;
    const int iy = ppy >> 1, ix = ppx >> 1;
    a_dst[i] = use ? ((ppy & 1) * 2 + (ppx & 1)) * SG + (iy * PS + ix) * 32 + (((quad >> 1) ^ (iy & 1)) << 4) + (quad & 1) * 8 : -1;
  }
  f32x4 ra[AQT];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i)
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, a_off[i], chunk * 64, 0));
  };
  auto store_a = [&]() {
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
  };

  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws), 0, (int)((size_t)NS * g.KC * p.Cout * 32), 0x00020000);
  const int b_lane = ((n0 + wave * 32 * NB + frow) * 16 + fk * 8) * 2;   // bytes
  const int b_plane = g.KC * p.Cout * 32;                                // bytes per plane
  auto load_b = [&](int kc, bf16x8 (&dst)[NS][NB]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        dst[s][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   b_rsrc, b_lane, s * b_plane + (kc * p.Cout + 32 * b) * 32, 0));
  };

  // ---- A fragments: output pixel m = 32 a + frow = (py, px); tap (dy, dx) reads patch pixel (2 py + dy, 2 px + dx) =
  //      sub-grid (dy & 1, dx & 1), row py + (dy >> 1), column px + (dx >> 1)
  int a_frag[2][2];                                                 // [sub-tile][dy >> 1]
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = a * 32 + frow;
    const int py = m >> 3, px = m & 7;
#pragma unroll
    for (int par = 0; par < 2; ++par) a_frag[a][par] = (py * PS + px) * 32 + ((fk ^ ((py + par) & 1)) << 4);
  }
  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  bf16x8 fb0[NS][NB], fb1[NS][NB];
  load_a(0);
  load_b(0, fb0);                                                   // (tap 0, chunk 0)
  for (int c2 = 0; c2 < cch16; c2 += 2) {                           // two chunks per trip: 18 k steps, the ping-pong closes
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      const int chunk = c2 + t / 9, tap = t % 9, dy = tap / 3, dx = tap % 3;
      if (tap == 0) {
        __syncthreads();                                            // every wave is done with the previous chunk's planes
        store_a();                                                  // (waits for this chunk's patch loads)
        __syncthreads();
      }
      const int kc_next = tap < 8 ? (tap + 1) * cch16 + chunk : chunk + 1;
      if (t & 1) load_b(kc_next, fb0);
      else load_b(kc_next, fb1);
      if (tap == 0) load_a(chunk + 1);                              // in flight under this chunk's MFMAs (past the end: unused)
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 (&fbu)[NS][NB] = (t & 1) ? fb1 : fb0;
      bf16x8 fa[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + s * PL + ((dy & 1) * 2 + (dx & 1)) * SG + a_frag[a][dy >> 1] +
                                                      ((dy >> 1) * PS + (dx >> 1)) * 32);
#pragma unroll
      for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
        for (int i = 0; i <= tt; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fbu[tt - i][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                  // every wave is done with the operand buffer

  // ---- epilogue: two halves of 32 output pixels through the LDS transpose; bias, clamp, 16-byte stores
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int TPR = CO / 4, RPP = kThreads / TPR, EP = 32 / RPP;
  const int e4 = (tid % TPR) * 4, er0 = tid / TPR;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n0 + e4);
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[i * LD4 + wave * 32 * NB + b * 32 + (lane & 31)] = acc[a][b][r];
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < EP; ++ps) {
      const int i = er0 + ps * RPP;
      const int m = a * 32 + i;
      const int ho = ty * 8 + (m >> 3), wo = tx * 8 + (m & 7);
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD4 + e4);
      v += bias;
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      if (ho < p.Ho && wo < p.Wo)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc,
                                               (((n * p.Ho + ho) * p.Wo + wo) * p.Cout + n0 + e4) * 4, 0, 0);
    }
    __syncthreads();
  }
}

int g_planes3_mode = -1;      // BGS_BFX_PLANES3 / bgs_conv3x3_planes_enable: 0 off | 1 automatic | 2 every eligible layer
int g_planes3_last = 0;

}  // namespace

extern "C" void bgs_conv3x3_planes_enable(int mode) { g_planes3_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode); }
extern "C" int bgs_conv3x3_planes_last_launch(void) { return g_planes3_last; }
void bgs_internal_conv3x3_planes_clear_last() { g_planes3_last = 0; }

// -1: not eligible / declined (the caller goes on to the halo kernels)
int bgs_internal_conv3x3_planes(const bgs_conv::ConvArgs& pc, const void* wsplit, int KC, hipStream_t st) {
  int mode = g_planes3_mode;                // the hook's value, else the environment (read at every call: tools/step_ab.py)
  if (mode < 0) {
    const char* e = getenv("BGS_BFX_PLANES3");
    mode = e ? atoi(e) : 1;
    if (mode < 0 || mode > 2) mode = 1;
  }
  if (mode == 0) return -1;
  const ConvArgs& p = pc;
  if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.rowmap || p.res_mode != 0) return -1;
  if ((p.Cin & 31) || (p.Cout & 127) || KC != 9 * (p.Cin / 16)) return -1;
  if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.bias | (uintptr_t)p.mask | (uintptr_t)wsplit) & 15) return -1;
  const long long lim = kOob;
  if ((long long)p.N * p.H * p.W * p.Cin * 4 >= lim || (long long)p.M * p.Cout * 4 >= lim || (long long)3 * KC * p.Cout * 32 >= lim) return -1;
  const int tiles_y = (p.H + 7) / 8, tiles_x = (p.W + 7) / 8;
  const long long tiles_px = (long long)p.N * tiles_y * tiles_x;
  static int nb_env = -1;
  if (nb_env < 0) {
    const char* e = getenv("BGS_BFX_PLANES3_NB");
    nb_env = e ? atoi(e) : 0;
  }
  int nb = (p.Cout & 255) ? 1 : 2;
  if (nb == 2 && (nb_env == 1 || (nb_env == 0 && tiles_px * (p.Cout / 256) <= 768))) nb = 1;
  const long long wgs = tiles_px * (p.Cout / (128 * nb));
  if (mode == 1) {
    // automatic: wherever this kernel's grid has a workgroup per CU (profiles/r11i_planes3_ab.txt, every 3x3 / stride-1
    // layer of a cfg[1] step, interleaved, against the default halo dispatch): P2 level 756 -> 685 us, P3 level 226 ->
    // 195, layer2 conv2 84 -> 63, layer3 conv2 80 -> 72, stride-16 level 77 -> 70.  Below that (the 25 x 42 maps: 96 - 192
    // workgroups) the sliced halo plan stays ahead: layer4 conv2 80 vs 83, stride-32 level 32 vs 37.
    if (wgs < 256) return -1;
  }
  Planes3Args g;
  g.c = p;
  g.ws = reinterpret_cast<const __bf16*>(wsplit);
  g.KC = KC;
  g.tiles_y = tiles_y;
  g.tiles_x = tiles_x;
  g.c.tiles_m = (int)tiles_px;
  g.c.tiles_n = p.Cout / (128 * nb);
  g.c.chunk = (int)((wgs + 7) / 8);
  g.c.partial = nullptr;
  const dim3 grid((unsigned)(8 * g.c.chunk)), block(kThreads);
  if (nb == 2) hipLaunchKernelGGL((conv3x3_planes_bfx_kernel<2>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((conv3x3_planes_bfx_kernel<1>), grid, block, 0, st, g);
  g_planes3_last = nb;
  bgs_internal_census_bump(BGS_CENSUS_PLANES_3X3);
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}

// the stride-2 form (forward 3x3 / stride 2 / pad 1 layers; called from launch_conv_bfx ahead of the operand ring)
int bgs_internal_conv3x3s2_planes(const bgs_conv::ConvArgs& pc, const void* wsplit, int KC, hipStream_t st) {
  int mode = g_planes3_mode;
  if (mode < 0) {
    const char* e = getenv("BGS_BFX_PLANES3");
    mode = e ? atoi(e) : 1;
    if (mode < 0 || mode > 2) mode = 1;
  }
  static int s2_env = -1;                   // BGS_BFX_PLANES3_S2=0: the stride-2 form alone off (A/B)
  if (s2_env < 0) {
    const char* e = getenv("BGS_BFX_PLANES3_S2");
    s2_env = e ? atoi(e) : 1;
  }
  if (mode == 0 || !s2_env) return -1;
  const ConvArgs& p = pc;
  if (p.R != 3 || p.S != 3 || p.stride != 2 || p.pad != 1 || p.mask || p.rowmap || p.res_mode != 0) return -1;
  if ((p.Cin & 31) || (p.Cout & 127) || KC != 9 * (p.Cin / 16)) return -1;
  if (p.Ho != (p.H - 1) / 2 + 1 || p.Wo != (p.W - 1) / 2 + 1) return -1;
  if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.bias | (uintptr_t)wsplit) & 15) return -1;
  const long long lim = kOob;
  if ((long long)p.N * p.H * p.W * p.Cin * 4 >= lim || (long long)p.M * p.Cout * 4 >= lim || (long long)3 * KC * p.Cout * 32 >= lim) return -1;
  const int tiles_y = (p.Ho + 7) / 8, tiles_x = (p.Wo + 7) / 8;
  const long long tiles_px = (long long)p.N * tiles_y * tiles_x;
  const int nb = 1;                         // (128 channels per workgroup: with 256 the five patch quads in flight per thread spill)
  const long long wgs = tiles_px * (p.Cout / (128 * nb));
  if (mode == 1 && wgs < 256) return -1;      // (layer4.0 conv2, 192 workgroups: 93.2 vs 93.5 us on the ring — left there)
  Planes3Args g;
  g.c = p;
  g.ws = reinterpret_cast<const __bf16*>(wsplit);
  g.KC = KC;
  g.tiles_y = tiles_y;
  g.tiles_x = tiles_x;
  g.c.tiles_m = (int)tiles_px;
  g.c.tiles_n = p.Cout / (128 * nb);
  g.c.chunk = (int)((wgs + 7) / 8);
  g.c.partial = nullptr;
  const dim3 grid((unsigned)(8 * g.c.chunk)), block(kThreads);
  hipLaunchKernelGGL((conv3x3s2_planes_bfx_kernel<1>), grid, block, 0, st, g);
  g_planes3_last = nb | 0x10;               // bit 4: the stride-2 form
  bgs_internal_census_bump(BGS_CENSUS_PLANES_3X3);
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}
