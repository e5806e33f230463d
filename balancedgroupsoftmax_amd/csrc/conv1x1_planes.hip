// 1x1 convolution / linear layer on the bf16 matrix cores with the A-operand split taken OUT of the MFMA loop
// (round 6).  Same contract and arithmetic as the 1x1 instantiations of conv_bfx.hip
// (mmdet/models/backbones/resnet.py:220-266 conv1 / conv3 / downsample, necks/fpn.py:118-127 lateral convs): fp32 NHWC in,
// fp32 out, every product from the exact three-way bf16 split of both operands, fp32 accumulate, bias / residual / ReLU.
//
// What the existing kernels do: the 64 x 64 operand ring and the 128 x 128 wide kernel deliver A as fp32 and split it in
// registers when a wave reads its fragment — 44 VALU instructions beside 6 (ring) or 24 (wide) MFMAs, repeated by every
// workgroup that covers the same pixels for another slice of the output channels (2 - 16 of them), and VALU and MFMA time
// add on a SIMD (profiles/r9c).  Here a workgroup owns 64 pixels x 256 output channels:
//   * per 64-deep K chunk the [64 pixels x 64 k] fp32 tile is loaded ONCE (16-byte buffer loads, the next chunk in flight
//     under this chunk's MFMAs), split ONCE (4 split3 per thread) and written to LDS as three bf16 planes in the B
//     layout ([plane][k step][pixel][32 B], halves swapped on odd 8-pixel groups: conflict-free ds_read_b128), double
//     buffered, ONE barrier per chunk;
//   * the MFMA phase of a chunk is fragments + MFMAs only: wave w owns all 64 pixels x channels 64 w .. 64 w + 63
//     (acc[2][2]), A fragments from LDS, the filter fragments of a k step straight from L2 into registers by buffer loads
//     (one k step ahead); the filter never passes through LDS;
//   * epilogue through the LDS transpose in two quarters of 32 pixels: bias, then the residual (same shape, or the
//     nearest-2x-upsampled coarser map of the FPN top-down path), then the clamp — 16-byte loads / stores.
// k steps ascend and the six plane products of a step come in the ring kernel's order, so every output is BIT-IDENTICAL to
// conv_igemm_bfx_dma_kernel / conv1x1_bfx_wide_kernel (tests/test_gpu_det_ops.py).  LDS 51 KB, <= 168 VGPRs: three
// workgroups per CU.  Eligible: 1x1 / stride 1 or 2 / no padding, Cin % 64 == 0, Cout % 256 == 0, no split-K, no ReLU mask,
// tensors below 2 GB (32-bit buffer offsets).
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

struct PlanesArgs {
  ConvArgs c;
  const __bf16* ws;      // split weights [3][KC][Cout][16]
  int KC;                // K / 16
};

// ABL: timing-only ablations (tools/planes_ablate.py; results are WRONG for ABL != 0): 1 = filter fragments loaded once,
// 2 = activation tile loaded once, 4 = no output stores, 8 = no MFMAs
// NB: 32-channel fragments per wave: 2 = 256 output channels per workgroup, 1 = 128 (the 128-channel layers, and the
// 256-channel layers on the small maps whose 64-pixel tiles alone leave half the CUs without a workgroup)
template <int ABL, int NB>
__global__ __launch_bounds__(kThreads, 3) void conv1x1_planes_bfx_kernel(PlanesArgs g) {
  const ConvArgs& p = g.c;
  constexpr int NS = 3, KCH = 4;                                    // k steps per 64-deep chunk
  // 2 KB per k step + 64 B of padding: the 16 lanes that write one pixel row's 64 k cover FOUR k steps, and a stride of
  // 2048 B would put the four on the same 8 banks (measured: SQ_LDS_BANK_CONFLICT = 43 % of SQ_LDS_IDX_ACTIVE); planes
  // padded to a multiple of 512 B (ds_write2st64_b64 pairs)
  constexpr int CH = 64 * 32 + 64, PL = KCH * CH + 256, BUF = NS * PL;
  constexpr int CO = 128 * NB, LD4 = CO + 8;      // (4 rows apart = 32 banks: the two half-waves of the transpose write never share a bank)
  constexpr int LDS_BYTES = 2 * BUF;                                // 52,224 (>= the 33,280-byte quarter tile)
  static_assert(PL % 512 == 0 && 3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");
  static_assert(32 * LD4 * 4 <= LDS_BYTES, "epilogue tile overlays the operand buffers");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int frow = lane & 31, fk = lane >> 5;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;                       // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int m0 = tm * 64, n0 = tn * CO;
  const int nchunks = p.Cin >> 6;

  // ---- A loader: 16 threads per pixel row (64 k = 16 quads), 16 rows per pass, 4 passes; rows past M re-read row M - 1
  const int c4 = (tid & 15) * 4, r0 = tid >> 4;
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
  int a_off[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int m = min(m0 + r0 + 16 * ps, p.M - 1);
    int src = m;                                                    // input pixel of output pixel m
    if (p.stride != 1) {                                            // (projection shortcuts: 1x1 / stride 2)
      const int hw_o = p.Ho * p.Wo;
      const int n = m / hw_o, rem = m - n * hw_o;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      src = (n * p.H + ho * p.stride) * p.W + wo * p.stride;
    }
    a_off[ps] = (src * p.Cin + c4) * 4;
  }
  f32x4 ra[4];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
      ra[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, a_off[ps], chunk * 256, 0));
  };
  const int kcs_w = c4 >> 4, kq_w = (c4 & 15) >> 2;
  auto store_a = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int i = r0 + 16 * ps;
      u32x2 hh, mm, ll;
      split3p(ra[ps], hh, mm, ll);
      unsigned char* d = lds + buf * BUF + kcs_w * CH + i * 32 + (((kq_w >> 1) ^ ((i >> 3) & 1)) << 4) + (kq_w & 1) * 8;
      *reinterpret_cast<u32x2*>(d) = hh;
      *reinterpret_cast<u32x2*>(d + PL) = mm;
      *reinterpret_cast<u32x2*>(d + 2 * PL) = ll;
    }
  };

  // ---- B fragments: buffer loads (descriptor in SGPRs, one 32-bit lane offset, the per-load constant as scalar offset)
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws), 0, (int)((size_t)NS * g.KC * p.Cout * 32), 0x00020000);
  const int b_lane = ((n0 + wave * 32 * NB + frow) * 16 + fk * 8) * 2;   // bytes
  const int b_plane = g.KC * p.Cout * 32;                           // bytes per plane
  auto load_b = [&](int kc, bf16x8 (&dst)[NS][NB]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        dst[s][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   b_rsrc, b_lane, s * b_plane + (kc * p.Cout + 32 * b) * 32, 0));
  };

  int a_frag[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = a * 32 + frow;
    a_frag[a] = m * 32 + ((fk ^ ((m >> 3) & 1)) << 4);
  }
  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // Loop shape: every prefetch is UNCONDITIONAL (a buffer load past the end of the tensor returns 0 and is never used), so
  // a chunk is one basic block; the filter fragments ping-pong between two register sets (no copies); the next chunk's
  // activation loads are issued BEHIND the first filter prefetch of the chunk — vmcnt retires in order, so a wait for
  // filter fragments also waits for every load issued before them: this order gives the HBM loads two k steps of cover.
  bf16x8 fb0[NS][NB], fb1[NS][NB];
  load_a(0);
  load_b(0, fb0);
  auto kstep = [&](int buf, int kcs, const bf16x8 (&fbu)[NS][NB]) {
    bf16x8 fa[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + buf * BUF + a_frag[a] + kcs * CH + s * PL);
#pragma unroll
    for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
      for (int i = 0; i <= tt; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (!(ABL & 8)) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fbu[tt - i][b], acc[a][b], 0, 0, 0);
    if (ABL & 8) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          asm volatile("" ::"v"(fa[s][a]));
          asm volatile("" ::"v"(fbu[s][a % NB]));
        }
    }
  };
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    const int kc = chunk * KCH;
    store_a(buf);                                                   // (waits for this chunk's A loads)
    // one barrier per chunk: buffer `buf` was last read in chunk - 2, and every wave has passed the barrier of chunk - 1
    // (behind its chunk - 2 reads) before any wave writes it again
    __syncthreads();
    // (sched_barrier: the scheduler otherwise sinks every prefetch to just above its first use to save registers)
    if (!(ABL & 1)) load_b(kc + 1, fb1);
    if (!(ABL & 2)) load_a(chunk + 1);                              // in flight under this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    kstep(buf, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 1)) load_b(kc + 2, fb0);
    __builtin_amdgcn_sched_barrier(0);
    kstep(buf, 1, (ABL & 1) ? fb0 : fb1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 1)) load_b(kc + 3, fb1);
    __builtin_amdgcn_sched_barrier(0);
    kstep(buf, 2, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 1)) load_b(kc + 4, fb0);
    __builtin_amdgcn_sched_barrier(0);
    kstep(buf, 3, (ABL & 1) ? fb0 : fb1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                                  // every wave is done with the operand buffers

  // ---- epilogue: two quarters of 32 pixels through the LDS transpose; bias, then residual, then clamp
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int TPR = CO / 4, RPP = kThreads / TPR, EP = 32 / RPP;  // threads per pixel row, rows per pass, passes per quarter
  const int e4 = (tid % TPR) * 4, er0 = tid / TPR;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n0 + e4);
  const int hw = p.Ho * p.Wo;
  size_t res_bytes = 0;
  if (p.res_mode == 1) res_bytes = (size_t)p.M * p.Cout * 4;
  else if (p.res_mode == 2) res_bytes = (size_t)p.N * (p.Ho >> 1) * (p.Wo >> 1) * p.Cout * 4;
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res_mode ? p.res : p.x), 0, (int)res_bytes, 0x00020000);      // 0 bytes: every read is 0
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
  // ReLU-backward mask of the data-gradient form (y = mask > 0 ? y : 0, [M, Cout] like y)
  const __amdgpu_buffer_rsrc_t mk_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.mask ? p.mask : p.x), 0, p.mask ? (int)((size_t)p.M * p.Cout * 4) : 0, 0x00020000);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    f32x4 rs[EP];
#pragma unroll
    for (int ps = 0; ps < EP; ++ps) {
      const int m = min(m0 + a * 32 + er0 + ps * RPP, p.M - 1);
      int row = m;
      if (p.res_mode == 2) {
        const int n = m / hw;
        const int rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        row = (n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1);
      }
      rs[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, (row * p.Cout + n0 + e4) * 4, 0, 0));
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
      const int m = m0 + a * 32 + i;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD4 + e4);
      v += bias;
      if (p.res_mode) v += rs[ps];
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      if (p.mask) {      // (loaded here: the residual prefetch already holds EP registers quads across the transpose)
        const f32x4 mk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       mk_rsrc, (min(m, p.M - 1) * p.Cout + n0 + e4) * 4, 0, 0));
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
      }
      if (m < p.M && (!(ABL & 4) || v[0] == 1.2345e-30f))
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, (m * p.Cout + n0 + e4) * 4, 0, 0);
    }
    __syncthreads();
  }
}

int g_planes_mode = -1;       // BGS_BFX_PLANES / bgs_conv1x1_planes_enable: 0 off | 1 automatic | 2 every eligible layer
int g_planes_last = 0;

}  // namespace

extern "C" void bgs_conv1x1_planes_enable(int mode) { g_planes_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode); }
extern "C" int bgs_conv1x1_planes_last_launch(void) { return g_planes_last; }
void bgs_internal_conv1x1_planes_clear_last() { g_planes_last = 0; }

// -1: not eligible (the caller goes on to the wide / ring kernels)
int bgs_internal_conv1x1_planes(const bgs_conv::ConvArgs& pc, const void* wsplit, int KC, hipStream_t st) {
  int mode = g_planes_mode;                 // the hook's value, else the environment (read at every call: tools/step_ab.py)
  if (mode < 0) {
    const char* e = getenv("BGS_BFX_PLANES");
    mode = e ? atoi(e) : 1;
    if (mode < 0 || mode > 2) mode = 1;
  }
  if (mode == 0) return -1;
  const ConvArgs& p = pc;
  if (p.R != 1 || p.S != 1 || (p.stride != 1 && p.stride != 2) || p.pad != 0 || p.rowmap) return -1;      // (p.partial: set by the caller's plan later)
  if ((p.Cin & 63) || (p.Cout & 127) || KC != p.Cin / 16) return -1;
  if (p.res_mode < 0 || p.res_mode > 2 || (p.res_mode && !p.res)) return -1;
  if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.mask | (uintptr_t)wsplit) & 15) return -1;
  const long long lim = 0x7fffffffLL;
  if ((long long)p.N * p.H * p.W * p.Cin * 4 > lim || (long long)p.M * p.Cout * 4 > lim || (long long)3 * KC * p.Cout * 32 > lim) return -1;
  // channels per workgroup: 256 on the large grids; 128 on the 128-channel layers and where the 64-pixel tiles x
  // 256-channel slabs do not fill one round of 3 workgroups per CU (or barely two with a short K): half-size workgroups
  // then pack better (profiles/r11d_planes_channels_per_workgroup.txt: l3.c3 35.4 -> 33.0 us, fpn.lat1 59.8 -> 57.4,
  // l4.c3 38.8 -> 33.4, l2.c3 39.4 -> 37.1; the other way on the 2100-tile grids: l1.c3 71.1 vs 75.7, fpn.lat0 114 vs
  // 127).  BGS_BFX_PLANES_NB = 1 / 2 forces one (A/B).
  static int nb_env = -1;
  if (nb_env < 0) {
    const char* e = getenv("BGS_BFX_PLANES_NB");
    nb_env = e ? atoi(e) : 0;
  }
  const int tiles_m = (p.M + 63) / 64;
  int nb = (p.Cout & 255) ? 1 : 2;
  if (nb == 2) {
    const long long tiles2 = (long long)tiles_m * (p.Cout / 256);
    if (nb_env == 1 || (nb_env == 0 && (tiles2 <= 768 || (tiles2 <= 1100 && p.Cin <= 128)))) nb = 1;
  }
  if (mode == 1) {
    // automatic: where it was measured ahead of the default dispatch (profiles/r11d, every eligible 1x1 layer of a
    // cfg[1] step, interleaved): 5 - 20 % wherever the grid has at least one workgroup per CU.  Below that the default
    // plan slices K over gridDim.z to fill the chip and this kernel (no split-K) loses: l4.c1 (132 workgroups) 49.9 vs
    // 44.7 us, fpn.lat3 (66) 47.9 vs 28.4.
    if ((long long)tiles_m * (p.Cout / (128 * nb)) < 256) return -1;
  }
  PlanesArgs g;
  g.c = p;
  g.ws = reinterpret_cast<const __bf16*>(wsplit);
  g.KC = KC;
  g.c.tiles_m = tiles_m;
  g.c.tiles_n = p.Cout / (128 * nb);
  g.c.chunk = (g.c.tiles_m * g.c.tiles_n + 7) / 8;
  g.c.partial = nullptr;
  static int abl = -1;                      // BGS_BFX_PLANES_ABLATE: timing-only, see the kernel's ABL
  if (abl < 0) {
    const char* e = getenv("BGS_BFX_PLANES_ABLATE");
    abl = e ? atoi(e) : 0;
  }
  const dim3 grid((unsigned)(8 * g.c.chunk)), block(kThreads);
#define BGS_PL(A_) \
  do { \
    if (nb == 2) hipLaunchKernelGGL((conv1x1_planes_bfx_kernel<A_, 2>), grid, block, 0, st, g); \
    else hipLaunchKernelGGL((conv1x1_planes_bfx_kernel<A_, 1>), grid, block, 0, st, g); \
  } while (0)
  switch (abl) {
    case 1: BGS_PL(1); break;
    case 2: BGS_PL(2); break;
    case 3: BGS_PL(3); break;
    case 4: BGS_PL(4); break;
    case 7: BGS_PL(7); break;
    case 8: BGS_PL(8); break;
    case 12: BGS_PL(12); break;
    case 15: BGS_PL(15); break;
    default: BGS_PL(0); break;
  }
#undef BGS_PL
  g_planes_last = nb;
  bgs_internal_census_bump(BGS_CENSUS_PLANES_1X1);
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}
