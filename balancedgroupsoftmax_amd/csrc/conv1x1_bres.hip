// 1x1 convolution with the FILTER RESIDENT IN REGISTERS and the activations read ONCE (gfx950,
// bf16x6 arithmetic of conv_bfx.hip: fp32 tensors, every product from the exact three-way bf16 split
// of both operands, fp32 accumulate — bit-identical results to conv_igemm_bfx_dma_kernel).
//
// Layers: the 1x1 convolutions of the ResNet bottlenecks, the projection shortcuts and the FPN
// laterals (mmdet/models/backbones/resnet.py:220-266, necks/fpn.py:101-141) with a short reduction,
// K = Cin in {64, 128, 256}, and Cout a multiple of 256 — fpn.lat0 (256 -> 256 on the 200x336 map),
// layer1 conv3 / shortcut (64 -> 256), layer2 shortcut (256 -> 512, stride 2) and conv3 (128 -> 512),
// layer3 conv3 (256 -> 1024).
//
// Why (profiles/r2z_pmc_ring_kernel.md): the 64 x 64 operand ring moves 4 KB of A
// and 6 KB of B through the CU's vector-memory path per 24 MFMAs, fetches every activation row
// Cout / 64 times and the split filter once per pixel tile — fpn.lat0: ~1.5 GB through L1/TA for
// 310 MB of algorithmic traffic, texture path 80 % busy, matrix pipe 0.30.  With K <= 256 the split
// filter slice of 32 output channels is 16 K steps x 3 planes x 4 VGPRs = 192 registers: a workgroup
// of FOUR waves holds a 256-channel slab of the filter in its register files for its whole life
// and streams pixel tiles past it:
//   * persistent workgroups (one per CU and column slab), each loads its filter fragments ONCE and
//     loops over its share of the 32-pixel tiles;
//   * a pixel tile is 32 rows x K fp32 (<= 32 KB), DMA'd global -> LDS (`global_load_lds_dwordx4`,
//     no VGPR staging) into a ring of three slots, two tiles in flight; every wave reads the SAME A
//     fragments (bank-conflict-free through an XOR swizzle of the 16-byte quads applied to the DMA
//     source addresses), splits them into the three bf16 planes and issues 6 MFMAs per K step
//     against each of its two resident B column blocks: 192 MFMAs per wave between two barriers
//     instead of 6;
//   * through the vector-memory path: A once (x Cout / 256 column slabs, from L2), the filter once
//     per workgroup — fpn.lat0: ~0.2 GB instead of ~1.5 GB; the layer becomes HBM-bound;
//   * epilogue (bias, residual same-shape or nearest-2x-upsampled, ReLU, ReLU-backward mask) through
//     an LDS transpose: 16-byte residual loads and stores.
#include <stdlib.h>

#include "conv_args.h"
#include "bfx_split.h"

using namespace bgs_conv;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) {
  return __builtin_bit_cast(float, u & 0xffff0000u);
}
// x (4 consecutive k) -> three planes of 4 packed bf16 each (conv_bfx.hip's split3)
__device__ __forceinline__ void split3(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
  const f32x4 r = {bfx_resid_lo(hi[0], v[0]), bfx_resid_hi(hi[0], v[1]), bfx_resid_lo(hi[1], v[2]),
                   bfx_resid_hi(hi[1], v[3])};
  mid = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
  const f32x4 r2 = {bfx_resid_lo(mid[0], r[0]), bfx_resid_hi(mid[0], r[1]), bfx_resid_lo(mid[1], r[2]),
                    bfx_resid_hi(mid[1], r[3])};
  lo = u32x2{pack_bf16(r2[0], r2[1]), pack_bf16(r2[2], r2[3])};
}
__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct BresArgs {
  const unsigned* zero;  // device zero page: DMA source of rows past M
  ConvArgs c;            // x, bias, res, mask, y, H, W, Cin, Ho, Wo, Cout, stride, M, relu, res_mode
  const __bf16* ws;      // split weights [3][KC][Cout][16]
  int KC;
  int tiles_m;           // ceil(M / 32)
  int ncg;               // column slabs of BN channels
  int wg_per_cg;         // workgroups per slab (grid = ncg * wg_per_cg)
};

constexpr int kWaves8 = 4, kThreads8 = 256, kBM = 32, kNST = 3;   // (names kept: the workgroup WAS 8 waves)

// KS = K / 16 (4, 8, 16); NCB = 32-column blocks per wave (BN = 128 NCB = 256).
// First version: EIGHT waves x 32 columns (192 filter registers of the 256 a wave of an 8-wave
// workgroup can have): hipcc spilled 23-47 VGPRs, and every scratch reload inside the tile loop is a
// VMEM load whose s_waitcnt also drains the tile DMAs in flight — 10 us per tile instead of 2.6
// (profiles/r5b: SLOWER than the operand ring on every layer but K = 64).  Now FOUR waves x 64
// columns with the whole 512-register file of a SIMD per wave (one wave per SIMD): 384 filter
// registers for K = 256, no spill; the wave hides its own LDS / VALU latency under its 12 MFMAs per
// K step (the loop is fully unrolled: the next step's fragment reads and splits are independent
// of the current step's MFMAs).
// S2: stride 2 (the projection shortcuts); otherwise a tile is 32 consecutive rows of x
template <int KS, int NCB, bool S2>
__global__ __launch_bounds__(kThreads8, KS >= 8 ? 1 : 2) void conv1x1_bres_kernel(BresArgs q) {
  const ConvArgs& p = q.c;
  constexpr int BN = 128 * NCB;
  constexpr int KB = KS * 64;                  // bytes of a tile row (K fp32)
  constexpr int SLOT = kBM * KB;               // one A tile
  constexpr int QPR = KS * 4;                  // 16-byte quads per row (16 / 32 / 64)
  constexpr int RPP = 64 / QPR;                // rows per 1 KB DMA piece (4 / 2 / 1)
  constexpr int PPW = (kBM / RPP) / kWaves8;   // DMA pieces per wave and tile (2 / 4 / 8)
  constexpr int EPI = kBM * BN * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kNST * SLOT + EPI];
  float* scratch = reinterpret_cast<float*>(lds + kNST * SLOT);
  const unsigned* __restrict__ zero_page = q.zero;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int cg = blockIdx.x % q.ncg, slot0 = blockIdx.x / q.ncg;
  const int n0 = cg * BN;
  const int frow = lane & 31, fk = lane >> 5;

  // ---- resident filter fragments: rows n0 + 32 NCB wave + 32 b + frow, k half fk, all KS steps
  bf16x8 fb[KS][NCB][3];
  {
    const size_t plane = (size_t)q.KC * p.Cout * 16;
#pragma unroll
    for (int b = 0; b < NCB; ++b) {
      const int row = n0 + (wave * NCB + b) * 32 + frow;
      const bool ok = row < p.Cout;
      const __bf16* base = q.ws + (size_t)(ok ? row : 0) * 16 + fk * 8;
#pragma unroll
      for (int kc = 0; kc < KS; ++kc)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const __bf16* src = ok ? base + s * plane + (size_t)kc * p.Cout * 16
                                 : reinterpret_cast<const __bf16*>(zero_page);
          fb[kc][b][s] = *reinterpret_cast<const bf16x8*>(src);
        }
    }
  }

  // ---- A DMA roles: piece (wave * PPW + j) of a tile = rows (piece * RPP + lane / QPR), physical
  //      quad lane % QPR; the LOGICAL quad fetched is physical ^ (row & 15) (16-lane groups of a
  //      ds_read_b128 then hit 16 distinct 16-byte slots; rows are KB apart, a multiple of 256 B)
  //      (recomputed per issue: four VGPRs per piece are worth more than a few VALU ops here — the
  //      resident filter takes 192 of the 256 registers a wave of an 8-wave workgroup can have)
  const int hw = p.Ho * p.Wo;
  auto issue = [&](int tile, int stage) {
    unsigned char* st = lds + stage * SLOT;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int d_row = (wave * PPW + j) * RPP + lane / QPR;
      const int d_lq = (lane % QPR) ^ (d_row & 15);
      const int m = tile * kBM + d_row;
      const float* src = reinterpret_cast<const float*>(zero_page);
      if (m < p.M) {                       // (tile >= tiles_m implies m >= M)
        size_t pix = (size_t)m;
        if (S2) {
          const int n = m / hw;
          const int rem = m - n * hw;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          pix = ((size_t)n * p.H + (size_t)ho * 2) * p.W + (size_t)wo * 2;
        }
        src = p.x + pix * p.Cin + d_lq * 4;
      }
      glds16(src, st + (wave * PPW + j) * 1024);
    }
  };

  // ---- A fragment roles: row frow; logical quads 4 kc + 2 fk (+1) -> physical ^ (frow & 15)
  const int a_rowoff = frow * KB;
  const int a_x = frow & 15;

  // ---- epilogue roles: TPR threads per row, 16-byte column quads
  constexpr int TPR = BN / 4, RPPS = kThreads8 / TPR;
  const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
  const int jcol = n0 + c4;

  const int stride_t = q.wg_per_cg;
  issue(slot0, 0);
  issue(slot0 + stride_t, 1);
  int it = 0;
  for (int tile = slot0; tile < q.tiles_m; tile += stride_t, ++it) {
    // this wave's pieces of stage `it` have landed once at most the PPW younger DMAs (tile + stride)
    // are outstanding (loads return in order; the previous epilogue's stores are older still)
    if (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave's pieces are visible; every wave is done
    asm volatile("" ::: "memory");         // with the slot refilled next and with the epilogue tile
    issue(tile + 2 * stride_t, (it + 2) % kNST);
    const unsigned char* st = lds + (it % kNST) * SLOT;

    f32x16 acc[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KS; ++kc) {
      const int q0 = (4 * kc + 2 * fk) ^ a_x, q1 = (4 * kc + 2 * fk + 1) ^ a_x;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + a_rowoff + (q0 << 4));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + a_rowoff + (q1 << 4));
      u32x2 h0, m0, l0, h1, m1, l1;
      split3(a0, h0, m0, l0);
      split3(a1, h1, m1, l1);
      bf16x8 fa[3];
      fa[0] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
      fa[1] = __builtin_bit_cast(bf16x8, u32x4{m0[0], m0[1], m1[0], m1[1]});
      fa[2] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
      // products (i, j) with i + j <= 2, smallest terms first (the order of conv_bfx.hip)
#pragma unroll
      for (int t = 2; t >= 0; --t)
#pragma unroll
        for (int i = 0; i <= t; ++i)
#pragma unroll
          for (int b = 0; b < NCB; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[kc][b][t - i], acc[b], 0, 0, 0);
    }

    // ---- epilogue through LDS: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[i * BN + (wave * NCB + b) * 32 + (lane & 31)] = acc[b][r];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (jcol < p.Cout) {
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + jcol);
#pragma unroll
      for (int ps = 0; ps < kBM / RPPS; ++ps) {
        const int i = r0 + ps * RPPS;
        const int m = tile * kBM + i;
        if (m >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * BN + c4);
        v += bias;
        if (p.res_mode == 1) {
          v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + jcol);
        } else if (p.res_mode == 2) {
          const int n = m / hw;
          const int rem = m - n * hw;
          const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
          v += *reinterpret_cast<const f32x4*>(
              p.res + (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + jcol);
        }
        if (p.relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (p.mask) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m * p.Cout + jcol);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
        }
        *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + jcol) = v;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the (zero-page) tail DMAs
}

int g_bres_enabled = -1, g_bres_last = 0;

}  // namespace

// Eligibility + launch.  Returns BGS_OK / BGS_ERR_LAUNCH when the layer was handled here, -1 when it
// is not eligible (the caller takes the general path).  `min_m`: smallest M worth a persistent grid.
int bgs_internal_conv1x1_bres(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, int planes,
                              const unsigned* zero, hipStream_t st) {
  if (g_bres_enabled < 0) {
    // OFF by default: in the detector step the one layer it won in isolation (fpn.lat0) runs 186 us
    // against the ring's ~160 (its 137 MB input is not cache-resident there), step 6.807 vs 6.782 ms
    // (profiles/r5l_*).  BGS_CONV1X1_BRES=1: fpn.lat0-shaped layers only; =2: every eligible layer.
    const char* e = getenv("BGS_CONV1X1_BRES");
    g_bres_enabled = e ? atoi(e) : 0;
    if (g_bres_enabled < 0 || g_bres_enabled > 2) g_bres_enabled = 0;
  }
  g_bres_last = 0;
  if (!g_bres_enabled || planes != 3) return -1;
  if (p.R != 1 || p.S != 1 || p.pad != 0 || (p.stride != 1 && p.stride != 2)) return -1;
  if (p.K != p.Cin || (p.Cin != 64 && p.Cin != 128 && p.Cin != 256)) return -1;
  if (p.res_mode == 3 || p.partial) return -1;
  const int bn = 256;
  if (p.Cout % bn != 0) return -1;
  if (p.M < 4096) return -1;
  // Measured (profiles/r5c_bres_ab.txt, interleaved A/B against the operand ring, bit-identical): the
  // filter-resident kernel wins only where a workgroup streams many tiles past its filter slab and
  // the layer is large enough to be bound by the memory path — fpn.lat0 (M = 134,400, K = Cout = 256:
  // 0.127 vs 0.135 ms); on the smaller layers the 32-row tile's fixed costs (two barriers, the LDS
  // transpose, 8 DMA pieces per wave) and ONE wave per SIMD lose to the ring (l3.c3 0.055 vs 0.041,
  // l2.c3 0.064 vs 0.044 ms).  g_bres_enabled == 2 (tests, A/B) lifts the restriction.
  if (g_bres_enabled != 2 && !(p.Cin == 256 && p.Cout == 256 && p.M >= 65536)) return -1;
  const uintptr_t al = (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.mask | (uintptr_t)p.bias |
                       (uintptr_t)p.x | (uintptr_t)wsplit;
  if (al & 15) return -1;
  BresArgs q;
  q.zero = zero;
  q.c = p;
  q.ws = reinterpret_cast<const __bf16*>(wsplit);
  q.KC = KC;
  q.tiles_m = (p.M + kBM - 1) / kBM;
  q.ncg = p.Cout / bn;
  // one workgroup per CU for K >= 128 (more than 256 registers per wave); two for K = 64
  const int per_cu = p.Cin >= 128 ? 1 : 2;
  int wpc = (256 * per_cu) / q.ncg;
  if (wpc < 1) wpc = 1;
  if (wpc > q.tiles_m) wpc = q.tiles_m;
  q.wg_per_cg = wpc;
  dim3 grid((unsigned)(q.ncg * wpc));
  bgs_internal_census_bump(BGS_CENSUS_CONV1X1_BRES);
  g_bres_last = 1;
#define BRES_L(KS_, NCB_)                                                                              \
  do {                                                                                                 \
    if (p.stride == 2)                                                                                 \
      hipLaunchKernelGGL((conv1x1_bres_kernel<KS_, NCB_, true>), grid, dim3(kThreads8), 0, st, q);     \
    else                                                                                               \
      hipLaunchKernelGGL((conv1x1_bres_kernel<KS_, NCB_, false>), grid, dim3(kThreads8), 0, st, q);    \
  } while (0)
  if (p.Cin == 256) BRES_L(16, 2);
  else if (p.Cin == 128) BRES_L(8, 2);
  else BRES_L(4, 2);
#undef BRES_L
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}

// test / A-B hook: 0 = never, 1 = where it was measured faster (default; env BGS_CONV1X1_BRES), 2 = on
// every layer the kernel can run
extern "C" void bgs_conv1x1_bres_enable(int on) { g_bres_enabled = on < 0 ? 0 : (on > 2 ? 2 : on); }
extern "C" int bgs_conv1x1_bres_last_launch(void) { return g_bres_last; }
