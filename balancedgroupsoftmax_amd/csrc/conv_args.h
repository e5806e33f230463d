// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip: fp32 MFMA; conv_bfx.hip: bf16
// MFMA on split operands): the argument block and the fused epilogue.
#pragma once

#include "bgs_common.h"

namespace bgs_conv {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
// LDS row = [k even: 8 floats][k odd: 8 floats][4 pad] = 20 floats.  The A/B fragment of
// v_mfma_f32_32x32x2_f32 wants, in lane l, k = 2*kk + (l >> 5) for kk = 0..7: lanes 0-31 need
// the even k of their row and lanes 32-63 the odd k — 8 contiguous floats each, fetched with two
// ds_read_b128 per K tile (instead of 8 ds_read_b32: the 64x64 tile was LDS-bandwidth bound).
// Stride 20 floats makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots.
// (BK = 16: stride 20 floats; BK = 32: stride 36 floats — both spread 16 consecutive rows over
// 16 distinct 16-byte slots of the 256-byte bank row.)

struct ConvArgs {
  const float* x;     // [N, H, W, Cin]
  const float* w;     // [Cout, R, S, Cin]
  const float* bias;  // [Cout] or null
  const float* res;   // residual or null: [N, Ho, Wo, Cout] (mode 1) / [N, Ho/2, Wo/2, Cout] (mode 2:
                      // nearest-2x upsampled) / [N, 2Ho, 2Wo, Cout] (mode 3: 2x2 sum-pooled)
  const float* mask;  // null, or [N, Ho, Wo, Cout]: y = mask > 0 ? y : 0 (ReLU backward)
  float* y;           // [N, Ho, Wo, Cout]
  int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
  int M, K;           // M = N*Ho*Wo, K = R*S*Cin
  int relu, res_mode;
  // split-K (small-M layers expose too few workgroups to fill 256 CUs): gridDim.z slices of
  // `kt_per_split` K tiles each write raw partial sums to partial[z][M][Cout]; the epilogue
  // (bias / residual / ReLU / mask) then runs in conv_splitk_epilogue_kernel.
  float* partial;
  int kt_per_split;
  // XCD-aware tile order: the launch is 1-D; workgroup b runs on XCD b % 8 (round-robin dispatch),
  // and is given tile (b % 8) * chunk + b / 8 — every XCD walks its own contiguous band of the
  // image, with the Cout tiles of one pixel tile back to back, so the 3x3 halo rows and the
  // re-read of the same pixels for the next Cout tile hit that XCD's L2 (PMC: 2.46 GB fetched per
  // 158-GFLOP layer before, profiles/r2d_pmc_conv.md).
  int tiles_m, tiles_n, chunk;
  // Parity-class row order (stride-2 data gradients, conv_bfx.hip UP == 3): the GEMM rows are the output
  // pixels GROUPED BY (y & 1, x & 1) — class c = 2 (y & 1) + (x & 1) occupies virtual rows
  // [c * rm_mcp, c * rm_mcp + rm_mc), rm_mcp = rm_mc rounded up to the 64-row tile — so that the filter
  // taps that meet a tile's rows are the same for the whole tile (the other taps see the zeros of the
  // upsampled input and are skipped).  conv_out_row maps a virtual row to its pixel.
  int rowmap = 0, rm_hh = 0, rm_wh = 0, rm_mc = 0, rm_mcp = 0;
};

// virtual GEMM row -> output pixel index (n * Ho + y) * Wo + x; -1 for the padding rows of a class
__device__ __forceinline__ long long conv_out_row(const ConvArgs& p, int m) {
  const int cls = m / p.rm_mcp, r = m - cls * p.rm_mcp;
  if (r >= p.rm_mc) return -1;
  const int hw = p.rm_hh * p.rm_wh;
  const int n = r / hw, rem = r - n * hw;
  const int i = rem / p.rm_wh, j = rem - i * p.rm_wh;
  return ((long long)n * p.Ho + 2 * i + (cls >> 1)) * p.Wo + 2 * j + (cls & 1);
}

// Fused epilogue of a (64*MB) x (64*NB) workgroup tile held as MB x NB accumulators of the 32x32
// MFMA per wave (2 x 2 waves).  C/D layout of every 32x32 MFMA on gfx950 (dtype-independent):
// col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int MB, int NB>
__device__ __forceinline__ void conv_store_tile(const ConvArgs& p, const f32x16 (&acc)[MB][NB],
                                                int m0, int n0, int wm, int wn, int lane) {
  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (p.partial) {
    float* part = p.partial + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = m0 + wm * 32 * MB + a * 32 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int j = n0 + wn * 32 * NB + b * 32 + (lane & 31);
          if (j < p.Cout) part[(size_t)m * p.Cout + j] = acc[a][b][r];
        }
      }
    return;
  }
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + wm * 32 * MB + a * 32 + i;
      if (m >= p.M) continue;
      size_t res_row = 0;
      if (p.res_mode == 1) {
        res_row = (size_t)m * p.Cout;
      } else if (p.res_mode == 2) {
        const int n = m / hw;
        const int rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
        res_row = (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout;
      } else if (p.res_mode == 3) {
        const int n = m / hw;
        const int rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
        res_row = (((size_t)n * (p.Ho * 2) + ho * 2) * (p.Wo * 2) + wo * 2) * p.Cout;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = n0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (j >= p.Cout) continue;
        float v = acc[a][b][r];
        if (p.bias) v += p.bias[j];
        if (p.res_mode == 3) {
          const size_t down = (size_t)p.Wo * 2 * p.Cout;
          v += (p.res[res_row + j] + p.res[res_row + p.Cout + j]) +
               (p.res[res_row + down + j] + p.res[res_row + down + p.Cout + j]);
        } else if (p.res_mode) {
          v += p.res[res_row + j];
        }
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.mask) v = p.mask[(size_t)m * p.Cout + j] > 0.f ? v : 0.f;
        p.y[(size_t)m * p.Cout + j] = v;
      }
    }
  }
}

// The same epilogue through an LDS transpose: the accumulators are written to a row-major tile
// in LDS (the operand buffers are free after the K loop) and every thread then handles four
// consecutive channels of a row — ONE 16-byte residual load and ONE 16-byte store where the C/D
// layout path issues four 4-byte ones.  The short-K layers (K = 64: four K steps) spend most of their
// vector-memory instructions in the epilogue; the CU's vector-memory path is what bounds these
// kernels (profiles/r2z_pmc_ring_kernel.md).  Identical arithmetic, identical results.
// Preconditions (checked by conv_epilogue_vec_ok): Cout % 4 == 0, 16-byte aligned y / residual /
// mask / bias / slab, residual mode 0 / 1 / 2.  `scratch`: (64 MB) x (64 NB + 4) floats of LDS that
// no wave reads any more (callers barrier first).
__device__ __forceinline__ bool conv_epilogue_vec_ok(const ConvArgs& p) {
  const uintptr_t a = (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.mask | (uintptr_t)p.bias |
                      (uintptr_t)p.partial;
  return (p.Cout & 3) == 0 && (a & 15) == 0 && p.res_mode != 3;
}

// Residual row of GEMM row m for the vector epilogue (modes 1 and 2; no row map).
__device__ __forceinline__ const float* conv_res_row(const ConvArgs& p, int m) {
  if (p.res_mode == 1) return p.res + (size_t)m * p.Cout;
  const int hw = p.Ho * p.Wo;
  const int n = m / hw;
  const int rem = m - n * hw;
  const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
  return p.res + (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout;
}

// The residual values a thread adds in conv_store_tile_lds, loaded AHEAD of the K loop: the epilogue's residual
// read is otherwise a third dependent HBM round trip in the life of a workgroup (operands -> MFMAs -> residual ->
// store), and the short-K layers (K = 64 .. 256: 4 - 16 steps) have too few bytes in flight per CU to hide it.
// Same addresses, same values, same arithmetic.  Returns false (nothing loaded) when the epilogue would not take
// the vector path or the tile is a split-K slab / row-mapped.
template <int MB, int NB>
__device__ __forceinline__ bool conv_prefetch_residual(const ConvArgs& p, int m0, int n0,
                                                       f32x4 (&pre)[64 * MB / (kThreads / (16 * NB))]) {
  constexpr int BM = 64 * MB, BN = 64 * NB;
  constexpr int TPR = BN / 4, RPP = kThreads / TPR;
  if (!(p.res_mode == 1 || p.res_mode == 2) || p.partial || p.rowmap || !conv_epilogue_vec_ok(p)) return false;
  const int tid = threadIdx.x;
  const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
  const int j = n0 + c4;
  const int jj = j < p.Cout ? j : 0;                     // (clamped, unconditional loads: no branch, all in flight)
#pragma unroll
  for (int ps = 0; ps < BM / RPP; ++ps) {
    const int m = min(m0 + r0 + ps * RPP, p.M - 1);
    pre[ps] = *reinterpret_cast<const f32x4*>(conv_res_row(p, m) + jj);
  }
  return true;
}

template <int MB, int NB, bool PRE>
__device__ __forceinline__ void conv_store_tile_lds_impl(const ConvArgs& p, const f32x16 (&acc)[MB][NB],
                                                         int m0, int n0, int wm, int wn, int lane,
                                                         float* scratch,
                                                         const f32x4 (&pre)[64 * MB / (kThreads / (16 * NB))]) {
  constexpr int BM = 64 * MB, BN = 64 * NB, LD = BN + 4;
  constexpr int TPR = BN / 4, RPP = kThreads / TPR;     // threads per row, rows per pass
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = wm * 32 * MB + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[i * LD + wn * 32 * NB + b * 32 + (lane & 31)] = acc[a][b][r];
      }
  __syncthreads();
  const int tid = threadIdx.x;
  const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
  const int j = n0 + c4;
  if (j >= p.Cout) return;
  if (p.partial) {
    float* part = p.partial + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
    for (int ps = 0; ps < BM / RPP; ++ps) {
      const int i = r0 + ps * RPP;
      const int m = m0 + i;
      if (m >= p.M) break;
      *reinterpret_cast<f32x4*>(part + (size_t)m * p.Cout + j) =
          *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
    }
    return;
  }
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + j);
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int ps = 0; ps < BM / RPP; ++ps) {
    const int i = r0 + ps * RPP;
    const int m = m0 + i;
    if (m >= p.M) break;
    size_t orow = (size_t)m;                             // output pixel of this GEMM row
    if (p.rowmap) {
      const long long rr = conv_out_row(p, m);
      if (rr < 0) continue;
      orow = (size_t)rr;
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
    v += bias;
    if (PRE) {
      v += pre[ps];
    } else if (p.res_mode == 1) {
      v += *reinterpret_cast<const f32x4*>(p.res + orow * p.Cout + j);
    } else if (p.res_mode == 2) {
      const int n = m / hw;
      const int rem = m - n * hw;
      const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
      v += *reinterpret_cast<const f32x4*>(
          p.res + (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + j);
    }
    if (p.relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
    }
    if (p.mask) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + orow * p.Cout + j);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
    }
    *reinterpret_cast<f32x4*>(p.y + orow * p.Cout + j) = v;
  }
}

template <int MB, int NB>
__device__ __forceinline__ void conv_store_tile_lds(const ConvArgs& p, const f32x16 (&acc)[MB][NB],
                                                    int m0, int n0, int wm, int wn, int lane,
                                                    float* scratch) {
  const f32x4 none[64 * MB / (kThreads / (16 * NB))] = {};
  conv_store_tile_lds_impl<MB, NB, false>(p, acc, m0, n0, wm, wn, lane, scratch, none);
}

}  // namespace bgs_conv

// split-K reduction + epilogue launch (defined in conv_igemm.hip)
int bgs_internal_conv_splitk_epilogue(bgs_conv::ConvArgs& p, int splits, hipStream_t st);
