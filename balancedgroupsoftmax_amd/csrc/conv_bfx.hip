// Implicit-GEMM convolution / linear layer on the bf16 matrix cores of gfx950 (MI355X) with
// fp32-faithful results: "bf16x6" operand splitting.
//
// Same contract as conv_igemm.hip (the layers the reference delegates to cuDNN / cuBLAS:
// mmdet/models/backbones/resnet.py:220-266, necks/fpn.py:101-141, anchor_heads/rpn_head.py:30-35,
// bbox_heads/convfc_bbox_head.py:132-168): fp32 NHWC activations in, fp32 out, fp32 accumulate.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (157 vs 2500 TFLOP/s).  An fp32
// value x is EXACTLY  hi + mid + lo + e,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (round-to-nearest-even; the subtractions are exact in fp32), |e| <= 2^-27 |x|.  The product
// a*b is then the sum of nine bf16 x bf16 products, each EXACT in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16; the six with i + j <= 2 are kept, the three dropped ones are
// <= 2^-25 |a b| — below fp32 rounding (2^-24).  Six bf16 MFMAs replace eight fp32 MFMA steps of
// 1/8 the K depth: 6/16 of the matrix-pipe time, the same (or smaller) error against an fp64
// reference as the fp32 MFMA kernel (tests/test_gpu_det_ops.py::test_bfx_error_not_above_f32_mfma).
//
//   * weights are split ONCE on the device (bgs_conv_bfx_split_weights) into three bf16 planes laid
//     out [plane][K/16][Cout][16]: the B slice of a K step is one contiguous, fully coalesced run;
//   * activations stay fp32 in HBM (nothing else in the detector changes); the A tile is split
//     in registers on its way to LDS (5.5 VALU ops per element, hidden under the 24 MFMAs of a
//     K step);
//   * workgroup = 4 waves (2 x 2), wave tile MB x NB blocks of 32 x 32, BK = 16 (one bf16 MFMA K
//     step); LDS rows are 32 B of data + 16 B pad (stride 3 x 16 B: the four 16-lane groups of
//     ds_read_b128 hit 16 distinct slots); double-buffered, one barrier per K step;
//   * epilogue, split-K and the XCD-banded tile order are those of conv_igemm.hip (conv_args.h).
#include <stdlib.h>

#include "conv_args.h"
#include "bfx_split.h"

using namespace bgs_conv;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Out-of-range operands (image border, K tail, rows past Cout) are LOADED from this zero page: the
// address is selected before the load, so the loaded registers need no select after it and stay
// in flight until they are staged — a branch around the load or a select behind it makes hipcc
// wait for the load right where it is issued (in front of the MFMAs instead of behind them).
__device__ __attribute__((aligned(16))) unsigned g_zero_page[16];

}  // namespace

// conv1x1_bres.hip
int bgs_internal_conv1x1_bres(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, int planes,
                              const unsigned* zero, hipStream_t st);
// conv_bfx_wide.hip: 128 x 128 tile, four M-stacked waves (-1: not eligible)
int bgs_internal_conv1x1_bfx_wide(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, void* workspace,
                                  size_t workspace_bytes, int forced_only, hipStream_t st);
void bgs_internal_conv1x1_bfx_wide_clear_last();
size_t bgs_internal_conv1x1_bfx_wide_workspace(long long M, int Cout, int K);
// conv1x1_planes.hip (round 6): 64 pixels x 256 channels per workgroup, A split once per tile into LDS planes (-1: not eligible)
int bgs_internal_conv1x1_planes(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, hipStream_t st);
void bgs_internal_conv1x1_planes_clear_last();
int bgs_internal_conv3x3_planes(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, hipStream_t st);   // conv3x3_planes.hip
int bgs_internal_conv3x3s2_planes(const bgs_conv::ConvArgs& p, const void* wsplit, int KC, hipStream_t st);
int bgs_internal_bottleneck_tail_planes(const float* x, const void* w2split, const float* bias2, const void* w3split,
                                        const float* bias3, const float* residual, float* y, int N, int H, int W,
                                        int relu3, hipStream_t st);     // bottleneck_tail_planes.hip
void bgs_internal_conv3x3_planes_clear_last();

namespace {

struct BfxArgs {
  const unsigned* zero;  // device address of g_zero_page (read through SGPRs by the DMA kernel)
  int ns;                // operand planes used: 3 = fp32-faithful (six products), 1 = bf16 operands
  ConvArgs c;
  const __bf16* ws;      // split weights [NS][KC][Cout][16]
  int KC;                // ceil(K / 16)
  int res_prefetch;      // LDS-DMA ring: load the epilogue's residual tile ahead of the K loop
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) {
  return __builtin_bit_cast(float, u & 0xffff0000u);
}

// x (4 consecutive k) -> three planes of 4 packed bf16 each
__device__ __forceinline__ void split3(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
  const f32x4 r = {bfx_resid_lo(hi[0], v[0]), bfx_resid_hi(hi[0], v[1]), bfx_resid_lo(hi[1], v[2]),
                   bfx_resid_hi(hi[1], v[3])};
  mid = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
  const f32x4 r2 = {bfx_resid_lo(mid[0], r[0]), bfx_resid_hi(mid[0], r[1]), bfx_resid_lo(mid[1], r[2]),
                    bfx_resid_hi(mid[1], r[3])};
  lo = u32x2{pack_bf16(r2[0], r2[1]), pack_bf16(r2[2], r2[3])};
}


// NS = 3: fp32-faithful (six products).  NS = 2: hi/mid only, three products (error ~2^-17: a
// tuning / ablation arm, not used by the detector).  UP as in conv_igemm.hip.
// BK = 16 or 32 (k depth staged per barrier).
// (Tried: a register prefetch depth of two K steps — hipcc re-uses the destination registers of the
// loads in flight for address arithmetic and waits for them at the top of the next step: slower.)
template <int MB, int NB, int BK, int NS, int UP>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_bfx_kernel(BfxArgs q) {
  const ConvArgs& p = q.c;
  constexpr int BM = 64 * MB, BN = 64 * NB;
  constexpr int KS = BK / 16;            // bf16 MFMA K steps per staged tile
  constexpr int LDR = BK * 2 + 16;       // LDS row stride in bytes: 48 / 80 (odd multiples of 16)
  constexpr int KQ = BK / 4;             // fp32 quads per A row per K step
  constexpr int RP = kThreads / KQ;      // A rows staged per pass
  constexpr int PA = BM / RP;
  constexpr int NPIECE = NS * KS * BN * 2;   // 16-byte pieces of the B tile (all planes)
  constexpr int PB = (NPIECE + kThreads - 1) / kThreads;
  constexpr int A_PLANE = BM * LDR, B_PLANE = BN * LDR;
  constexpr int BUF = NS * (A_PLANE + B_PLANE);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;          // workgroup-uniform
  const int m0 = (vtile / p.tiles_n) * BM, n0 = (vtile % p.tiles_n) * BN;

  // ---- A staging role: 4 consecutive k (one 16-byte load) of rows srow + RP*pass
  const int kq = tid % KQ;
  const int srow = tid / KQ;
  int a_hi0[PA], a_wi0[PA];
  const float* a_base[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + srow + RP * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_hi0[i] = ho * p.stride - p.pad;
    a_wi0[i] = wo * p.stride - p.pad;
    a_base[i] = p.x + (size_t)n * p.H * p.W * p.Cin;
  }
  // ---- B staging role: piece id = tid + 256*i -> (plane, row, half)
  const __bf16* b_src[PB];
  int b_dst[PB];
  bool b_use[PB], b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int id = tid + kThreads * i;
    b_use[i] = id < NPIECE;
    const int idc = b_use[i] ? id : 0;
    const int plane = idc / (KS * BN * 2);
    const int rem0 = idc - plane * (KS * BN * 2);
    const int ch = rem0 / (BN * 2);
    const int rem = rem0 - ch * (BN * 2);
    const int row = rem >> 1, half = rem & 1;
    b_ok[i] = b_use[i] && (n0 + row < p.Cout);
    b_src[i] = q.ws + (((size_t)plane * q.KC + ch) * p.Cout + (b_ok[i] ? n0 + row : 0)) * 16 + half * 8;
    b_dst[i] = NS * A_PLANE + plane * B_PLANE + row * LDR + ch * 32 + half * 16;
  }

  const int nk_all = q.KC / KS;          // KC is a multiple of 2 (zero-padded planes); KS = 1 here
  const int kt_begin = p.partial ? blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.partial ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
  int kt_load = kt_begin;
  int kg = kt_begin * BK + kq * 4;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }

  f32x4 ra0[PA];
  u32x4 rb0[PB];
  auto load_tile = [&](f32x4 (&ra)[PA], u32x4 (&rb)[PB]) {
    const bool kok = kg < p.K;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int hi = a_hi0[i] + kr, wi = a_wi0[i] + ks;
      bool ok = a_ok[i] && kok && hi >= 0 && wi >= 0;
      if (UP == 2) {
        ok = ok && !((hi | wi) & 1);
        hi >>= 1;
        wi >>= 1;
      }
      ok = ok && hi < p.H && wi < p.W;
      const float* src = ok ? a_base[i] + ((size_t)hi * p.W + wi) * p.Cin + kc
                            : reinterpret_cast<const float*>(g_zero_page);
      ra[i] = *reinterpret_cast<const f32x4*>(src);
    }
    const size_t koff = (size_t)kt_load * KS * p.Cout * 16;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const __bf16* src = b_ok[i] ? b_src[i] + koff : reinterpret_cast<const __bf16*>(g_zero_page);
      rb[i] = *reinterpret_cast<const u32x4*>(src);
    }
    ++kt_load;
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
  auto store_tile = [&](int buf, const f32x4 (&ra)[PA], const u32x4 (&rb)[PB]) {
    unsigned char* base = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      u32x2 h, m, l;
      split3(ra[i], h, m, l);
      unsigned char* d = base + (srow + RP * i) * LDR + kq * 8;
      *reinterpret_cast<u32x2*>(d) = h;
      if (NS >= 2) *reinterpret_cast<u32x2*>(d + A_PLANE) = m;
      if (NS >= 3) *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = l;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (b_use[i]) *reinterpret_cast<u32x4*>(base + b_dst[i]) = rb[i];
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  const int a_frag = (wm * 32 * MB + frow) * LDR + fk * 16;
  const int b_frag = NS * A_PLANE + (wn * 32 * NB + frow) * LDR + fk * 16;

  const int nk = kt_end - kt_begin;
  auto compute = [&](int buf) {
    const unsigned char* base = lds + buf * BUF;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      bf16x8 fa[NS][MB], fb[NS][NB];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int a = 0; a < MB; ++a)
          fa[s][a] = *reinterpret_cast<const bf16x8*>(base + a_frag + s * A_PLANE + a * 32 * LDR + s2 * 32);
#pragma unroll
        for (int b = 0; b < NB; ++b)
          fb[s][b] = *reinterpret_cast<const bf16x8*>(base + b_frag + s * B_PLANE + b * 32 * LDR + s2 * 32);
      }
      // products (i, j) with i + j <= NS - 1, smallest terms first; the MB x NB accumulators
      // interleave, so consecutive MFMAs are independent
#pragma unroll
      for (int t = NS - 1; t >= 0; --t)
#pragma unroll
        for (int i = 0; i <= t; ++i)
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[t - i][b], acc[a][b],
                                                                  0, 0, 0);
    }
  };
  load_tile(ra0, rb0);
  store_tile(0, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tile(ra0, rb0);      // global loads of tile kt+1 in flight under the MFMAs
    compute(buf);
    if (more) store_tile(buf ^ 1, ra0, rb0);
    __syncthreads();
  }
  // (the loop's last __syncthreads() is behind every wave's last fragment read)
  if (sizeof(lds) >= (size_t)BM * (BN + 4) * 4 && conv_epilogue_vec_ok(p)) {
    conv_store_tile_lds<MB, NB>(p, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(lds));
    return;
  }
  conv_store_tile<MB, NB>(p, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// 64 x 64 tile with an LDS-DMA operand ring (the small-grid layers: ResNet layer3/4, the upper FPN /
// RPN levels, the 1x1 convs, the FC heads — 50-100 us launches whose K loops are LATENCY-bound: one
// register-staged step costs an L2 round trip + split + ds_write + barrier ~ 1.2 us for 0.08 us of
// MFMA work, and a register prefetch deeper than one step is undone by hipcc, see above).
//   * both operands travel global -> LDS by `global_load_lds_dwordx4` (no VGPR staging, no
//     ds_write): a ring of NST = 4 stages of 10 KB (A: 64 rows x 16 fp32 raw = 4 KB, B: 3 planes x
//     64 rows x 16 bf16 = 6 KB), THREE K steps in flight; one counted `s_waitcnt vmcnt` + one raw
//     `s_barrier` per step (a plain __syncthreads() would drain the DMA queue);
//   * the DMA writes LDS lane-linearly, so the bank-conflict-free layouts are produced by permuting
//     the per-lane SOURCE addresses: A rows (64 B) keep their four 16-byte quads XOR-swizzled by
//     (row >> 2) & 3, B rows (32 B) swap halves on odd 8-row groups;
//   * A stays fp32 in LDS and is split into the three bf16 planes when a wave reads its fragment
//     (8 values per lane per K step: 44 VALU ops beside 6 MFMAs, on the other issue port);
//   * out-of-range operands (borders, K tail, rows >= Cout, steps past the end of the K loop) are
//     fetched from the zero page, so every wave issues the same number of DMAs per step and the
//     vmcnt arithmetic stays uniform.
constexpr int DMA_A_BYTES = 64 * 64, DMA_B_PLANE = 64 * 32;

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

//   * the loop is instruction-issue bound (four waves per SIMD share ~190 instructions per step for
//     6 MFMAs each): P1X1 (1x1 filter, no padding — the bulk of these launches) replaces the tap /
//     border arithmetic by one pointer increment per operand, and the zero page arrives as a
//     kernel argument (SGPRs) instead of a GOT load per step.
// ABL (only instantiated != 0 under -DBGS_ABLATE, tools/ablate.py): timing-only variants that drop
// one component of the loop — 1: MFMAs, 2: DMA issue, 4: fragment ds_reads, 8: the A split.
// NS = 3: fp32-faithful (six products).  NS = 1: the bf16 mode of cfg[4] — hi planes only: the B
// stage is one plane (two DMA pieces), A is rounded to bf16 when a wave reads its fragment, one MFMA
// per step.
// DMA_NST = 3 (default: 30 KB, FIVE workgroups per CU) | 4 (40 KB, four per CU).  Many layers of
// cfg[1] have a grid just above a multiple of the 1024 slots four workgroups per CU give (2100,
// 1050, 2112, 1056, 1232 workgroups: a last round of 2-5 % of the chip that costs a full workgroup
// duration); with five per CU those grids need one round less, and two stages in flight are enough
// to cover the L2 latency (188 cycles): -4 % over the igemm layers, never worse.
template <int UP, bool P1X1, int ABL = 0, int NS = 3, int DMA_NST = 4>
__global__ __launch_bounds__(kThreads, NS == 1 ? (DMA_NST == 3 ? 8 : 6) : (DMA_NST == 3 ? 5 : 4)) void conv_igemm_bfx_dma_kernel(BfxArgs q) {
  const ConvArgs& p = q.c;
  const unsigned* __restrict__ zero_page = q.zero;
  constexpr int DMA_STAGE = DMA_A_BYTES + NS * DMA_B_PLANE;
  constexpr int NBP = 2 * NS;                          // B pieces of 1 KB per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[DMA_NST * DMA_STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;          // workgroup-uniform
  const int m0 = (vtile / p.tiles_n) * 64, n0 = (vtile % p.tiles_n) * 64;

  // ---- A DMA role: wave w fills rows 16w .. 16w+15 of a stage; lane -> (row, physical quad)
  const int arow = wave * 16 + (lane >> 2);
  const int aq = (lane & 3) ^ ((arow >> 2) & 3);       // logical quad (4 consecutive k) fetched
  int a_hi0, a_wi0;
  const float* a_base;
  bool a_ok;
  // UP == 3: stride-2 data gradient with the rows grouped by output-pixel parity (ConvArgs::rowmap): the
  // tile's rows share (y & 1, x & 1), so only the taps kr = r_first, r_first + 2, .. / ks = s_first, .. meet
  // non-zero entries of the zero-upsampled dy — the K loop runs over those taps alone (1, 2 or 4 of the 9
  // of a 3x3 filter: a quarter of the MFMAs and operand loads of the UP == 2 form; none at all for three of
  // the four classes of a 1x1 filter, whose rows are just residual / mask epilogues)
  int r_first = 0, s_first = 0, n_r = 0, n_s = 0;
  if (UP == 3) {
    const int cls = m0 / p.rm_mcp;                      // tile-uniform: rm_mcp is a multiple of the tile
    const int r = m0 + arow - cls * p.rm_mcp;
    a_ok = r < p.rm_mc;
    const int rr = a_ok ? r : 0;
    const int hw = p.rm_hh * p.rm_wh;
    const int n = rr / hw;
    const int rem = rr - n * hw;
    const int i = rem / p.rm_wh, j = rem - i * p.rm_wh;
    a_hi0 = 2 * i + (cls >> 1) - p.pad;
    a_wi0 = 2 * j + (cls & 1) - p.pad;
    a_base = p.x + (size_t)n * p.H * p.W * p.Cin;
    r_first = (p.pad - (cls >> 1)) & 1;
    s_first = (p.pad - (cls & 1)) & 1;
    n_r = r_first < p.R ? (p.R - r_first + 1) / 2 : 0;
    n_s = s_first < p.S ? (p.S - s_first + 1) / 2 : 0;
  } else {
    const int m = m0 + arow;
    a_ok = m < p.M;
    const int mm = a_ok ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_hi0 = ho * p.stride - p.pad;
    a_wi0 = wo * p.stride - p.pad;
    a_base = p.x + (size_t)n * p.H * p.W * p.Cin;
  }
  // ---- B DMA role: six 1 KB blocks per stage (plane j/2, rows 32 (j%2) ..); wave w takes block w
  //      and, for w < 2, block w + 4
  const __bf16* b_src[2];
  bool b_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave + 4 * i;
    const int plane = j >> 1;
    const int row = (j & 1) * 32 + (lane >> 1);
    const int half = (lane & 1) ^ ((row >> 3) & 1);
    b_ok[i] = j < NBP && (n0 + row < p.Cout);
    b_src[i] = q.ws + ((size_t)plane * q.KC * p.Cout + (b_ok[i] ? n0 + row : 0)) * 16 + half * 8;
  }
  const bool one_b = wave < NBP;                        // wave-uniform: this wave carries B piece `wave`
  const bool two_b = wave + 4 < NBP;                    // .. and B piece `wave + 4`

  const int nk_all = q.KC;
  const int kt_begin = p.partial ? blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.partial ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
  const int c16n = p.Cin >> 4;                          // (UP == 3: Cin % 16 == 0, no split-K)
  const int nk = UP == 3 ? n_r * n_s * c16n : kt_end - kt_begin;
  int t_a = 0, t_b = 0, c16 = 0;                        // UP == 3: tap (r_first + 2 t_a, s_first + 2 t_b), channel chunk
  int kg = kt_begin * 16 + aq * 4;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  int kt_issue = 0;                                    // K steps issued so far (relative)
  // P1X1: the tap never changes -> fixed per-lane source pointers, advanced by one K step per issue
  const float* a_ptr = nullptr;
  if (P1X1) {
    const int hi = a_hi0, wi = a_wi0;                   // pad == 0: always inside the image
    a_ptr = a_base + ((size_t)hi * p.W + wi) * p.Cin + kg;
  }
  const size_t b_step = (size_t)p.Cout * 16;
  const __bf16* b_ptr0 = b_src[0] + (size_t)kt_begin * b_step;
  const __bf16* b_ptr1 = b_src[1] + (size_t)kt_begin * b_step;
  auto issue = [&]() {
    unsigned char* st = lds + (kt_issue % DMA_NST) * DMA_STAGE;
    const bool live = kt_issue < nk;
    const float* asrc;
    if (P1X1) {
      asrc = (live && a_ok && kg < p.K) ? a_ptr : reinterpret_cast<const float*>(zero_page);
      a_ptr += 16;
      kg += 16;
    } else if (UP == 3) {
      const int tkr = r_first + 2 * t_a, tks = s_first + 2 * t_b;
      const int hi = (a_hi0 + tkr) >> 1, wi = (a_wi0 + tks) >> 1;     // both sums are even
      const bool ok = live && a_ok && hi >= 0 && wi >= 0 && hi < p.H && wi < p.W;
      asrc = ok ? a_base + ((size_t)hi * p.W + wi) * p.Cin + c16 * 16 + aq * 4
                : reinterpret_cast<const float*>(zero_page);
      const size_t kstep = (size_t)((tkr * p.S + tks) * c16n + c16) * b_step;
      b_ptr0 = b_src[0] + kstep;
      b_ptr1 = b_src[1] + kstep;
      if (++c16 == c16n) {
        c16 = 0;
        if (++t_b == n_s) {
          t_b = 0;
          ++t_a;
        }
      }
    } else {
      int hi = a_hi0 + kr, wi = a_wi0 + ks;
      bool ok = live && a_ok && kg < p.K && hi >= 0 && wi >= 0;
      if (UP == 2) {
        ok = ok && !((hi | wi) & 1);
        hi >>= 1;
        wi >>= 1;
      }
      ok = ok && hi < p.H && wi < p.W;
      asrc = ok ? a_base + ((size_t)hi * p.W + wi) * p.Cin + kc
                : reinterpret_cast<const float*>(zero_page);
      kg += 16;
      kc += 16;
      while (kc >= p.Cin) {
        kc -= p.Cin;
        if (++ks == p.S) {
          ks = 0;
          ++kr;
        }
      }
    }
    if (!(ABL & 2)) glds16(asrc, st + wave * 1024);
    if (one_b && !(ABL & 2)) {
      const __bf16* bsrc = (live && b_ok[0]) ? b_ptr0 : reinterpret_cast<const __bf16*>(zero_page);
      glds16(bsrc, st + DMA_A_BYTES + wave * 1024);
    }
    if (two_b && !(ABL & 2)) {
      const __bf16* bsrc = (live && b_ok[1]) ? b_ptr1 : reinterpret_cast<const __bf16*>(zero_page);
      glds16(bsrc, st + DMA_A_BYTES + (wave + 4) * 1024);
    }
    b_ptr0 += b_step;
    b_ptr1 += b_step;
    ++kt_issue;
  };

  // ---- fragment roles
  const int frow = lane & 31, fk = lane >> 5;
  const int ar = wm * 32 + frow;
  const int ac = (ar >> 2) & 3;
  const int a_off0 = ar * 64 + (((2 * fk) ^ ac) << 4);
  const int a_off1 = ar * 64 + (((2 * fk + 1) ^ ac) << 4);
  const int br = wn * 32 + frow;
  const int b_off = DMA_A_BYTES + br * 32 + ((fk ^ ((br >> 3) & 1)) << 4);

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  // residual tile of the epilogue, requested before the first operand stage (older than every DMA: the counted
  // waits below are unaffected)
  f32x4 res_pre[4];
  const bool res_pf = q.res_prefetch && conv_prefetch_residual<1, 1>(p, m0, n0, res_pre);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < DMA_NST - 1; ++i) issue();
  f32x4 a0, a1;
  bf16x8 fb[3];
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most DMA_NST - 2 younger stages (1, 2 or 3 DMAs each) are still in flight
    if (DMA_NST == 4) {
      if (two_b) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (one_b) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      if (two_b) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (one_b) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // every wave's part of stage kt is visible; every wave is
    asm volatile("" ::: "memory");         // done reading stage kt-1 (= the slot refilled next)
    issue();
    const unsigned char* st = lds + (((ABL & 4) ? 0 : kt) % DMA_NST) * DMA_STAGE;
    if (!(ABL & 4) || kt == 0) {
      a0 = *reinterpret_cast<const f32x4*>(st + a_off0);
      a1 = *reinterpret_cast<const f32x4*>(st + a_off1);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        fb[s] = *reinterpret_cast<const bf16x8*>(st + b_off + s * DMA_B_PLANE);
    }
    bf16x8 fa[3];
    if (NS == 1) {
      fa[0] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(a0[0], a0[1]), pack_bf16(a0[2], a0[3]),
                                               pack_bf16(a1[0], a1[1]), pack_bf16(a1[2], a1[3])});
    } else if (ABL & 8) {
      fa[0] = __builtin_bit_cast(bf16x8, a0);
      fa[1] = __builtin_bit_cast(bf16x8, a1);
      fa[2] = __builtin_bit_cast(bf16x8, a0 + a1);
    } else {
      u32x2 h0, m0_, l0, h1, m1, l1;
      split3(a0, h0, m0_, l0);
      split3(a1, h1, m1, l1);
      fa[0] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
      fa[1] = __builtin_bit_cast(bf16x8, u32x4{m0_[0], m0_[1], m1[0], m1[1]});
      fa[2] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
    }
    if (ABL & 1) {
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(fa[s]), "v"(fb[s]));
    } else {
#pragma unroll
      for (int t = NS - 1; t >= 0; --t)
#pragma unroll
        for (int i = 0; i <= t; ++i)
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[t - i], acc[0][0], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the (zero-page) tail DMAs
  if (conv_epilogue_vec_ok(p)) {                       // workgroup-uniform
    __syncthreads();                                   // every wave is done with the ring
    if (res_pf)
      conv_store_tile_lds_impl<1, 1, true>(p, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(lds), res_pre);
    else
      conv_store_tile_lds<1, 1>(p, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(lds));
    return;
  }
  conv_store_tile<1, 1>(p, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// bf16 mode (NS = 1), large layers: 128 x 128 tile on the operand ring with EIGHT waves (4 x 2, each
// 32 x 64).  With one MFMA per product the 64 x 64 ring moves 6 KB per 4 MFMAs — the bf16 mode is
// bound by the memory path outright (cascade X101: 46 launches of M = 8400, K = 1024, Cout = 1024 at
// 61 us = 290 TFLOP/s of a 2500 peak); this tile moves 12 KB per 16 MFMAs and keeps 32 waves per CU
// (four workgroups of 37 KB).  (For the fp32-faithful mode the same tile was within 2 % of the 64 x 64
// ring and is not instantiated: profiles/r3h_ring8_sweep.txt.)
//   * stage = A 128 rows x 64 B fp32 (quads XOR-swizzled) + B 128 rows x 32 B (hi plane, halves
//     swapped on odd 8-row groups) = 12 KB, three stages; 12 DMA pieces, two slots per wave (wave w:
//     A rows 16 w.., B piece w for w < 4, a dummy piece into the scratch block otherwise);
//   * epilogue: the LDS transpose in two halves of 64 rows (512 threads, 16 rows per pass).
template <bool P1X1>
__global__ __launch_bounds__(512, 2) void conv_igemm_bf16_ring8_kernel(BfxArgs q) {
  const ConvArgs& p = q.c;
  const unsigned* __restrict__ zero_page = q.zero;
  constexpr int NST = 3, A_BYTES = 128 * 64, B_PLANE = 128 * 32, STAGE = A_BYTES + B_PLANE;
  constexpr int SCR = NST * STAGE;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[SCR + 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);               // 0..7
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;          // workgroup-uniform
  const int m0 = (vtile / p.tiles_n) * 128, n0 = (vtile % p.tiles_n) * 128;

  const int nk_all = q.KC;
  const int kt_begin = p.partial ? blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.partial ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
  const int nk = kt_end - kt_begin;

  // ---- A DMA role: rows 16 wave + (lane >> 2); lane -> physical quad, logical = physical ^ ((row >> 2) & 3)
  const int arow = wave * 16 + (lane >> 2);
  const int aq = (lane & 3) ^ ((arow >> 2) & 3);
  int a_hi0, a_wi0;
  const float* a_base;
  bool a_ok;
  {
    const int m = m0 + arow;
    a_ok = m < p.M;
    const int mm = a_ok ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_hi0 = ho * p.stride - p.pad;
    a_wi0 = wo * p.stride - p.pad;
    a_base = p.x + (size_t)n * p.H * p.W * p.Cin;
  }
  // ---- B DMA role (waves 0..3): rows 32 wave + (lane >> 1) of the hi plane
  const int brow_d = (wave & 3) * 32 + (lane >> 1);
  const int bhalf_d = (lane & 1) ^ ((brow_d >> 3) & 1);
  const bool has_b = wave < 4;                          // wave-uniform
  const bool b_ok = has_b && (n0 + brow_d < p.Cout);
  const __bf16* b_ptr = q.ws + (size_t)(b_ok ? n0 + brow_d : 0) * 16 + bhalf_d * 8 +
                        (size_t)kt_begin * p.Cout * 16;
  int kg = kt_begin * 16 + aq * 4;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  const float* a_ptr = nullptr;
  if (P1X1) a_ptr = a_base + ((size_t)a_hi0 * p.W + a_wi0) * p.Cin + kg;
  const size_t b_step = (size_t)p.Cout * 16;
  int kt_issue = 0;
  auto issue = [&]() {
    unsigned char* st = lds + (kt_issue % NST) * STAGE;
    const bool live = kt_issue < nk;
    const float* asrc;
    if (P1X1) {
      asrc = (live && a_ok && kg < p.K) ? a_ptr : reinterpret_cast<const float*>(zero_page);
      a_ptr += 16;
      kg += 16;
    } else {
      const int hi = a_hi0 + kr, wi = a_wi0 + ks;
      const bool ok = live && a_ok && kg < p.K && hi >= 0 && wi >= 0 && hi < p.H && wi < p.W;
      asrc = ok ? a_base + ((size_t)hi * p.W + wi) * p.Cin + kc
                : reinterpret_cast<const float*>(zero_page);
      kg += 16;
      kc += 16;
      while (kc >= p.Cin) {
        kc -= p.Cin;
        if (++ks == p.S) {
          ks = 0;
          ++kr;
        }
      }
    }
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page);
    glds16(asrc, st + wave * 1024);
    if (has_b) glds16((live && b_ok) ? b_ptr : zp, st + A_BYTES + wave * 1024);
    else glds16(zp, lds + SCR);
    b_ptr += b_step;
    ++kt_issue;
  };

  // ---- fragment roles: wave tile = rows 32 wm .., columns 64 wn .. (two 32-wide sub-tiles)
  const int frow = lane & 31, fk = lane >> 5;
  const int ar = wm * 32 + frow;
  const int ac = (ar >> 2) & 3;
  const int a_off0 = ar * 64 + (((2 * fk) ^ ac) << 4);
  const int a_off1 = ar * 64 + (((2 * fk + 1) ^ ac) << 4);
  const int br = wn * 64 + frow;                        // + 32 b keeps the 8-row-group parity
  const int b_off = A_BYTES + br * 32 + ((fk ^ ((br >> 3) & 1)) << 4);

  f32x16 acc[1][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

  issue();
  issue();
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // one younger stage (2 DMAs) may be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    const unsigned char* st = lds + (kt % NST) * STAGE;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + a_off0);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + a_off1);
    const bf16x8 fa = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(a0[0], a0[1]), pack_bf16(a0[2], a0[3]),
                                                       pack_bf16(a1[0], a1[1]), pack_bf16(a1[2], a1[3])});
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(st + b_off + b * 32 * 32);
      acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0][b], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the (zero-page) tail DMAs
  if (!conv_epilogue_vec_ok(p)) {
    conv_store_tile<1, 2>(p, acc, m0, n0, wm, wn, lane);
    return;
  }
  // ---- epilogue through LDS, two halves of 64 rows: scratch 64 x 132 floats
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int LD = 132;
  const int c4 = (tid & 31) * 4, r0 = tid >> 5;         // 32 threads per row, 16 rows per pass
  const int j = n0 + c4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && !p.partial && j < p.Cout) bias = *reinterpret_cast<const f32x4*>(p.bias + j);
  float* dst = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : p.y;
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                   // ring (h = 0) / previous half (h = 1) no longer read
    if ((wm >> 1) == h) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (wm & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          scratch[i * LD + wn * 64 + b * 32 + (lane & 31)] = acc[0][b][r];
        }
    }
    __syncthreads();
    if (j >= p.Cout) continue;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int i = r0 + ps * 16;
      const int m = m0 + h * 64 + i;
      if (m >= p.M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
      if (!p.partial) {
        v += bias;
        if (p.res_mode == 1) {
          v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + j);
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
          const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m * p.Cout + j);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(dst + (size_t)m * p.Cout + j) = v;
    }
  }
}

// (Tried: the same tile and ring with TWO waves per workgroup — each wave 32 rows x all 64 columns,
//  so that a row block's A fragment is split by one wave instead of two and the fixed per-step
//  instructions are shared by 12 MFMAs instead of 6.  Equal on the small grids, 15-30 % SLOWER on the
//  large ones (stem 0.216 vs 0.169 ms, FPN lateral P2 0.190 vs 0.159): at 40 KB of LDS per workgroup
//  only two waves per SIMD are resident.  profiles/r2s_bfx_sweep_dma2.txt.  Removed.)

// w [rows][K] fp32 -> out [NS][KC][rows][16] bf16 planes (zero-padded K tail)
__global__ __launch_bounds__(256) void bfx_split_weights_kernel(const float* __restrict__ w,
                                                                __bf16* __restrict__ out, int rows,
                                                                int K, int KC, int NS, int f32sec = 0) {
  const size_t total = (size_t)KC * rows * 8;          // bf16 pairs per plane
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (size_t)gridDim.x * 256) {
    const int kk = (int)(e & 7) * 2;
    const size_t t = e >> 3;
    const int row = (int)(t % rows);
    const int kc = (int)(t / rows);
    const int k = kc * 16 + kk;
    const float v0 = k < K ? w[(size_t)row * K + k] : 0.f;
    const float v1 = k + 1 < K ? w[(size_t)row * K + k + 1] : 0.f;
    const unsigned h = pack_bf16(v0, v1);
    const float r0 = bfx_resid_lo(h, v0), r1 = bfx_resid_hi(h, v1);
    const unsigned m = pack_bf16(r0, r1);
    const unsigned l = pack_bf16(bfx_resid_lo(m, r0), bfx_resid_hi(m, r1));
    unsigned* o = reinterpret_cast<unsigned*>(out) + e;
    o[0] = h;
    if (NS >= 2) o[total] = m;
    if (NS >= 3) o[2 * total] = l;
    // fourth section: the same [KC][rows][16] layout in fp32 (the halo kernels' FB32 form DMAs THIS — 2/3 of the
    // bytes of the three planes — and splits a wave's fragment in registers: same split3, same planes, same bits)
    // Only in the FB32 experiment (BGS_HALO_FB32=1 at process start): ADVICE r5 — every split weight paid 67 % more
    // bytes, and every per-step split of a trained weight the extra write traffic, for a default-off A/B arm.
    if (NS >= 3 && f32sec) reinterpret_cast<f32x2*>(reinterpret_cast<unsigned*>(out) + 3 * total)[e] = f32x2{v0, v1};
  }
}

// The same planes for the DATA-GRADIENT filter, straight from the forward filter: the data gradient is a
// convolution with wt[ci][r'][s'][co] = w[co][R-1-r'][S-1-s'][ci] (rows = Cin, K = R*S*Cout).  As tensor ops
// that is a flip + permute + contiguous (two launches) in front of the split, per trained conv and step.
__global__ __launch_bounds__(256) void bfx_split_weights_dgrad_kernel(const float* __restrict__ w,
                                                                      __bf16* __restrict__ out, int Cout,
                                                                      int R, int S, int Cin, int KC, int f32sec) {
  const int rows = Cin, K = R * S * Cout, RS = R * S;
  const size_t total = (size_t)KC * rows * 8;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (size_t)gridDim.x * 256) {
    const int kk = (int)(e & 7) * 2;
    const size_t t = e >> 3;
    const int row = (int)(t % rows);
    const int kc = (int)(t / rows);
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = kc * 16 + kk + u;
      v[u] = 0.f;
      if (k < K) {
        const int rs = k / Cout, co = k - rs * Cout;
        v[u] = w[((size_t)co * RS + (RS - 1 - rs)) * Cin + row];      // (R-1-r', S-1-s') = RS-1-rs
      }
    }
    const unsigned h = pack_bf16(v[0], v[1]);
    const float r0 = bfx_resid_lo(h, v[0]), r1 = bfx_resid_hi(h, v[1]);
    const unsigned m = pack_bf16(r0, r1);
    const unsigned l = pack_bf16(bfx_resid_lo(m, r0), bfx_resid_hi(m, r1));
    unsigned* o = reinterpret_cast<unsigned*>(out) + e;
    o[0] = h;
    o[total] = m;
    o[2 * total] = l;
    if (f32sec) reinterpret_cast<f32x2*>(o - e + 3 * total)[e] = f32x2{v[0], v[1]};      // (fp32 section, see above)
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 with a halo-resident A operand (the structure of conv_halo.hip, DESIGN.md
// appendix A) on split operands: the workgroup's 8 x 16 output pixels plus halo (10 x 18 pixels x
// 16 channels) are split ONCE per channel chunk into three bf16 planes in LDS and shared by the
// nine taps — the split cost and the A traffic drop 9x; only the pre-split filter slice streams
// per tap (contiguous 4 KB runs per plane).  A is single-buffered (one extra barrier per chunk),
// B double-buffered: 62.8 KB of LDS -> two workgroups per CU.  gridDim.z slices the channel chunks
// (split-K for the small maps); partial slabs go through the shared split-K epilogue.
constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = PH * PW;   // 180 patch pixels
constexpr int HLDR = 48;                                                    // LDS row stride, bytes
constexpr int AQ = PROWS * 4;                                               // fp32 quads per patch
constexpr int AQT = (AQ + kThreads - 1) / kThreads;                         // 3 per thread

struct HaloBfxArgs {
  int ns;
  ConvArgs c;            // x, bias, y, N, H, W, Cin, Cout, relu, M, partial, tiles_m/n, chunk
  const __bf16* ws;      // split weights [3][KC][Cout][16], K = 9 * Cin
  int KC;
  int tiles_y, tiles_x;
  int chunks_per_split;  // channel chunks per gridDim.z slice
  const unsigned* zero = nullptr;   // device zero page (DMA source of out-of-range rows, variant 4)
  int flags = 0;         // experiments (bgs_conv3x3_halo_bfx_tuning bits 20..23)
  // Tile window (variants 4 and 7): the launch covers tiles [tile_base, tile_base + tile_count) of the
  // (pixel tile, Cout tile) list; tile_count == 0: all of them.  The two-launch schedule of variant 7 gives
  // whole rounds of 256-pixel tiles to one launch and the remaining image rows to a variant-4 launch.
  int tile_base = 0, tile_count = 0;
};

// The halo kernels' epilogue through an LDS transpose (see conv_store_tile_lds): the 8 x 16 pixel x
// (64 NB) channel tile leaves in two halves of 64 pixels (the accumulators of the waves wm = 0, then
// wm = 1), every thread storing four consecutive channels of a pixel with one 16-byte access.
// `scratch`: 64 x (64 NB + 4) floats of LDS no wave reads any more.
template <int NB, int GTH = 8, int GTW = 16>
__device__ __forceinline__ void halo_store_tile_lds(const ConvArgs& p, const f32x16 (&acc)[2][NB], int n,
                                                    int ty, int tx, int n0, int wm, int wn, int lane,
                                                    float* scratch, bool nt_store = false) {
  constexpr int TH = GTH, TW = GTW;                       // pixel tile (shadows the 8 x 16 default)
  constexpr int BN = 64 * NB, LD = BN + 4, TPR = BN / 4, RPP = kThreads / TPR;
  const int tid = threadIdx.x;
  const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
  const int j = n0 + c4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && !p.partial && j < p.Cout) bias = *reinterpret_cast<const f32x4*>(p.bias + j);
  float* dst = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : p.y;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            scratch[i * LD + wn * 32 * NB + b * 32 + (lane & 31)] = acc[a][b][r];
          }
    }
    __syncthreads();
    if (j < p.Cout) {
#pragma unroll
      for (int ps = 0; ps < 64 / RPP; ++ps) {
        const int i = r0 + ps * RPP;
        const int m = h * 64 + i;
        if (TH * TW < 128 && m >= TH * TW) continue;              // padding rows of a tile of fewer than 128 pixels
        const int ho = ty * TH + m / TW, wo = tx * TW + m % TW;
        if (ho >= p.H || wo >= p.W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
        const size_t off = (((size_t)n * p.H + ho) * p.W + wo) * p.Cout + j;
        if (!p.partial) {
          v += bias;
          if (p.relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
          }
          if (p.mask) {      // data gradient: ReLU backward of the conv's input
            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + off);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
          }
        }
        if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + off));
        else *reinterpret_cast<f32x4*>(dst + off) = v;
      }
    }
    if (h == 0) __syncthreads();
  }
}

template <int NB>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_halo_bfx_kernel(HaloBfxArgs q) {
  const ConvArgs& p = q.c;
  constexpr int BN = 64 * NB;
  constexpr int A_PLANE = PROWS * HLDR, B_PLANE = BN * HLDR;
  constexpr int A_BYTES = 3 * A_PLANE, B_BUF = 3 * B_PLANE;
  constexpr int NPIECE = 3 * BN * 2;
  constexpr int PB = (NPIECE + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) unsigned char lds[A_BYTES + 2 * B_BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;                  // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int n = tm / (q.tiles_y * q.tiles_x);
  const int trem = tm - n * (q.tiles_y * q.tiles_x);
  const int ty = trem / q.tiles_x, tx = trem - ty * q.tiles_x;
  const int h0 = ty * TH - 1, w0 = tx * TW - 1;                // input coords of patch (0, 0)
  const int n0 = tn * BN;
  const int cchunks = p.Cin / 16;
  const int c_begin = p.partial ? blockIdx.z * q.chunks_per_split : 0;
  const int c_end = p.partial ? min(cchunks, c_begin + q.chunks_per_split) : cchunks;

  // ---- staging roles.  A: patch quads idx = tid + 256 i; prow = idx / 4, kq = idx % 4
  const float* a_src[AQT];
  int a_dst[AQT];
  bool a_use[AQT], a_in[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int idx = tid + kThreads * i;
    a_use[i] = idx < AQ;
    const int prow = a_use[i] ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    a_in[i] = a_use[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_src[i] = p.x + (((size_t)n * p.H + (a_in[i] ? hi : 0)) * p.W + (a_in[i] ? wi : 0)) * p.Cin + kq * 4;
    a_dst[i] = prow * HLDR + kq * 8;
  }
  // B: piece id = tid + 256 i -> (plane, row, half)
  const __bf16* b_src[PB];
  int b_dst[PB];
  bool b_use[PB], b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int id = tid + kThreads * i;
    b_use[i] = id < NPIECE;
    const int idc = b_use[i] ? id : 0;
    const int plane = idc / (BN * 2);
    const int rem = idc - plane * (BN * 2);
    const int row = rem >> 1, half = rem & 1;
    b_ok[i] = b_use[i] && (n0 + row < p.Cout);
    b_src[i] = q.ws + ((size_t)plane * q.KC * p.Cout + (b_ok[i] ? n0 + row : 0)) * 16 + half * 8;
    b_dst[i] = A_BYTES + plane * B_PLANE + row * HLDR + half * 16;
  }

  f32x4 ra[AQT];
  u32x4 rb[PB];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      const float* src = a_in[i] ? a_src[i] + chunk * 16 : reinterpret_cast<const float*>(g_zero_page);
      ra[i] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      if (!a_use[i]) continue;
      u32x2 h, m, l;
      split3(ra[i], h, m, l);
      unsigned char* d = lds + a_dst[i];
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + A_PLANE) = m;
      *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = l;
    }
  };
  auto load_b = [&](int chunk, int tap) {
    const size_t koff = (size_t)(tap * cchunks + chunk) * p.Cout * 16;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const __bf16* src = b_ok[i] ? b_src[i] + koff : reinterpret_cast<const __bf16*>(g_zero_page);
      rb[i] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (b_use[i]) *reinterpret_cast<u32x4*>(lds + buf * B_BUF + b_dst[i]) = rb[i];
  };

  // ---- fragment roles: lane frow of sub-tile a owns pixel m = 64 wm + 32 a + frow
  const int frow = lane & 31, fk = lane >> 5;
  int a_frag[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = wm * 64 + a * 32 + frow;
    a_frag[a] = ((m >> 4) * PW + (m & 15)) * HLDR + fk * 16;      // patch row of tap (0, 0)
  }
  const int b_frag = A_BYTES + (wn * 32 * NB + frow) * HLDR + fk * 16;

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nsteps = (c_end - c_begin) * 9;
  load_a(c_begin);
  load_b(c_begin, 0);
  store_a();
  store_b(0);
  __syncthreads();
  int chunk = c_begin, tap = 0;
  for (int t = 0; t < nsteps; ++t) {
    const int bbuf = t & 1;
    const bool more = t + 1 < nsteps;
    int nchunk = chunk, ntap = tap + 1;
    if (ntap == 9) {
      ntap = 0;
      ++nchunk;
    }
    if (more) load_b(nchunk, ntap);                             // next filter slice in flight
    const bool next_a = tap == 0 && chunk + 1 < c_end;
    if (next_a) load_a(chunk + 1);                              // next patch: held in registers

    const int tap_off = ((tap / 3) * PW + (tap % 3)) * HLDR;
    bf16x8 fa[3][2], fb[3][NB];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[a] + tap_off + s * A_PLANE);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        fb[s][b] = *reinterpret_cast<const bf16x8*>(lds + bbuf * B_BUF + b_frag + s * B_PLANE + b * 32 * HLDR);
    }
#pragma unroll
    for (int tt = 2; tt >= 0; --tt)
#pragma unroll
      for (int i = 0; i <= tt; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[tt - i][b], acc[a][b],
                                                                0, 0, 0);
    if (more) store_b(bbuf ^ 1);
    __syncthreads();
    if (tap == 8 && chunk + 1 < c_end) {                        // every wave is done with patch `chunk`
      store_a();
      __syncthreads();
    }
    tap = ntap;
    chunk = nchunk;
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* part = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = wm * 64 + a * 32 + i;
      const int ho = ty * TH + (m >> 4), wo = tx * TW + (m & 15);
      if (ho >= p.H || wo >= p.W) continue;
      const size_t row = (((size_t)n * p.H + ho) * p.W + wo) * p.Cout;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = n0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (j >= p.Cout) continue;
        float v = acc[a][b][r];
        if (part) {
          part[row + j] = v;
        } else {
          if (p.bias) v += p.bias[j];
          if (p.relu) v = fmaxf(v, 0.f);
          p.y[row + j] = v;
        }
      }
    }
  }
}

// Variant 2 (default): the same data flow with (i) the nine taps of a chunk fully unrolled — the
// tap's patch offset is an immediate of the ds_read, no per-step integer division — and (ii) the
// filter slices stored UNPADDED (32-byte rows) with the two 16-byte halves of a row swapped on odd
// 8-row groups (conflict-free ds_read_b128 without padding): 50.5 KB of LDS and <= 168 VGPRs ->
// THREE workgroups per CU.  Measured on the P2 layer (profiles/r2f_pmc_bfx_halo_raw.txt): with two
// workgroups per CU the matrix pipe is busy 49 % of the time — the four waves of a workgroup sit
// on four SIMDs, each shared with ONE wave of another workgroup, and every barrier couples them;
// a third resident workgroup fills the gaps.  (A mid-step barrier / fragment read-ahead pipeline
// inside the wave, as in conv_igemm.hip, measured +0 % here and cost 44 VGPRs: dropped.)
// ABL as above — 1: MFMAs, 2: filter-slice loads + stores, 4: fragment ds_reads (first step only),
// 8: barriers, 16: patch reload (load_a / store_a).
template <int NB, int NS, int ABL = 0>
__global__ __launch_bounds__(kThreads, 3) void conv3x3_halo_bfx3_kernel(HaloBfxArgs q) {
  const ConvArgs& p = q.c;
  constexpr int BN = 64 * NB;
  constexpr int A_PLANE = PROWS * HLDR, A_BYTES = NS * A_PLANE;
  constexpr int B_PLANE = BN * 32, B_BUF = NS * B_PLANE;
  constexpr int NPIECE = NS * BN * 2;
  constexpr int PB = (NPIECE + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) unsigned char lds[A_BYTES + 2 * B_BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;                  // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int n = tm / (q.tiles_y * q.tiles_x);
  const int trem = tm - n * (q.tiles_y * q.tiles_x);
  const int ty = trem / q.tiles_x, tx = trem - ty * q.tiles_x;
  const int h0 = ty * TH - 1, w0 = tx * TW - 1;                // input coords of patch (0, 0)
  const int n0 = tn * BN;
  const int cchunks = p.Cin / 16;
  const int c_begin = p.partial ? blockIdx.z * q.chunks_per_split : 0;
  const int c_end = p.partial ? min(cchunks, c_begin + q.chunks_per_split) : cchunks;

  // ---- staging roles.  A: patch quads idx = tid + 256 i; prow = idx / 4, kq = idx % 4
  const float* a_src[AQT];
  int a_dst[AQT];
  bool a_use[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int idx = tid + kThreads * i;
    a_use[i] = idx < AQ;
    const int prow = a_use[i] ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = a_use[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    // outside the image: the zero page, with a zero channel stride (a_cs)
    a_src[i] = in ? p.x + (((size_t)n * p.H + hi) * p.W + wi) * p.Cin + kq * 4
                  : reinterpret_cast<const float*>(g_zero_page);
    a_dst[i] = (in ? 1 : 0) | ((prow * HLDR + kq * 8) << 1);   // bit 0: advances with the chunk
  }
  // B: piece id = tid + 256 i -> (plane, row, half); halves swapped on odd 8-row groups
  const __bf16* b_src[PB];
  int b_dst[PB];
  bool b_use[PB], b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int id = tid + kThreads * i;
    b_use[i] = id < NPIECE;
    const int idc = b_use[i] ? id : 0;
    const int plane = idc / (BN * 2);
    const int rem = idc - plane * (BN * 2);
    const int row = rem >> 1, half = rem & 1;
    b_ok[i] = b_use[i] && (n0 + row < p.Cout);
    b_src[i] = b_ok[i] ? q.ws + ((size_t)plane * q.KC * p.Cout + n0 + row) * 16 + half * 8
                       : reinterpret_cast<const __bf16*>(g_zero_page);
    b_dst[i] = A_BYTES + plane * B_PLANE + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
  }

  f32x4 ra[AQT];
  u32x4 rb[PB];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + ((a_dst[i] & 1) ? chunk * 16 : 0));
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      if (!a_use[i]) continue;
      u32x2 h, m, l;
      split3(ra[i], h, m, l);
      unsigned char* d = lds + (a_dst[i] >> 1);
      *reinterpret_cast<u32x2*>(d) = h;
      if (NS >= 2) *reinterpret_cast<u32x2*>(d + A_PLANE) = m;
      if (NS >= 3) *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = l;
    }
  };
  auto load_b = [&](int chunk, int tap) {
    const size_t koff = (size_t)(tap * cchunks + chunk) * p.Cout * 16;
#pragma unroll
    for (int i = 0; i < PB; ++i)
      rb[i] = *reinterpret_cast<const u32x4*>(b_src[i] + (b_ok[i] ? koff : 0));
  };
  auto store_b = [&](int buf_off) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (b_use[i]) *reinterpret_cast<u32x4*>(lds + buf_off + b_dst[i]) = rb[i];
  };

  // ---- fragment roles: lane frow of sub-tile a owns pixel m = 64 wm + 32 a + frow
  const int frow = lane & 31, fk = lane >> 5;
  int a_frag[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = wm * 64 + a * 32 + frow;
    a_frag[a] = ((m >> 4) * PW + (m & 15)) * HLDR + fk * 16;      // patch row of tap (0, 0)
  }
  const int brow = wn * 32 * NB + frow;                           // + 32 b: same 8-row-group parity
  const int b_frag = A_BYTES + brow * 32 + ((fk ^ ((brow >> 3) & 1)) << 4);

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  load_a(c_begin);
  load_b(c_begin, 0);
  store_a();
  store_b(0);
  __syncthreads();
  int cur = 0, nxt = B_BUF;                                      // byte offsets of the two B buffers
  bf16x8 fa[NS][2], fb[NS][NB];
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const bool last_chunk = chunk + 1 >= c_end;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int rd = (tap & 1) ? nxt : cur;                       // buffer of this step
      const int wr = (tap & 1) ? cur : nxt;                       // buffer of the next step
      const bool more = tap < 8 || !last_chunk;
      if (!(ABL & 2)) {
        if (tap < 8) load_b(chunk, tap + 1);                      // next filter slice in flight
        else if (more) load_b(chunk + 1, 0);
      }
      if (tap == 0 && !last_chunk && !(ABL & 16)) load_a(chunk + 1);   // next patch: held in registers
      const int tap_off = ((tap / 3) * PW + (tap % 3)) * HLDR;    // compile-time constant
      if (!(ABL & 4) || (chunk == c_begin && tap == 0)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
            fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[a] + tap_off + s * A_PLANE);
#pragma unroll
          for (int b = 0; b < NB; ++b)
            fb[s][b] = *reinterpret_cast<const bf16x8*>(lds + rd + b_frag + s * B_PLANE + b * 32 * 32);
        }
      }
      if (ABL & 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int a = 0; a < 2; ++a) asm volatile("" ::"v"(fa[s][a]));
#pragma unroll
          for (int b = 0; b < NB; ++b) asm volatile("" ::"v"(fb[s][b]));
        }
      } else {
#pragma unroll
        for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
          for (int i = 0; i <= tt; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[tt - i][b], acc[a][b],
                                                                    0, 0, 0);
      }
      if (more && !(ABL & 2)) store_b(wr);
      if (!(ABL & 8)) __syncthreads();
      if (tap == 8 && !last_chunk && !(ABL & 16)) {               // every wave is done with this patch
        store_a();
        if (!(ABL & 8)) __syncthreads();
      }
    }
    // nine steps per chunk: the buffer roles swap from chunk to chunk
    const int t = cur;
    cur = nxt;
    nxt = t;
  }

  // (the loop's last barrier is behind every wave's last fragment read)
  if (sizeof(lds) >= (size_t)64 * (BN + 4) * 4 && conv_epilogue_vec_ok(p)) {
    halo_store_tile_lds<NB>(p, acc, n, ty, tx, n0, wm, wn, lane, reinterpret_cast<float*>(lds));
    return;
  }
  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* part = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = wm * 64 + a * 32 + i;
      const int ho = ty * TH + (m >> 4), wo = tx * TW + (m & 15);
      if (ho >= p.H || wo >= p.W) continue;
      const size_t row = (((size_t)n * p.H + ho) * p.W + wo) * p.Cout;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = n0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (j >= p.Cout) continue;
        float v = acc[a][b][r];
        if (part) {
          part[row + j] = v;
        } else {
          if (p.bias) v += p.bias[j];
          if (p.relu) v = fmaxf(v, 0.f);
          p.y[row + j] = v;
        }
      }
    }
  }
}

// Variant 4 (default): the filter slices reach LDS by DMA (`global_load_lds_dwordx4`) instead of a
// round trip through VGPRs.  Component ablation of variant 2 on the P2 layer (tools/ablate.py,
// profiles/r2w_ablate_conv_loops.txt): 0.89 ms = 0.61 ms of MFMA issue + 0.36 ms of staging that
// does not hide behind it; the filter slice (3 global loads + 3 ds_write_b128 per thread and tap)
// is the largest part.  Here every tap's slice is 12 (NB = 2) / 6 (NB = 1) DMA pieces of 1 KB
// issued right after the step's barrier into the other buffer: no VGPRs, no ds_write, one
// vmcnt(0) before the barrier.  Bit-identical to variant 2; 3.49 -> 3.33 ms over the 3x3 layers of
// a cfg[1] forward (profiles/r2x_halo4_sweep.txt).  The patch keeps the register path (global
// load, split, ds_write_b64): a DMA of a pre-split ("split-form") patch written by the producing
// layer's epilogue was built and measured too — no faster, 1.5x the activation bytes: removed.
// LDS: 24 KB filter double buffer + 1 KB scratch (dummy pieces) + 16.9 KB patch (8 x 16 tile; 25.3 KB otherwise).
// (NS = 1: the bf16 mode — hi planes only, one MFMA per product.)
// PF (variant 5): the A fragments of tap t + 1 are read from the patch DURING tap t's MFMAs — the
// patch does not change inside a channel chunk, so those reads need not sit behind the step's
// barrier; after the barrier only the six filter fragments remain between a wave and its MFMAs.
// GTH x GTW: the pixel tile (<= 128 pixels; the MFMA rows beyond GTH * GTW are padding: they read patch pixel
// (0, 0) and are never stored).  8 x 16 everywhere but on the small maps, where the tile that wastes the fewest
// rows is chosen per layer (halo_bfx_geom: 50 x 84 -> 10 x 12: 35 tiles per image instead of 42; 25 x 42 -> 5 x 21:
// 10 instead of 12).
// Measured and not kept (profiles/r8h_halo_lds_layout_ring3_dma_ablation.txt): a ring of three filter slices with
// counted waits (the slice gets two steps to land: 0.717 -> 0.744 ms on the P2 layer), ONE filter buffer with a
// second barrier (four workgroups per CU: 0.92 ms).  What the step is co-limited by besides the matrix pipe is the
// 12 KB filter slice per workgroup and tap itself (timing-only ablation: 0.77 ms with it, 0.52 ms without) — on
// either path: a variant whose waves load their filter fragments straight from global memory into the MFMA's
// registers (no filter LDS, no DMA, ONE barrier per channel chunk instead of ten, bit-identical) was 4.5 % slower.
template <int NB, int NS = 3, bool PF = false, int GTH = 8, int GTW = 16, bool PADDED = false>
__global__ __launch_bounds__(kThreads, 3) void conv3x3_halo_bfx4_kernel(HaloBfxArgs q) {
  const ConvArgs& p = q.c;
  constexpr int TH = GTH, TW = GTW, PH = TH + 2, PW = TW + 2, PROWS = PH * PW;      // (shadow the 8 x 16 default)
  constexpr int AQ = PROWS * 4, AQT = (AQ + kThreads - 1) / kThreads;
  static_assert(TH * TW <= 128 && PROWS <= 180 && AQT <= 3, "patch no larger than the 8 x 16 tile's");
  constexpr int BN = 64 * NB;
  constexpr int B_PLANE = BN * 32, B_BUF = NS * B_PLANE;
  constexpr int SCR = 2 * B_BUF;                         // 1 KB scratch: target of dummy pieces (NB = 1 only)
  constexpr int A_OFF = SCR + (NB == 1 ? 1024 : 0);
  // Patch layout.  8 x 16 tile: 32 bytes per patch pixel (its 16 bf16, no padding); the two 16-byte k halves of
  // pixel (r, c) are swapped when r + c is odd.  Every 16-lane group of the A-fragment ds_read_b128 (lanes
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: eight pixels of patch row R and eight of row R + 1) then covers
  // the 16 slots of the 256-byte bank row exactly once for every tap: the two rows' pixels that share a slot pair
  // have columns of equal parity, i.e. r + c of opposite parity.  The 48-byte rows used before (and still for the
  // other tiles) are conflict-free for 32 CONSECUTIVE pixels but not across the 18-pixel row wrap: every group
  // 2-way, 24 of a wave's 72 LDS cycles per step — the 34 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of
  // profiles/r5_pmc_final_tree.md — and the ds_write_b64 of the split (four pixels = 192 bytes per 16 lanes) was
  // 2-way as well; 32-byte pixels make that store one 128-byte window per lane group.
  constexpr bool SWZ = GTH == 8 && GTW == 16 && !PADDED;      // (PADDED: the 48-byte rows on the 8 x 16 tile, A/B only)
  constexpr int HL = SWZ ? 32 : HLDR;
  constexpr int A_PLANE = PROWS * HL;
  constexpr int OPER_BYTES = A_OFF + NS * A_PLANE, EPI_BYTES = 64 * (BN + 4) * 4;   // operands | epilogue tile
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES];
  const unsigned* __restrict__ zero_page = q.zero;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int vlocal = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vlocal >= (q.tile_count ? q.tile_count : p.tiles_m * p.tiles_n)) return;   // workgroup-uniform
  const int vtile = vlocal + q.tile_base;
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int n = tm / (q.tiles_y * q.tiles_x);
  const int trem = tm - n * (q.tiles_y * q.tiles_x);
  const int ty = trem / q.tiles_x, tx = trem - ty * q.tiles_x;
  const int h0 = ty * TH - 1, w0 = tx * TW - 1;                // input coords of patch (0, 0)
  const int n0 = tn * BN;
  const int cchunks = p.Cin / 16;
  const int c_begin = p.partial ? blockIdx.z * q.chunks_per_split : 0;
  const int c_end = p.partial ? min(cchunks, c_begin + q.chunks_per_split) : cchunks;

  // ---- filter DMA roles.  NB = 2: wave w carries rows 32 w .. 32 w + 31 of the three planes.
  //      NB = 1: six pieces (plane, row half): wave w takes piece w and, w < 2, piece w + 4 (a
  //      dummy piece into the scratch block otherwise: every wave issues the same DMA count).
  const int brow_d = (NB == 2 ? wave * 32 : (wave & 1) * 32) + (lane >> 1);
  const int bhalf_d = (lane & 1) ^ ((brow_d >> 3) & 1);
  const bool b_okd = n0 + brow_d < p.Cout;
  const __bf16* b_lane = q.ws + (size_t)(b_okd ? n0 + brow_d : 0) * 16 + bhalf_d * 8;
  const size_t b_plane = (size_t)q.KC * p.Cout * 16;
  auto issue_b = [&](int chunk, int tap, int buf_off) {
    const size_t koff = (size_t)(tap * cchunks + chunk) * p.Cout * 16;
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page);
    if (NB == 2) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#ifdef BGS_ABLATE
        if ((q.flags & 4) && s > 0) continue;                    // timing only: one of the three plane DMAs
        if (q.flags & 8) continue;                               // timing only: no filter DMA at all
#endif
        glds16(b_okd ? b_lane + s * b_plane + koff : zp, lds + buf_off + s * B_PLANE + wave * 1024);
      }
    } else if (NS == 3) {
      const int s0 = wave >> 1;                                  // piece w: plane w / 2, half w % 2
      glds16(b_okd ? b_lane + s0 * b_plane + koff : zp, lds + buf_off + s0 * B_PLANE + (wave & 1) * 1024);
      if (wave < 2) glds16(b_okd ? b_lane + 2 * b_plane + koff : zp, lds + buf_off + 2 * B_PLANE + wave * 1024);
      else glds16(zp, lds + SCR);
    } else {                                                     // one plane: two pieces
      if (wave < 2) glds16(b_okd ? b_lane + koff : zp, lds + buf_off + wave * 1024);
      else glds16(zp, lds + SCR);
    }
  };

  // ---- patch staging roles (variant 2's): patch quads idx = tid + 256 i; prow = idx / 4, kq = idx % 4
  const float* a_src[AQT];
  int a_dst[AQT];
  bool a_use[AQT];
  f32x4 ra[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int idx = tid + kThreads * i;
    a_use[i] = idx < AQ;
    const int prow = a_use[i] ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = a_use[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_src[i] = in ? p.x + (((size_t)n * p.H + hi) * p.W + wi) * p.Cin + kq * 4
                  : reinterpret_cast<const float*>(g_zero_page);
    const int kq_off = SWZ ? ((((kq >> 1) ^ ((pr + pc) & 1)) << 4) + (kq & 1) * 8) : kq * 8;
    a_dst[i] = (in ? 1 : 0) | ((prow * HL + kq_off) << 1);     // bit 0: advances with the chunk
  }
  const bool nt_load = (q.flags & 16) != 0;                      // BGS_HALO_NT bit 0 (A/B): patch loads non-temporal
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      const f32x4* src = reinterpret_cast<const f32x4*>(a_src[i] + ((a_dst[i] & 1) ? chunk * 16 : 0));
      ra[i] = nt_load ? __builtin_nontemporal_load(src) : *src;
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      if (!a_use[i]) continue;
      u32x2 h, m, l;
      split3(ra[i], h, m, l);
      unsigned char* d = lds + A_OFF + (a_dst[i] >> 1);
      *reinterpret_cast<u32x2*>(d) = h;
      if (NS >= 2) *reinterpret_cast<u32x2*>(d + A_PLANE) = m;
      if (NS >= 3) *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = l;
    }
  };

  // ---- fragment roles: lane frow of sub-tile a owns pixel m = 64 wm + 32 a + frow
  const int frow = lane & 31, fk = lane >> 5;
  int a_frag[2][2];                                                       // [sub-tile][parity of the tap's dy + dx]
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int m = wm * 64 + a * 32 + frow;
    if (TH * TW < 128 && m >= TH * TW) m = 0;                             // padding row: any valid patch pixel
    const int pr = m / TW, pc = m % TW;                                   // patch pixel of tap (0, 0)
#pragma unroll
    for (int par = 0; par < 2; ++par)
      a_frag[a][par] = A_OFF + (pr * PW + pc) * HL + ((SWZ ? (fk ^ ((pr + pc + par) & 1)) : fk) << 4);
  }
  const int brow = wn * 32 * NB + frow;                           // + 32 b: same 8-row-group parity
  const int b_frag = brow * 32 + ((fk ^ ((brow >> 3) & 1)) << 4);

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (q.flags & 2) {                                             // static priority = hardware wave slot on the SIMD
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_ID.WAVE_ID [3:0]
    const unsigned pr = hwid & 3u;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  load_a(c_begin);
  issue_b(c_begin, 0, 0);
  store_a();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0, nxt = B_BUF;                                      // byte offsets of the two B buffers
  bf16x8 fa_next[NS][2];
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const bool last_chunk = chunk + 1 >= c_end;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int rd = (tap & 1) ? nxt : cur;                       // buffer of this step
      const int wr = (tap & 1) ? cur : nxt;                       // buffer of the next step
      if (tap < 8) issue_b(chunk, tap + 1, wr);                   // next filter slice: DMA in flight
      else if (!last_chunk) issue_b(chunk + 1, 0, wr);
      if (tap == 0 && !last_chunk) {                              // next patch: held in registers
        __builtin_amdgcn_sched_barrier(0);                        // (behind the slice DMAs: see the counted wait below)
#ifdef BGS_ABLATE
        if (!(q.flags & 0x100))
#endif
        load_a(chunk + 1);
      }
      const int tap_off = ((tap / 3) * PW + (tap % 3)) * HL;      // compile-time constants
      const int tap_par = (tap / 3 + tap % 3) & 1;
      bf16x8 fa[NS][2], fb[NS][NB];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (PF && tap > 0) fa[s][a] = fa_next[s][a];
          else fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[a][tap_par] + tap_off + s * A_PLANE);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
          fb[s][b] = *reinterpret_cast<const bf16x8*>(lds + rd + b_frag + s * B_PLANE + b * 32 * 32);
      }
      if (PF && tap < 8) {                                         // tap + 1's patch rows: in flight under the MFMAs
        const int nxt_off = (((tap + 1) / 3) * PW + ((tap + 1) % 3)) * HL;
        const int nxt_par = ((tap + 1) / 3 + (tap + 1) % 3) & 1;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            fa_next[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[a][nxt_par] + nxt_off + s * A_PLANE);
      }
      if (q.flags & 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
      }
#pragma unroll
      for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
        for (int i = 0; i <= tt; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[tt - i][b], acc[a][b],
                                                                  0, 0, 0);
      if (q.flags & 1) __builtin_amdgcn_s_setprio(0);
      // next slice landed.  vmcnt retires in order: in the step that issued the patch loads (AQT per lane, BEHIND the
      // slice DMAs) the counted wait lets them fly one step longer (flag 64: BGS_HALO_NT bit 2, A/B)
      if (tap == 0 && !last_chunk && (q.flags & 64)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AQT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#ifdef BGS_ABLATE
      if (!(q.flags & 0x200))
#endif
      if (tap == 8 && !last_chunk) {                              // every wave is done with this patch
        store_a();
        __syncthreads();
      }
    }
    // nine steps per chunk: the buffer roles swap from chunk to chunk
    const int t = cur;
    cur = nxt;
    nxt = t;
  }

#ifdef BGS_ABLATE
  if (q.flags & 0x400) {                                         // timing only: no epilogue
    if (acc[0][0][0] + acc[1][0][5] + acc[0][NB - 1][9] + acc[1][NB - 1][13] == 1.2345e-30f) p.y[0] = 1.f;
    return;
  }
#endif
  // (the loop's last barrier is behind every wave's last fragment read)
  if (sizeof(lds) >= (size_t)64 * (BN + 4) * 4 && conv_epilogue_vec_ok(p)) {
    halo_store_tile_lds<NB, GTH, GTW>(p, acc, n, ty, tx, n0, wm, wn, lane, reinterpret_cast<float*>(lds),
                                      (q.flags & 32) != 0);
    return;
  }
  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* part = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = wm * 64 + a * 32 + i;
      if (TH * TW < 128 && m >= TH * TW) continue;
      const int ho = ty * TH + m / TW, wo = tx * TW + m % TW;
      if (ho >= p.H || wo >= p.W) continue;
      const size_t row = (((size_t)n * p.H + ho) * p.W + wo) * p.Cout;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = n0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (j >= p.Cout) continue;
        float v = acc[a][b][r];
        if (part) {
          part[row + j] = v;
        } else {
          if (p.bias) v += p.bias[j];
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.mask) v = p.mask[row + j] > 0.f ? v : 0.f;
          p.y[row + j] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Round 6: the second half of a frozen ResNet bottleneck in ONE launch — conv2 (3x3 / stride 1, Cmid = 64 -> 64, folded
// BN, ReLU) -> conv3 (1x1, 64 -> 256, folded BN) + residual + ReLU (mmdet/models/backbones/resnet.py:239-266) — for the
// stride-4 map of layer1, where the two launches are a 66 us MFMA-bound halo conv followed by a 64 us HBM-bound 1x1
// (34 MB written and read back for the 64-channel intermediate, and the 1x1's operands delivered to 8400 workgroups).
//   phase 1  = conv3x3_halo_bfx4_kernel<1, 3> unchanged (8 x 16 pixels, patch split once per 16-channel chunk, filter
//              slices by LDS-DMA): the conv2 tile in acc[2][1] per wave;
//   phase 2  = its epilogue (bias, ReLU) through the LDS transpose — but instead of leaving for HBM every value is split
//              into its three bf16 planes and written to LDS as the A operand of conv3: [plane][k chunk][pixel][32 B],
//              halves swapped on odd 8-pixel groups (the B layout: conflict-free ds_read_b128);
//   phase 3  = conv3 on the workgroup's 128 pixels x ALL 256 output channels: wave w owns channels 64 w .. 64 w + 63
//              (acc3[4][2]: 128 accumulator registers), A fragments from LDS, the filter fragments of a k step straight
//              from L2 into registers (the 98 KB filter is shared by every workgroup; 24 16-byte loads per lane in all);
//   phase 4  = bias3 + residual + ReLU through the LDS transpose in two halves of 64 pixels, 16-byte loads / stores.
// Arithmetic: the planes of phase 2 are split3 of the SAME fp32 values the unfused conv2 stores, the k steps and the six
// plane products of phase 3 come in the ring kernel's order, the epilogue adds bias, then the residual, then clamps:
// the output is BIT-IDENTICAL to the two-launch chain (tests/test_gpu_det_ops.py).  LDS: operands 29.9 KB (later the
// phase-2 scratch) | planes 48 KB; the phase-4 scratch (65 KB) overlays both: 78 KB -> two workgroups per CU.
struct FusedC3Args {
  HaloBfxArgs h;          // conv2: x = [N,H,W,64], ws = its split filter, bias, relu
  const __bf16* ws3;      // conv3 split filter [3][KC3][Cout3][16]
  const float* bias3;     // [Cout3] or null
  const float* res;       // residual [N,H,W,Cout3] or null
  float* y;               // [N,H,W,Cout3]
  int Cout3, KC3, relu3;
};

__global__ __launch_bounds__(kThreads, 3) void conv3x3_c3_fused_bfx_kernel(FusedC3Args g) {
  const HaloBfxArgs& q = g.h;
  const ConvArgs& p = q.c;
  constexpr int NS = 3, TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = PH * PW;
  constexpr int AQ = PROWS * 4, AQT = (AQ + kThreads - 1) / kThreads;
  constexpr int BN = 64, CO3 = 256, KC3 = 4;
  // phase 1 keeps NO filter slices in LDS (round 6, after conv3x3_planes.hip): with 64 output channels a step is 12 MFMAs
  // per wave (384 cycles) — a DMA ring needed a `vmcnt` + workgroup barrier per step to hand each 6 KB slice over; the
  // waves now load their three 32-channel fragments per step straight from L2 into registers one step ahead, and the
  // only barriers left are the two around the patch store of a 16-channel chunk
  constexpr int A_OFF = 0;
  constexpr int HL = 32, A_PLANE = PROWS * HL;
  constexpr int OPER_BYTES = A_OFF + NS * A_PLANE;                  // 17,280
  constexpr int P3_CHUNK = 64 * 32, P3_PLANE = KC3 * P3_CHUNK;      // (64 pixels:) 2 KB per k chunk, 8 KB per plane
  constexpr int LD2 = BN + 4, LD4 = CO3 + 4;
  constexpr int S2_BYTES = 64 * LD2 * 4;                            // phase-2 transpose tile: 17,408
  constexpr int Y_OFF = (S2_BYTES + 1023) & ~1023;                  // conv3's A planes, behind it: 18,432 .. 43,008
  constexpr int Q4_BYTES = 32 * LD4 * 4;                            // phase-4 quarter tile: 33,280 (overlays both)
  constexpr int LDS_BYTES = (Y_OFF + NS * P3_PLANE) > OPER_BYTES ? (Y_OFF + NS * P3_PLANE) : OPER_BYTES;
  static_assert(Q4_BYTES <= LDS_BYTES && S2_BYTES <= Y_OFF, "scratch tiles overlay dead regions");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const unsigned* __restrict__ zero_page = q.zero;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m) return;                                   // workgroup-uniform (one channel tile)
  const int n = vtile / (q.tiles_y * q.tiles_x);
  const int trem = vtile - n * (q.tiles_y * q.tiles_x);
  const int ty = trem / q.tiles_x, tx = trem - ty * q.tiles_x;
  const int h0 = ty * TH - 1, w0 = tx * TW - 1;
  const int cchunks = p.Cin / 16;

  // ---- phase 1: conv2 = the accumulation order of conv3x3_halo_bfx4_kernel<1, 3> (16-channel chunks ascending, nine taps
  //      per chunk); wave (wm, wn) owns pixels 64 wm .. and channels 32 wn ..: its filter fragments by buffer loads
  const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(q.ws), 0, (int)((size_t)NS * q.KC * p.Cout * 32), 0x00020000);
  const int b2_lane = (((wave & 1) * 32 + (lane & 31)) * 16 + (lane >> 5) * 8) * 2;   // bytes
  const int b2_plane = q.KC * p.Cout * 32;
  auto load_b2 = [&](int kc, bf16x8 (&dst)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b2_rsrc, b2_lane, s * b2_plane + kc * p.Cout * 32, 0));
  };
  const float* a_src[AQT];
  int a_dst[AQT];
  bool a_use[AQT];
  f32x4 ra[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int idx = tid + kThreads * i;
    a_use[i] = idx < AQ;
    const int prow = a_use[i] ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = a_use[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_src[i] = in ? p.x + (((size_t)n * p.H + hi) * p.W + wi) * p.Cin + kq * 4
                  : reinterpret_cast<const float*>(g_zero_page);
    const int kq_off = (((kq >> 1) ^ ((pr + pc) & 1)) << 4) + (kq & 1) * 8;
    a_dst[i] = (in ? 1 : 0) | ((prow * HL + kq_off) << 1);
  }
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + ((a_dst[i] & 1) ? chunk * 16 : 0));
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      if (!a_use[i]) continue;
      u32x2 hh, mm, ll;
      split3(ra[i], hh, mm, ll);
      unsigned char* d = lds + A_OFF + (a_dst[i] >> 1);
      *reinterpret_cast<u32x2*>(d) = hh;
      *reinterpret_cast<u32x2*>(d + A_PLANE) = mm;
      *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = ll;
    }
  };
  const int frow = lane & 31, fk = lane >> 5;
  int a_frag[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = wm * 64 + a * 32 + frow;
    const int pr = m / TW, pc = m % TW;
#pragma unroll
    for (int par = 0; par < 2; ++par)
      a_frag[a][par] = A_OFF + (pr * PW + pc) * HL + ((fk ^ ((pr + pc + par) & 1)) << 4);
  }
  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 fbA[NS], fbB[NS];
  load_a(0);
  load_b2(0, fbA);                                                  // (tap 0, chunk 0)
  for (int c2 = 0; c2 < cchunks; c2 += 2) {                         // two chunks per trip: 18 steps, the ping-pong closes
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      const int chunk = c2 + t / 9, tap = t % 9;
      if (tap == 0) {
        __syncthreads();                                            // every wave is done with the previous chunk's patch
        store_a();                                                  // (waits for this chunk's patch loads)
        __syncthreads();
      }
      load_b2(tap < 8 ? (tap + 1) * cchunks + chunk : chunk + 1, (t & 1) ? fbA : fbB);    // (past the end: unused)
      if (tap == 0 && chunk + 1 < cchunks) load_a(chunk + 1);       // next patch: in registers under this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 (&fb)[NS] = (t & 1) ? fbB : fbA;
      const int tap_off = ((tap / 3) * PW + (tap % 3)) * HL;
      const int tap_par = (tap / 3 + tap % 3) & 1;
      bf16x8 fa[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[a][tap_par] + tap_off + s * A_PLANE);
#pragma unroll
      for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
        for (int i = 0; i <= tt; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[tt - i], acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                  // every wave is done with the patch (phase 2 overlays it)

  // ---- phases 2 - 4, once per HALF of the tile (64 pixels: the conv2 accumulators of the waves wm == h).  Per half:
  //      conv2 epilogue -> split planes of 64 pixels (24 KB) -> conv3 on 64 pixels x 256 channels (wave w: channels
  //      64 w .. 64 w + 63, acc3[2][2]) -> bias3 + residual + ReLU in two quarters of 32 pixels.  LDS never holds more
  //      than scratch2 (17 KB) + planes (24 KB) or a 33 KB quarter tile: 41.5 KB, three workgroups per CU.
  // Buffer addressing (`buffer_load / store_dwordx4 v, voffset, srsrc, soffset offen`): a resource descriptor in four
  // SGPRs, ONE 32-bit lane offset, the per-load constant as the scalar offset.  With flat pointers hipcc kept a 64-bit
  // vector address per load alive across the half loop (24 for the filter fragments alone) and spilled 49 - 65
  // registers to fit the three workgroups per CU.  Reads past the end of a descriptor return 0: a NULL residual is a
  // descriptor of 0 bytes, and rows below the image need no clamp.
  const int b3_lane = ((wave * 64 + frow) * 16 + fk * 8) * 2;       // bytes
  const __amdgpu_buffer_rsrc_t b3_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(g.ws3), 0, NS * KC3 * CO3 * 16 * 2, 0x00020000);
  auto load_b3 = [&](int kc, bf16x8 (&dst)[NS][2]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        dst[s][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   b3_rsrc, b3_lane, (((s * KC3 + kc) * CO3 + 32 * b) * 16) * 2, 0));
  };
  const size_t img_base = (size_t)n * p.H * p.W * CO3;              // wave-uniform
  const int img_bytes = p.H * p.W * CO3 * 4;                        // < 2^31 (checked by the launcher)
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.res ? g.res + img_base : reinterpret_cast<const float*>(g_zero_page)), 0,
      g.res ? img_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.y + img_base, 0, img_bytes, 0x00020000);
  float* scratch = reinterpret_cast<float*>(lds);
  const int c2_4 = (tid & 15) * 4, r2_0 = tid >> 4;                 // phase 2: 16 threads per pixel row, 16 rows per pass
  f32x4 bias2 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias2 = *reinterpret_cast<const f32x4*>(p.bias + c2_4);
  const int kc2 = c2_4 >> 4, kq2 = (c2_4 & 15) >> 2;
  const int c4 = (tid & 63) * 4, r0 = tid >> 6;                     // phase 4: 64 threads per pixel row, 4 rows per pass
  f32x4 bias3 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias3) bias3 = *reinterpret_cast<const f32x4*>(g.bias3 + c4);
  int a3_frag[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = a * 32 + frow;
    a3_frag[a] = Y_OFF + m * 32 + ((fk ^ ((m >> 3) & 1)) << 4);
  }
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    bf16x8 fb3[NS][2];
    load_b3(0, fb3);                                                // in flight under phase 2
    // ---- phase 2
    if (wm == h) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          scratch[i * LD2 + wn * 32 + (lane & 31)] = acc[a][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int i = r2_0 + ps * 16;                                 // pixel of the half
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD2 + c2_4);
      v += bias2;
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      u32x2 hh, mm, ll;
      split3(v, hh, mm, ll);
      unsigned char* d = lds + Y_OFF + kc2 * P3_CHUNK + i * 32 + (((kq2 >> 1) ^ ((i >> 3) & 1)) << 4) + (kq2 & 1) * 8;
      *reinterpret_cast<u32x2*>(d) = hh;
      *reinterpret_cast<u32x2*>(d + P3_PLANE) = mm;
      *reinterpret_cast<u32x2*>(d + 2 * P3_PLANE) = ll;
    }
    __syncthreads();
    // ---- phase 3
    f32x16 acc3[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[a][b][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC3; ++kc) {
      bf16x8 fa3[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          fa3[s][a] = *reinterpret_cast<const bf16x8*>(lds + a3_frag[a] + kc * P3_CHUNK + s * P3_PLANE);
#pragma unroll
      for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
        for (int i = 0; i <= tt; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc3[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[i][a], fb3[tt - i][b], acc3[a][b], 0, 0, 0);
      // the next k chunk's filter fragments into the same registers, behind this chunk's MFMAs (a second register
      // set for them costs the third workgroup per CU; the other workgroups' MFMAs cover the L2 latency)
      if (kc + 1 < KC3) load_b3(kc + 1, fb3);
    }
    __syncthreads();                                                // every wave is done with the planes
    // ---- phase 4: two quarters of 32 pixels
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      // the quarter's residual values: 8 unconditional 16-byte loads per thread (clamped coordinates / the zero page),
      // in flight under the scratch writes and the barrier (a load behind the bounds branch is waited for where it is
      // issued: 32 dependent HBM round trips per workgroup in the first version of this kernel)
      f32x4 rs[8];
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int m = h * 64 + a * 32 + r0 + ps * 4;
        const int ho = ty * TH + m / TW, wo = tx * TW + m % TW;     // (a column past the image reads a pixel nobody stores)
        rs[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, ((ho * p.W + wo) * CO3 + c4) * 4,
                                                                                 0, 0));
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
        const int i = r0 + ps * 4;
        const int m = h * 64 + a * 32 + i;
        const int ho = ty * TH + m / TW, wo = tx * TW + m % TW;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD4 + c4);
        v += bias3;
        if (g.res) v += rs[ps];                                    // (kept conditional: v + 0 could turn a -0 into +0)
        if (g.relu3) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (ho < p.H && wo < p.W)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, ((ho * p.W + wo) * CO3 + c4) * 4, 0, 0);
      }
      __syncthreads();
    }
  }
}


// Variant 7 ("wide pixel tile", round 5): a 16 x 16 = 256-pixel x 128-channel output tile per workgroup — twice the
// pixels per filter byte of variant 4.  Why: variant 4's step is co-limited by the matrix pipe and by the 12 KB
// filter slice every workgroup pulls through the L2 -> LDS path per tap (three workgroups per CU: 36 KB per 2304
// MFMA cycles = the ~16 B / clk / CU that `global_load_lds_dwordx4` delivers; timing-only ablation 0.77 -> 0.52 ms
// without that DMA, profiles/r8h).  Here a wave owns 128 pixels x 64 channels (acc[4][2]: 128 accumulator
// registers, two waves per SIMD = two workgroups per CU): 48 MFMAs per wave and step against the same 12 KB slice
// (8 B / clk / CU), half the barriers per MFMA, an 18 x 18 patch (1.27x halo instead of 1.41x) and one read of it
// per TWO of the old pixel tiles.  The accumulation order of every output element is variant 4's (chunk, tap,
// plane products in the same sequence): the results are bit-identical.
// What it costs is launch quantisation — 1092 units on 512 slots at the P2 level would be 2.13 rounds — and that is
// the SCHEDULER's job (launch_halo_wide below): whole rounds of these units in one launch, the image rows that are
// left over in a variant-4 launch of the old 128-pixel units (which fill a partial round far better).
template <int NS>
__device__ __forceinline__ void halo7_store_tile_lds(const ConvArgs& p, const f32x16 (&acc)[4][2], int n, int ty,
                                                     int tx, int n0, int wm, int wn, int lane, float* scratch,
                                                     bool nt_store = false) {
  constexpr int BN = 128, LD = BN + 4, TPR = BN / 4, RPP = kThreads / TPR;      // 32 threads per row, 8 rows per pass
  const int tid = threadIdx.x;
  const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
  const int j = n0 + c4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && j < p.Cout) bias = *reinterpret_cast<const f32x4*>(p.bias + j);
#pragma unroll
  for (int h = 0; h < 4; ++h) {                            // four quarters of 64 pixels: wave row h / 2, sub-tiles 2 (h % 2) + {0, 1}
    if (wm == (h >> 1)) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = a2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            scratch[i * LD + wn * 64 + b * 32 + (lane & 31)] = acc[2 * (h & 1) + a2][b][r];
          }
    }
    __syncthreads();
    if (j < p.Cout) {
#pragma unroll
      for (int ps = 0; ps < 64 / RPP; ++ps) {
        const int i = r0 + ps * RPP;
        const int m = h * 64 + i;
        const int ho = ty * 16 + (m >> 4), wo = tx * 16 + (m & 15);
        if (ho >= p.H || wo >= p.W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
        const size_t off = (((size_t)n * p.H + ho) * p.W + wo) * p.Cout + j;
        v += bias;
        if (p.relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (p.mask) {      // data gradient: ReLU backward of the conv's input
          const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + off);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = mk[t] > 0.f ? v[t] : 0.f;
        }
        if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.y + off));
        else *reinterpret_cast<f32x4*>(p.y + off) = v;
      }
    }
    if (h < 3) __syncthreads();
  }
}

// ABL (timing-only ablations, results are WRONG; tools/halo_wide_ablate.py): 1 = the second half's A fragments are
// not read (half 0's are reused), 2 = the patch is staged once and never again, 4 = no filter DMA after the first.
// FB32 (NS = 3): the filter slice travels as fp32 (the packed fourth section of the split-weight buffer: 8 KB per
// tap instead of the 12 KB of three bf16 planes) and every wave splits ITS fragment rows in registers in front of
// the MFMAs (split3 on the same values = the same planes: bit-identical).  Why: the launch is POWER-limited, and what
// the L2 -> CU path moves costs clock — PMC on the two-round map (profiles/r9b): 1.60 GHz with the filter DMA, 1.95
// GHz without it (timing-only ablation), while the cycle count moves by 8 % only.
template <int NS, int ABL = 0, bool FB32 = false>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_halo_bfx7_kernel(HaloBfxArgs q) {
  const ConvArgs& p = q.c;
  constexpr int PW = 18, PROWS = 18 * 18;                                   // 16 x 16 pixels + halo
  constexpr int AQ = PROWS * 4, AQT = (AQ + kThreads - 1) / kThreads;       // 1296 fp32 quads: 6 per thread (the 6th: 16 threads)
  constexpr int BN = 128, HL = 32;                                          // 32-byte patch pixels, k halves swapped on odd r + c
  static_assert(!FB32 || NS == 3, "the fp32 filter form is the fp32-faithful mode's");
  // FB32: filter rows of 16 fp32 = 64 bytes, the four 16-byte quads of row r XOR-swizzled by (r >> 2) & 3 on the DMA's
  // source side (the 16-lane groups of the fragment ds_read_b128 — rows {0-3, 12-15, 20-27} / ... — then cover the 64
  // banks exactly once)
  constexpr int B_PLANE = BN * 32, B_BUF = FB32 ? BN * 64 : NS * B_PLANE;
  constexpr int A_OFF = 2 * B_BUF;
  constexpr int A_PLANE = PROWS * HL;
  constexpr int A_SUB = 2 * PW * HL;                                        // sub-tile a -> a + 1: two patch rows down
  constexpr int OPER_BYTES = A_OFF + NS * A_PLANE, EPI_BYTES = 64 * (BN + 4) * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES];
  const unsigned* __restrict__ zero_page = q.zero;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int vlocal = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vlocal >= q.tile_count) return;                          // workgroup-uniform
  const int vtile = vlocal + q.tile_base;
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int n = tm / (q.tiles_y * q.tiles_x);
  const int trem = tm - n * (q.tiles_y * q.tiles_x);
  const int ty = trem / q.tiles_x, tx = trem - ty * q.tiles_x;
  const int h0 = ty * 16 - 1, w0 = tx * 16 - 1;                // input coords of patch (0, 0)
  const int n0 = tn * BN;
  const int cchunks = p.Cin / 16;

  // ---- filter DMA roles (variant 4's, NB = 2): wave w carries rows 32 w .. 32 w + 31 of every plane
  //      (FB32: two pieces of 16 fp32 rows each per wave, lane = 4 (row % 16) + quad slot)
  const int brow_d = FB32 ? wave * 32 + (lane >> 2) : wave * 32 + (lane >> 1);
  const int bhalf_d = (lane & 1) ^ ((brow_d >> 3) & 1);
  const bool b_okd = n0 + brow_d < p.Cout;
  const bool b_okd2 = n0 + brow_d + 16 < p.Cout;               // (FB32: the wave's second piece, rows + 16)
  const __bf16* b_lane = q.ws + (size_t)(b_okd ? n0 + brow_d : 0) * 16 + bhalf_d * 8;
  const size_t b_plane = (size_t)q.KC * p.Cout * 16;
  // fp32 section behind the three planes; the lane's source quad = its LDS slot ^ ((row >> 2) & 3) (same for row + 16)
  const float* b_lane32 = reinterpret_cast<const float*>(q.ws + 3 * b_plane) +
                          (size_t)(b_okd ? n0 + brow_d : 0) * 16 + (((lane & 3) ^ ((brow_d >> 2) & 3)) << 2);
  auto issue_b = [&](int chunk, int tap, int buf_off) {
    const size_t koff = (size_t)(tap * cchunks + chunk) * p.Cout * 16;
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page);
    if (FB32) {
      glds16(b_okd ? b_lane32 + koff : reinterpret_cast<const float*>(zp), lds + buf_off + wave * 2048);
      glds16(b_okd2 ? b_lane32 + koff + 16 * 16 : reinterpret_cast<const float*>(zp), lds + buf_off + wave * 2048 + 1024);
      return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
      glds16(b_okd ? b_lane + s * b_plane + koff : zp, lds + buf_off + s * B_PLANE + wave * 1024);
  };

  // ---- patch staging roles: patch quads idx = tid + 256 i; prow = idx / 4, kq = idx % 4
  unsigned a_off[AQT];                                         // element offset of the quad in x (chunk 0)
  int a_dst[AQT];                                              // bit 0: inside the image; bits 1..: LDS byte offset
  f32x4 ra[AQT];
#pragma unroll
  for (int i = 0; i < AQT; ++i) {
    const int idx = tid + kThreads * i;
    const bool use = idx < AQ;
    const int prow = use ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = use && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_off[i] = in ? (unsigned)((((size_t)n * p.H + hi) * p.W + wi) * p.Cin + kq * 4) : 0u;
    const int kq_off = (((kq >> 1) ^ ((pr + pc) & 1)) << 4) + (kq & 1) * 8;
    a_dst[i] = (in ? 1 : 0) | (use ? 2 : 0) | ((prow * HL + kq_off) << 2);
  }
  const bool nt_load = (q.flags & 16) != 0;                      // BGS_HALO_NT bit 0 (A/B)
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      const f32x4* src = reinterpret_cast<const f32x4*>((a_dst[i] & 1) ? p.x + a_off[i] + chunk * 16
                                                                        : reinterpret_cast<const float*>(g_zero_page));
      ra[i] = nt_load ? __builtin_nontemporal_load(src) : *src;
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AQT; ++i) {
      if (!(a_dst[i] & 2)) continue;
      u32x2 h, m, l;
      split3(ra[i], h, m, l);
      unsigned char* d = lds + A_OFF + (a_dst[i] >> 2);
      *reinterpret_cast<u32x2*>(d) = h;
      if (NS >= 2) *reinterpret_cast<u32x2*>(d + A_PLANE) = m;
      if (NS >= 3) *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = l;
    }
  };

  // ---- fragment roles: lane frow of sub-tile a (0..3) owns pixel m = 128 wm + 32 a + frow = patch (8 wm + 2 a + frow / 16,
  //      frow % 16) of tap (0, 0); a -> a + 1 moves two patch rows down (same parity of r + c)
  const int frow = lane & 31, fk = lane >> 5;
  int a_frag[2];                                               // [parity of the tap's dy + dx], sub-tile 0
  {
    const int pr = wm * 8 + (frow >> 4), pc = frow & 15;
#pragma unroll
    for (int par = 0; par < 2; ++par)
      a_frag[par] = A_OFF + (pr * PW + pc) * HL + ((fk ^ ((pr + pc + par) & 1)) << 4);
  }
  const int brow = wn * 64 + frow;                             // + 32 b: same 8-row-group parity
  const int b_frag = brow * 32 + ((fk ^ ((brow >> 3) & 1)) << 4);
  // FB32: the lane's eight k of row brow (+ 32 b: same (row >> 2) & 3) are quads 2 fk and 2 fk + 1 of the 64-byte row
  const int b_frag32_0 = brow * 64 + (((2 * fk) ^ ((brow >> 2) & 3)) << 4);
  const int b_frag32_1 = brow * 64 + (((2 * fk + 1) ^ ((brow >> 2) & 3)) << 4);

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (q.flags & 2) {
    // static priority by the wave's hardware slot on its SIMD (two waves per SIMD, one of each co-resident workgroup):
    // the odd slot always wins issue arbitration, the even one takes the matrix pipe whenever the odd one is between
    // its MFMA clusters — the two workgroups of a CU cannot fall into lockstep (both fetching fragments at once, the
    // pipe idle).  A/B: bgs_conv3x3_halo_bfx_tuning flags bit 1.
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_ID.WAVE_ID [3:0]
    if (hwid & 1u) __builtin_amdgcn_s_setprio(3);
  }
  load_a(0);
  issue_b(0, 0, 0);
  store_a();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0, nxt = B_BUF;                                    // byte offsets of the two B buffers
  for (int chunk = 0; chunk < cchunks; ++chunk) {
    const bool last_chunk = chunk + 1 >= cchunks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int rd = (tap & 1) ? nxt : cur;                     // buffer of this step
      const int wr = (tap & 1) ? cur : nxt;                     // buffer of the next step
      if (!(ABL & 4)) {
        if (tap < 8) issue_b(chunk, tap + 1, wr);               // next filter slice: DMA in flight
        else if (!last_chunk) issue_b(chunk + 1, 0, wr);
      }
      if (tap == 5 && !last_chunk && !(ABL & 2)) {                // next patch: held in registers over three steps
        __builtin_amdgcn_sched_barrier(0);
        load_a(chunk + 1);
      }
      const int tap_off = ((tap / 3) * PW + (tap % 3)) * HL;    // compile-time constants
      const int tap_par = (tap / 3 + tap % 3) & 1;
      bf16x8 fb[NS][2];
      if (FB32) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + rd + b_frag32_0 + b * 32 * 64);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + rd + b_frag32_1 + b * 32 * 64);
          u32x2 h0, m0, l0, h1, m1, l1;
          split3(v0, h0, m0, l0);
          split3(v1, h1, m1, l1);
          fb[0][b] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
          if (NS >= 2) fb[NS >= 2 ? 1 : 0][b] = __builtin_bit_cast(bf16x8, u32x4{m0[0], m0[1], m1[0], m1[1]});
          if (NS >= 3) fb[NS >= 3 ? 2 : 0][b] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
        }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            fb[s][b] = *reinterpret_cast<const bf16x8*>(lds + rd + b_frag + s * B_PLANE + b * 32 * 32);
      }
      bf16x8 fa[NS][2];
#pragma unroll
      for (int ah = 0; ah < 2; ++ah) {                          // two halves of the wave's 128 pixels: 24 fragment registers each
        if (!((ABL & 1) && ah == 1)) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
              fa[s][a] = *reinterpret_cast<const bf16x8*>(lds + a_frag[tap_par] + tap_off + (2 * ah + a) * A_SUB +
                                                          s * A_PLANE);
        }
        if (q.flags & 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_setprio(1);
        }
#pragma unroll
        for (int tt = NS - 1; tt >= 0; --tt)
#pragma unroll
          for (int i = 0; i <= tt; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                acc[2 * ah + a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[tt - i][b],
                                                                             acc[2 * ah + a][b], 0, 0, 0);
        if (q.flags & 1) __builtin_amdgcn_s_setprio(0);
      }
      if (tap == 5 && !last_chunk && !(ABL & 2) && (q.flags & 64))   // (as in variant 4: the patch loads fly one step longer)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AQT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // next slice (and patch) landed
      __syncthreads();
      if (tap == 8 && !last_chunk && !(ABL & 2)) {              // every wave is done with this patch
        store_a();
        __syncthreads();
      }
    }
    // nine steps per chunk: the buffer roles swap from chunk to chunk
    const int t = cur;
    cur = nxt;
    nxt = t;
  }
  // (the loop's last barrier is behind every wave's last fragment read)
  halo7_store_tile_lds<NS>(p, acc, n, ty, tx, n0, wm, wn, lane, reinterpret_cast<float*>(lds), (q.flags & 32) != 0);
}

int g_halo_last_nb = 0, g_halo_last_splits = 0, g_halo_last_variant = 0;
int g_halo_force_splits = -1, g_halo_variant = 4, g_halo_pf = 0, g_halo_padded = 0, g_halo_flags = 1;
// BGS_HALO_NT (A/B, read once): bit 0 = the patch loads, bit 1 = the output stores carry the non-temporal hint — the
// activations stream through once while every workgroup re-reads the filter slices from L2 (kernel flag bits 4 / 5)
int g_halo_nt = -1;
int halo_nt_flags() {
  if (g_halo_nt < 0) {
    const char* e = getenv("BGS_HALO_NT");
    g_halo_nt = e ? (atoi(e) & 7) : 0;                // (bit 2: counted wait behind the patch loads)
#ifdef BGS_ABLATE
    // timing-only (tools/halo_ablate2.py): BGS_HALO_ABL bit 0 = patch loaded once, bit 1 = patch split / stored once (and
    // no second barrier at the chunk boundary), bit 2 = no epilogue -> kernel flags 0x100 / 0x200 / 0x400
    const char* a = getenv("BGS_HALO_ABL");
    if (a) g_halo_nt |= (atoi(a) & 7) << 4;
#endif
  }
  return g_halo_nt << 4;
}

// Pixel-tile geometry of the v4 kernel for an H x W map: the instantiated tile with the fewest tiles per image
// (every tile costs the same 128 MFMA rows), 8 x 16 unless another one saves at least 5 % — mode 3; modes 0 / 1 / 2
// = 8 x 16 / 10 x 12 / 5 x 21 everywhere.
int g_halo_geom = -1;      // -1 default (8 x 16; BGS_HALO_GEOM) | 0 | 1 | 2 | 3 = fewest tiles per image (bgs_conv3x3_halo_bfx_tuning)
void halo_geom_dims(int g, int& th, int& tw) {
  th = g == 1 ? 10 : (g == 2 ? 5 : 8);
  tw = g == 1 ? 12 : (g == 2 ? 21 : 16);
}
int halo_bfx_geom(int H, int W) {
  // Default: 8 x 16 everywhere.  The small-map launches of cfg[1] are single-round (560-672 workgroups on 768 slots
  // after the channel-chunk split): their time is one workgroup's duration whatever the tile count, and the step did
  // not move with the tighter tiles (6.59-6.60 ms with either, interleaved runs on one box, round 4) — the tighter
  // tiles stay available (BGS_HALO_GEOM=auto | 1 | 2, the tuning hook) for grids that run several rounds.
  static const int env = [] {
    const char* e = getenv("BGS_HALO_GEOM");
    if (!e) return 0;
    if (e[0] == 'a') return 3;
    const int v = atoi(e);
    return v >= 0 && v <= 3 ? v : 0;
  }();
  const int mode = (g_halo_geom >= 0 && g_halo_geom <= 3) ? g_halo_geom : env;
  if (mode != 3) return mode;
  int best = 0;
  long long best_tiles = (long long)((H + 7) / 8) * ((W + 15) / 16);
  for (int g = 1; g <= 2; ++g) {
    int th, tw;
    halo_geom_dims(g, th, tw);
    const long long t = (long long)((H + th - 1) / th) * ((W + tw - 1) / tw);
    if (t * 100 <= best_tiles * 95 && (best == 0 || t < best_tiles)) {
      best = g;
      best_tiles = t;
    }
  }
  return best;
}
int g_halo_last_geom = 0;

// Variant 7 (wide pixel tile) dispatch: -1 = unset (env BGS_HALO_WIDE, default 0: see below) | 0 off | 1 automatic: layers with at
// least one whole round of 512 wide units, Cout % 128 == 0, no channel-chunk split | 2 every eligible layer
// (bgs_conv3x3_halo_bfx_tuning bits 24..27: value + 1).  g_halo_wide_tail: 0 = the left-over image rows as ONE variant-4
// launch (default) | k > 1 = that launch split k ways over the channel chunks (+ the split-K epilogue).
int g_halo_wide = -1;
int g_halo_last_wide_units = 0, g_halo_last_tail_units = 0;
// fp32 filter slices + in-register split (FB32): env BGS_HALO_FB32, default 0.  Read ONCE per process: the mode decides
// whether split-weight buffers carry the fp32 section the FB32 kernels DMA (bgs_conv_bfx_weight_bytes), so it cannot
// change between the split of a weight and its use.
int halo_fb32_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("BGS_HALO_FB32");
    mode = e ? (atoi(e) != 0) : 0;
  }
  return mode;
}
int halo_wide_mode() {
  if (g_halo_wide >= 0) return g_halo_wide;
  // Default OFF.  Measured (profiles/r9a, r9b, r9d): alone, the two-launch schedule takes the P2 layer from 0.741 to 0.710
  // ms (whole rounds: 17 % more work per second than variant 4) — but inside the cfg[1] step the launches that the
  // side streams place beside the P2 conv (small pyramid levels, functional.forked) already fill the slots variant 4
  // leaves idle in its third round, and the 256-register workgroups of variant 7 leave them no room: 6.209 (v4) vs 6.234
  // ms per step with the forks, 6.386 vs 6.334 without them.  The automatic mode stays for runs without side streams.
  static const int env = [] {
    const char* e = getenv("BGS_HALO_WIDE");
    if (!e) return 0;
    const int v = atoi(e);
    return v >= 0 && v <= 2 ? v : 0;
  }();
  return env;
}

int halo_bfx_plan(long long M, int tiles_m, int Cin, int Cout, int& nb) {
  nb = Cout <= 64 ? 1 : 2;
  const int tiles_n = (Cout + 64 * nb - 1) / (64 * nb);
  const long long wgs = (long long)tiles_m * tiles_n;
  const int cchunks = Cin / 16;
  // measured (profiles/r2g_bfx_sweep.txt; 768 resident workgroups): 68 workgroups -> 8 slices,
  // 132 -> 4, 263 -> 2, 526 -> 2, 1050 / 2100 -> 1
  int want = 1;
  if (wgs < 100) want = 8;
  else if (wgs < 200) want = 4;
  else if (wgs < 700) want = 2;
  if (want > cchunks / 2) want = cchunks / 2;
  if (want > 8) want = 8;
  if (g_halo_force_splits >= 1) want = g_halo_force_splits < cchunks ? g_halo_force_splits : cchunks;
  if (want < 1) want = 1;
  return want;
}

struct BfxKnobs {
  int tile = 0, splitk = -1, dma = 1, nst = 0, ring8 = 1, respf = 1;
  BfxKnobs() {
    if (const char* e = getenv("BGS_BFX_TILE")) tile = atoi(e);
    if (const char* e = getenv("BGS_BFX_SPLITK")) splitk = atoi(e);
    if (const char* e = getenv("BGS_BFX_NST")) nst = atoi(e);      // 3 | 4: ring depth of the 64 x 64 kernel
    if (const char* e = getenv("BGS_BF16_RING8")) ring8 = atoi(e);  // 0: bf16 mode on the 64 x 64 ring only
    if (const char* e = getenv("BGS_BFX_RESPF")) respf = atoi(e);   // 0: residual read in the epilogue (A/B)
  }
};
BfxKnobs& bfx_knobs() {
  static BfxKnobs k;
  return k;
}
int g_last_tile = 0, g_last_splits = 0, g_last_dma = 0, g_last_nst = 4;

// ring depth of the 64 x 64 DMA kernel: 3 stages (30 KB: five workgroups per CU) unless the tuning
// hook asks for the 4-stage ring (40 KB, four per CU).
int bfx_ring_stages(const BfxKnobs& knobs, long long wgs) {
  // measured (profiles/r3u_ring_stages_ab.txt): the 3-stage ring wins or ties on every layer of
  // cfg[1] — most where the grid sits just above a multiple of 1024 (fpn.lat1 0.078 -> 0.068,
  // l2.c1 0.044 -> 0.040, l4.b0.c1 0.076 -> 0.072), never behind by more than 1 %
  (void)wgs;
  return knobs.nst == 4 ? 4 : 3;
}

int g_ablate = 0;          // -DBGS_ABLATE builds only (tools/ablate.py): component-ablation timing

// tile (MB*10 + NB), K depth per barrier and split-K factor for a layer
void bfx_plan(long long M, int Cout, int KC, int& tile, int& bk, int& want) {
  const BfxKnobs& knobs = bfx_knobs();
  // measured on the cfg[1] shapes (profiles/r2a_bfx_sweep.txt): the 64x64 tile wins or ties on
  // every conv layer (these K steps are short: 6 MFMAs per wave per barrier, the grid matters more
  // than the tile); 128x128 only for the very deep reductions (fc1: K = 12544, 0.183 vs 0.253 ms)
  tile = (KC >= 512 && Cout >= 256) ? 22 : 11;
  if (knobs.tile == 22 || knobs.tile == 21 || knobs.tile == 12 || knobs.tile == 11) tile = knobs.tile;
  bk = 16;
  const int bm = tile / 10 * 64, bn = tile % 10 * 64;
  const long long wgs = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
  const int nk = KC;
  // re-measured with the LDS-transpose epilogue (profiles/r3l_splitk_resweep.txt): a slice must be
  // worth its slab traffic and the second launch — grids below 400 workgroups aim at ~1100, 400-600
  // split three ways only for K >= 2048, 600-1500 two ways only for K >= 1152; K < 512 never splits
  // (l2.c1 0.049 -> 0.042, l4.c3 0.056 -> 0.042, the RPN heads 0.018 -> 0.013 ms)
  want = 1;
  if (wgs < 400) want = (int)((1100 + wgs - 1) / wgs);
  else if (wgs < 600) want = nk >= 128 ? 3 : 1;
  else if (wgs < 1500) want = nk >= 72 ? 2 : 1;
  if (nk < 32) want = 1;
  if (want > nk / 8) want = nk / 8;
  if (want > 8) want = 8;
  if (knobs.splitk >= 1 && knobs.splitk <= 16) want = knobs.splitk < nk ? knobs.splitk : nk;
  if (want < 1) want = 1;
}

const unsigned* zero_page_device() {
  static const unsigned* ptr = nullptr;
  if (!ptr) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_zero_page)) == hipSuccess) ptr = (const unsigned*)a;
  }
  return ptr;
}

int launch_conv_bfx(BfxArgs& q, int up, hipStream_t st, void* workspace, size_t workspace_bytes) {
  ConvArgs& p = q.c;
  q.zero = zero_page_device();
  if (!q.zero) return BGS_ERR_LAUNCH;
  const BfxKnobs& knobs = bfx_knobs();
  bgs_internal_conv1x1_bfx_wide_clear_last();
  bgs_internal_conv1x1_planes_clear_last();
  const bool wide_ok = up == 1 && q.ns == 3 && knobs.tile == 0 && knobs.splitk < 0 && knobs.dma;
  if (wide_ok) {            // (the same preconditions: forward 1x1 layers of the fp32-faithful mode, no forced tile / split)
    const int rc = bgs_internal_conv1x1_planes(p, q.ws, q.KC, st);
    if (rc >= 0) {
      g_last_tile = 0x8000;        // bit 15: the planes-in-LDS 1x1 kernel ran
      g_last_splits = 1;
      g_last_dma = 0;
      return rc;
    }
  }
  if (wide_ok && p.R == 3 && p.stride == 2) {     // forward 3x3 / stride 2: 8 x 8 output pixels per workgroup, parity sub-grids in LDS
    bgs_internal_conv3x3_planes_clear_last();
    const int rc = bgs_internal_conv3x3s2_planes(p, q.ws, q.KC, st);
    if (rc >= 0) {
      g_last_tile = 0x8000;
      g_last_splits = 1;
      g_last_dma = 0;
      return rc;
    }
  }
  for (int pass = 0; pass < 2; ++pass) {
    // wide-N 1x1 layers: 128 x 128 tile with four M-stacked waves (conv_bfx_wide.hip); -1 = not eligible.
    // Pass 0 (ahead of the filter-resident kernel) only under its "every eligible layer" mode.
    if (wide_ok) {
      const int rc = bgs_internal_conv1x1_bfx_wide(p, q.ws, q.KC, workspace, workspace_bytes, pass == 0, st);
      if (rc >= 0) {
        g_last_tile = 0x4000;      // bit 14: the wide 1x1 kernel ran
        g_last_splits = 1;
        g_last_dma = 0;
        return rc;
      }
      p.partial = nullptr;
      p.kt_per_split = 0;
    }
    if (pass == 0 && up == 1 && knobs.tile == 0 && knobs.splitk < 0 && knobs.dma) {
      // short-reduction 1x1 layers with Cout % 256 == 0: filter resident in registers, activations
      // read once (conv1x1_bres.hip); -1 = not eligible
      p.partial = nullptr;
      p.kt_per_split = 0;
      const int rc = bgs_internal_conv1x1_bres(p, q.ws, q.KC, q.ns, q.zero, st);
      if (rc >= 0) {
        g_last_tile = 0x1000;      // bit 12: the filter-resident 1x1 kernel ran
        g_last_splits = 1;
        g_last_dma = 0;
        return rc;
      }
    }
  }
  const long long M = p.M;
  int tile, bk, want;
  bfx_plan(M, p.Cout, q.KC, tile, bk, want);
  // bf16 mode: the 8-wave 128 x 128 ring for the large layers (memory-path bound: half the bytes per MFMA)
  const bool ring8 = q.ns == 1 && up == 1 && knobs.tile == 0 && knobs.ring8 && M >= 2048 &&
                     p.Cout >= 256 && q.KC >= 16;
  if (ring8) {
    tile = 22;
    const long long wgs = ((M + 127) / 128) * ((p.Cout + 127) / 128);
    const int want_ring64 = want;     // the workspace query sizes the scratch by the 64 x 64 plan (and, for three
    want = 1;                         // planes, the wide kernel's): slice K only where that plan does
    if (wgs < 400) want = (int)((1100 + wgs - 1) / wgs);
    else if (wgs < 600 && q.KC >= 128) want = 2;
    if (want > q.KC / 8) want = q.KC / 8;
    if (want > 8) want = 8;
    if (want_ring64 <= 1) want = 1;
    if (knobs.splitk >= 1 && knobs.splitk <= 16) want = knobs.splitk < q.KC ? knobs.splitk : q.KC;
    if (want < 1) want = 1;
  }
  const int bm = tile / 10 * 64, bn = tile % 10 * 64;
  int splits = 1;
  p.partial = nullptr;
  p.kt_per_split = 0;
  if (!workspace) want = 1;
  while (want > 1 && (size_t)want * (size_t)M * p.Cout * sizeof(float) > workspace_bytes) --want;
  if (want > 1) {
    const int nk = q.KC;
    p.kt_per_split = (nk + want - 1) / want;
    splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
    p.partial = reinterpret_cast<float*>(workspace);
  }
  p.tiles_m = (int)((M + bm - 1) / bm);
  p.tiles_n = (p.Cout + bn - 1) / bn;
  p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;
  dim3 grid((unsigned)(8 * p.chunk), 1u, (unsigned)splits);
  g_last_tile = tile;
  g_last_splits = splits;
#define BFX_L(MB_, NB_, NS_, UP_) \
  hipLaunchKernelGGL((conv_igemm_bfx_kernel<MB_, NB_, 16, NS_, UP_>), grid, dim3(kThreads), 0, st, q)
#define BFX_T(MB_, NB_)                                                                  \
  do {                                                                                   \
    if (q.ns == 1) { if (up == 2) BFX_L(MB_, NB_, 1, 2); else BFX_L(MB_, NB_, 1, 1); }   \
    else { if (up == 2) BFX_L(MB_, NB_, 3, 2); else BFX_L(MB_, NB_, 3, 1); }             \
  } while (0)
  g_last_dma = 0;
  g_last_nst = 4;
  if (ring8) {
    g_last_dma = 1;
    g_last_nst = 3;
    bgs_internal_census_bump(BGS_CENSUS_BF16_RING8);
    if (p.R == 1 && p.S == 1 && p.pad == 0)
      hipLaunchKernelGGL((conv_igemm_bf16_ring8_kernel<true>), grid, dim3(512), 0, st, q);
    else
      hipLaunchKernelGGL((conv_igemm_bf16_ring8_kernel<false>), grid, dim3(512), 0, st, q);
  } else if (tile == 22) BFX_T(2, 2);
  else if (tile == 21) BFX_T(2, 1);
  else if (tile == 12) BFX_T(1, 2);
  else if (knobs.dma && q.ns == 1) {
    g_last_dma = 1;
    bgs_internal_census_bump(BGS_CENSUS_DMA_RING64);
    const bool p1x1 = up == 1 && p.R == 1 && p.S == 1 && p.pad == 0;
    if (knobs.nst != 3) {         // default: 4 x 6 KB, six workgroups per CU (cascade X101 bf16: 10.06 vs 10.15 ms)
      if (up == 2) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<2, false, 0, 1, 4>), grid, dim3(kThreads), 0, st, q);
      else if (p1x1) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, true, 0, 1, 4>), grid, dim3(kThreads), 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, false, 0, 1, 4>), grid, dim3(kThreads), 0, st, q);
    } else {                      // 3 x 6 KB: eight per CU
      g_last_nst = 3;
      if (up == 2) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<2, false, 0, 1, 3>), grid, dim3(kThreads), 0, st, q);
      else if (p1x1) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, true, 0, 1, 3>), grid, dim3(kThreads), 0, st, q);
      else hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, false, 0, 1, 3>), grid, dim3(kThreads), 0, st, q);
    }
  } else if (knobs.dma && q.ns == 3 && bfx_ring_stages(knobs, (long long)p.tiles_m * p.tiles_n * splits) == 3) {
    g_last_dma = 1;
    bgs_internal_census_bump(BGS_CENSUS_DMA_RING64);
    g_last_nst = 3;
    if (up == 2)
      hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<2, false, 0, 3, 3>), grid, dim3(kThreads), 0, st, q);
    else if (p.R == 1 && p.S == 1 && p.pad == 0)
      hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, true, 0, 3, 3>), grid, dim3(kThreads), 0, st, q);
    else
      hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, false, 0, 3, 3>), grid, dim3(kThreads), 0, st, q);
  } else if (knobs.dma && q.ns == 3) {
    g_last_dma = 1;
    bgs_internal_census_bump(BGS_CENSUS_DMA_RING64);
    const bool p1x1 = up == 1 && p.R == 1 && p.S == 1 && p.pad == 0;
    if (up == 2) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<2, false>), grid, dim3(kThreads), 0, st, q);
#ifdef BGS_ABLATE
    else if (p1x1 && g_ablate) {
#define ABL_D(A_) case A_: hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, true, A_>), grid, dim3(kThreads), 0, st, q); break;
      switch (g_ablate) { ABL_D(1) ABL_D(2) ABL_D(3) ABL_D(4) ABL_D(8) ABL_D(9) ABL_D(12) ABL_D(6) default: return BGS_ERR_UNSUPPORTED; }
#undef ABL_D
    }
#endif
    else if (p1x1) hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, true>), grid, dim3(kThreads), 0, st, q);
    else hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<1, false>), grid, dim3(kThreads), 0, st, q);
  } else BFX_T(1, 1);
#undef BFX_T
#undef BFX_L
  if (splits > 1) {
    if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
    return bgs_internal_conv_splitk_epilogue(p, splits, st);
  }
  BGS_RETURN_LAUNCH_STATUS();
}

// Stride-2 data gradient with parity-class rows (UP == 3 of the operand ring): q.c describes the
// zero-upsampled problem exactly as for UP == 2 (x = dy [N, H, W, Cin], y = dx [N, Ho, Wo, Cout], stride 1,
// pad = R - 1 - pad_fwd).  Eligible: even Ho / Wo, Cin % 16 == 0, the vectorised epilogue, residual mode 0 / 1.
// Returns -1 when not eligible (the caller takes the UP == 2 path).
int g_dgrad_parity = -1;     // BGS_DGRAD_S2_PARITY=0: the zero-upsampled form everywhere (A/B)
int launch_conv_bfx_dgrad_parity(BfxArgs& q, hipStream_t st) {
  ConvArgs& p = q.c;
  if (g_dgrad_parity < 0) {
    const char* e = getenv("BGS_DGRAD_S2_PARITY");
    g_dgrad_parity = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (!g_dgrad_parity) return -1;
  if ((p.Ho & 1) || (p.Wo & 1) || (p.Cin & 15) || p.R != p.S || (p.res_mode != 0 && p.res_mode != 1)) return -1;
  p.partial = nullptr;
  p.kt_per_split = 0;
  const uintptr_t al = (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.mask | (uintptr_t)p.x;
  if ((p.Cout & 3) || (al & 15)) return -1;
  q.zero = zero_page_device();
  if (!q.zero) return BGS_ERR_LAUNCH;
  p.rowmap = 1;
  p.rm_hh = p.Ho >> 1;
  p.rm_wh = p.Wo >> 1;
  p.rm_mc = p.N * p.rm_hh * p.rm_wh;
  p.rm_mcp = (p.rm_mc + 63) & ~63;
  if ((long long)4 * p.rm_mcp > 0x7fffffffLL) return -1;
  p.M = 4 * p.rm_mcp;                                   // virtual rows (the four classes, each padded to the tile)
  p.tiles_m = p.M / 64;
  p.tiles_n = (p.Cout + 63) / 64;
  p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;
  dim3 grid((unsigned)(8 * p.chunk));
  g_last_tile = 0x2000 | 11;                            // bit 13: the parity-class data-gradient order ran
  g_last_splits = 1;
  g_last_dma = 1;
  g_last_nst = 3;
  bgs_internal_census_bump(BGS_CENSUS_DMA_RING64);
  if (q.ns == 1)
    hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<3, false, 0, 1, 3>), grid, dim3(kThreads), 0, st, q);
  else
    hipLaunchKernelGGL((conv_igemm_bfx_dma_kernel<3, false, 0, 3, 3>), grid, dim3(kThreads), 0, st, q);
  BGS_RETURN_LAUNCH_STATUS();
}

inline int bfx_kc(int K) { return 2 * ((K + 31) / 32); }

}  // namespace

extern "C" size_t bgs_conv_bfx_weight_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0) return 0;
  // three bf16 planes [3][KC][rows][16]; in the FB32 experiment (BGS_HALO_FB32=1 at process start, default off) the
  // packed fp32 copy [KC][rows][16] behind them
  return (size_t)3 * bfx_kc(K) * rows * 16 * sizeof(__bf16) +
         (halo_fb32_mode() ? (size_t)bfx_kc(K) * rows * 16 * sizeof(float) : (size_t)0);
}

extern "C" int bgs_conv_bfx_split_weights(const float* w, void* out, int rows, int K,
                                          bgs_stream_t stream) {
  if (!w || !out || rows <= 0 || K <= 0) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)out % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int KC = bfx_kc(K);
  const size_t total = (size_t)KC * rows * 8;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(bfx_split_weights_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     w, reinterpret_cast<__bf16*>(out), rows, K, KC, 3, halo_fb32_mode());
  BGS_RETURN_LAUNCH_STATUS();
}

// w [Cout][R][S][Cin] (the forward filter) -> the split planes of its data-gradient filter
// (= bgs_conv_bfx_split_weights of the flipped, transposed filter [Cin][R][S][Cout]: rows = Cin,
// K = R*S*Cout; out: bgs_conv_bfx_weight_bytes(Cin, R*S*Cout) bytes) in ONE launch.
extern "C" int bgs_conv_bfx_split_weights_dgrad(const float* w, void* out, int Cout, int R, int S, int Cin,
                                                bgs_stream_t stream) {
  if (!w || !out || Cout <= 0 || R <= 0 || S <= 0 || Cin <= 0) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)out % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int KC = bfx_kc(R * S * Cout);
  const size_t total = (size_t)KC * Cin * 8;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(bfx_split_weights_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     w, reinterpret_cast<__bf16*>(out), Cout, R, S, Cin, KC, halo_fb32_mode());
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" size_t bgs_conv_bfx_workspace_bytes(long long M, int Cout, int K) {
  if (M <= 0 || Cout <= 0 || K <= 0) return 0;
  int tile, bk, want;
  bfx_plan(M, Cout, bfx_kc(K), tile, bk, want);
  const size_t ring = want > 1 ? (size_t)want * (size_t)M * Cout * sizeof(float) : 0;
  const size_t wide = bgs_internal_conv1x1_bfx_wide_workspace(M, Cout, K);   // (1x1 layers only; harmless otherwise)
  return ring > wide ? ring : wide;
}

// tuning / test hook: tile 0 = auto | 11 | 12 | 21 | 22; splitk -1 = auto | 1..16.
// Process-wide; not for concurrent use.
extern "C" void bgs_conv_bfx_tuning(int tile, int splitk) {
  BfxKnobs& k = bfx_knobs();
  k.tile = tile & 0xff;               // bit 8 set: the register-staged 64x64 kernel instead of the
  k.dma = (tile & 0x100) ? 0 : 1;     // LDS-DMA ring (A/B runs and tests of both)
  k.nst = (tile & 0x400) ? 3 : ((tile & 0x800) ? 4 : 0);   // bit 10 / 11: force the 3- / 4-stage ring
  k.respf = (tile & 0x1000) ? 0 : 1;  // bit 12: residual read in the epilogue instead of ahead of the K loop
  k.splitk = splitk;
}

extern "C" int bgs_conv_bfx_last_launch(int* tile, int* splits) {
  if (tile) *tile = g_last_tile | (g_last_dma ? 0x200 : 0) | (g_last_nst == 3 ? 0x400 : 0);   // bit 9: the LDS-DMA kernel ran; bit 10: its 3-stage (five workgroups / CU) instantiation
  if (splits) *splits = g_last_splits;
  return BGS_OK;
}

extern "C" int bgs_conv2d_nhwc_f32_bfx_ws(const float* x, const void* wsplit, const float* bias,
                                          const float* residual, float* y, int N, int H, int W,
                                          int Cin, int Cout, int R, int S, int stride, int pad,
                                          int relu, int residual_mode, int planes, void* workspace,
                                          size_t workspace_bytes, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!x || !wsplit || !y) return BGS_ERR_INVALID_ARG;
  if (planes != 1 && planes != 3) return BGS_ERR_INVALID_ARG;
  if (Cin % 4 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)wsplit) % 16 != 0) return BGS_ERR_INVALID_ARG;
  if (residual_mode < 0 || residual_mode > 2 || (residual_mode != 0 && !residual))
    return BGS_ERR_INVALID_ARG;
  BfxArgs q;
  q.ns = planes;
  ConvArgs& p = q.c;
  p.x = x; p.w = nullptr; p.bias = bias; p.res = residual; p.mask = nullptr; p.y = y;
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
  q.ws = reinterpret_cast<const __bf16*>(wsplit);
  q.KC = bfx_kc(p.K);
  q.res_prefetch = bfx_knobs().respf;
  return launch_conv_bfx(q, 1, (hipStream_t)stream, workspace, workspace_bytes);
}

// Data gradient (see bgs_conv2d_dgrad_nhwc_f32_ws): wt_split = split of the flipped, transposed
// filter [Cin][R][S][Cout] (rows = Cin, K = R*S*Cout).
extern "C" int bgs_conv2d_dgrad_nhwc_f32_bfx_ws(const float* dy, const void* wt_split,
                                                const float* residual, const float* mask, float* dx,
                                                int N, int H, int W, int Cin, int Cout, int R, int S,
                                                int stride, int pad, int residual_mode, int planes,
                                                void* workspace, size_t workspace_bytes,
                                                bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!dy || !wt_split || !dx) return BGS_ERR_INVALID_ARG;
  if (planes != 1 && planes != 3) return BGS_ERR_INVALID_ARG;
  if (stride != 1 && stride != 2) return BGS_ERR_UNSUPPORTED;
  if (Cout % 4 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)dy | (uintptr_t)wt_split) % 16 != 0) return BGS_ERR_INVALID_ARG;
  if (!(residual_mode == 0 || residual_mode == 1 || residual_mode == 3) ||
      (residual_mode != 0 && !residual))
    return BGS_ERR_INVALID_ARG;
  if (R != S || R - 1 - pad < 0) return BGS_ERR_UNSUPPORTED;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return BGS_ERR_INVALID_ARG;
  BfxArgs q;
  q.ns = planes;
  ConvArgs& p = q.c;
  p.x = dy; p.w = nullptr; p.bias = nullptr; p.res = residual; p.mask = mask; p.y = dx;
  p.N = N; p.H = Ho; p.W = Wo; p.Cin = Cout; p.Cout = Cin; p.R = R; p.S = S;
  p.stride = 1; p.pad = R - 1 - pad;
  p.Ho = H; p.Wo = W;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  p.M = (int)M;
  p.K = R * S * Cout;
  p.relu = 0;
  p.res_mode = residual_mode;
  q.ws = reinterpret_cast<const __bf16*>(wt_split);
  q.KC = bfx_kc(p.K);
  q.res_prefetch = bfx_knobs().respf;
  if (stride == 2) {       // rows grouped by output-pixel parity: only the taps that meet non-zeros are multiplied
    const int rc = launch_conv_bfx_dgrad_parity(q, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  return launch_conv_bfx(q, stride, (hipStream_t)stream, workspace, workspace_bytes);
}

// tuning / test hook: 0 = stride-2 data gradients in the zero-upsampled form, 1 = parity-class rows (default)
extern "C" void bgs_conv_dgrad_parity_enable(int on) { g_dgrad_parity = on ? 1 : 0; }

extern "C" size_t bgs_conv3x3_halo_bfx_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int th, tw;
  halo_geom_dims(g_halo_variant == 4 ? halo_bfx_geom(H, W) : 0, th, tw);
  const int tiles_m = N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
  int nb;
  const int want = halo_bfx_plan((long long)N * H * W, tiles_m, Cin, Cout, nb);
  return want > 1 ? (size_t)want * (size_t)N * H * W * Cout * sizeof(float) : 0;
}

extern "C" void bgs_conv3x3_halo_bfx_tuning(int splits, int variant) {
  g_ablate = (variant >> 8) & 0xff;          // timing-only ablation modes (-DBGS_ABLATE builds)
  const int variant0 = variant;
  g_halo_geom = ((variant >> 16) & 0xf) - 1; // 0: default pixel tile | 1: 8 x 16 | 2: 10 x 12 | 3: 5 x 21 | 4: fewest tiles
  if (g_halo_geom > 3) g_halo_geom = -1;
  variant &= 0xff;
  g_halo_force_splits = splits;
  // 1 = first version (2 workgroups / CU), 2 = taps unrolled + 3 workgroups / CU (register-staged
  // filter slices), 4 (and 0 = the default) = filter slices by LDS-DMA
  g_halo_variant = variant == 1 ? 1 : (variant == 2 ? 2 : 4);
  g_halo_pf = variant == 5 ? 1 : 0;        // 5 = variant 4 + A-fragment prefetch across the barrier
  // bits 20..23: 0 default (= 1) | 1: s_setprio 1 around the MFMA cluster | 2: static priority = hardware wave slot |
  // 8: none (A/B);  -DBGS_ABLATE builds: 4 / 8 drop two / all of the filter-plane DMAs (timing only)
  g_halo_flags = (variant0 >> 20) & 0xf;
  if (g_halo_flags == 0) g_halo_flags = 1;
  g_halo_padded = variant == 6 ? 1 : 0;    // 6 = variant 4 with the 48-byte patch rows (before the swizzled 32-byte pixels)
  // bits 24..27: 0 leave as is | 1 wide pixel tile off | 2 automatic | 3 every eligible layer
  const int wide = (variant0 >> 24) & 0xf;
  if (wide >= 1 && wide <= 3) g_halo_wide = wide - 1;
  else if (variant0 == 0 && splits < 0) g_halo_wide = -1;      // (the reset call: back to the environment's default)
}

// mode -1 = back to the environment's default | 0 off | 1 automatic | 2 every eligible layer; returns the previous mode
// (-1 when it was unset).  What train.TrunkPipeline switches while it has batches in flight: without side-stream forks
// beside the P2 convs the two-launch schedule of variant 7 is the faster one (5.14 -> 5.10 ms per step, profiles/r9h).
extern "C" int bgs_conv3x3_halo_bfx_wide(int mode) {
  const int prev = g_halo_wide;
  g_halo_wide = (mode >= 0 && mode <= 2) ? mode : -1;
  return prev;
}

extern "C" int bgs_conv3x3_halo_bfx_last_wide(int* wide_units, int* tail_units) {
  if (wide_units) *wide_units = g_halo_last_wide_units;      // 256-pixel x 128-channel units of the variant-7 launch (0: it did not run)
  if (tail_units) *tail_units = g_halo_last_tail_units;      // 128-pixel units of the variant-4 launch behind it
  return BGS_OK;
}

extern "C" int bgs_conv3x3_halo_bfx_last_launch(int* nb, int* splits) {
  if (nb) *nb = g_halo_last_nb | (g_halo_last_variant << 8) | (g_halo_last_geom << 16);   // bits 8..15: the kernel variant that ran; 16..: its pixel tile (0: 8 x 16, 1: 10 x 12, 2: 5 x 21)
  if (splits) *splits = g_halo_last_splits;
  return BGS_OK;
}

// 3x3 / stride 1 / pad 1, Cin % 16 == 0; wsplit = bgs_conv_bfx_split_weights of [Cout][3][3][Cin].
extern "C" int bgs_conv3x3_halo_nhwc_f32_bfx_ex(const float* x, const void* wsplit, const float* bias,
                                                const float* mask, float* y, int N, int H, int W,
                                                int Cin, int Cout, int relu, int planes, void* workspace,
                                                size_t workspace_bytes, bgs_stream_t stream);

extern "C" int bgs_conv3x3_halo_nhwc_f32_bfx(const float* x, const void* wsplit, const float* bias,
                                             float* y, int N, int H, int W, int Cin, int Cout,
                                             int relu, int planes, void* workspace,
                                             size_t workspace_bytes, bgs_stream_t stream) {
  return bgs_conv3x3_halo_nhwc_f32_bfx_ex(x, wsplit, bias, nullptr, y, N, H, W, Cin, Cout, relu, planes,
                                          workspace, workspace_bytes, stream);
}

// conv2 (3x3 / s1 / p1, Cmid -> Cmid, bias2, ReLU) -> conv3 (1x1, Cmid -> Cout3, bias3) + residual + ReLU in one
// launch (conv3x3_c3_fused_bfx_kernel): x [N,H,W,Cmid], w2split / w3split = bgs_conv_bfx_split_weights of the folded
// filters [Cmid][9 Cmid] / [Cout3][Cmid], residual [N,H,W,Cout3] or NULL, y [N,H,W,Cout3].  Supported: Cmid = 64,
// Cout3 = 256 (ResNet-50 layer1), fp32-faithful planes; BGS_ERR_UNSUPPORTED otherwise (callers run the two launches).
// Bit-identical to bgs_conv3x3_halo_nhwc_f32_bfx followed by bgs_conv2d_nhwc_f32_bfx_ws.
extern "C" int bgs_conv3x3_c3_fused_nhwc_f32_bfx(const float* x, const void* w2split, const float* bias2,
                                                 const void* w3split, const float* bias3, const float* residual,
                                                 float* y, int N, int H, int W, int Cmid, int Cout3, int relu3,
                                                 bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || !x || !w2split || !w3split || !y) return BGS_ERR_INVALID_ARG;
  if (Cmid != 64 || Cout3 != 256) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w2split | (uintptr_t)w3split | (uintptr_t)y | (uintptr_t)residual |
       (uintptr_t)bias2 | (uintptr_t)bias3) % 16 != 0)
    return BGS_ERR_UNSUPPORTED;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL || (long long)H * W * Cout3 > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;   // 32-bit offsets inside an image
  {
    // the planes form (bottleneck_tail_planes.hip: 8 x 8-pixel workgroups, the whole 64-channel patch staged once);
    // BGS_FUSED_C3_PLANES=0 keeps the 8 x 16-pixel kernel below (A/B, read at every call); bit-identical
    const char* e = getenv("BGS_FUSED_C3_PLANES");
    if (!e || atoi(e) != 0) {
      const int rc = bgs_internal_bottleneck_tail_planes(x, w2split, bias2, w3split, bias3, residual, y, N, H, W, relu3,
                                                         (hipStream_t)stream);
      if (rc >= 0) {
        bgs_internal_census_bump(BGS_CENSUS_FUSED_C3);
        return rc;
      }
    }
  }
  FusedC3Args g;
  HaloBfxArgs& q = g.h;
  q.ns = 3;
  ConvArgs& p = q.c;
  p.x = x; p.w = nullptr; p.bias = bias2; p.res = nullptr; p.mask = nullptr; p.y = nullptr;
  p.N = N; p.H = H; p.W = W; p.Cin = Cmid; p.Cout = Cmid; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1;
  p.Ho = H; p.Wo = W; p.M = (int)M; p.K = 9 * Cmid; p.relu = 1; p.res_mode = 0;
  p.partial = nullptr; p.kt_per_split = 0;
  q.ws = reinterpret_cast<const __bf16*>(w2split);
  q.KC = bfx_kc(p.K);
  q.tiles_y = (H + 7) / 8;
  q.tiles_x = (W + 15) / 16;
  p.tiles_m = N * q.tiles_y * q.tiles_x;
  p.tiles_n = 1;
  p.chunk = (p.tiles_m + 7) / 8;
  q.chunks_per_split = Cmid / 16;
  q.zero = zero_page_device();
  if (!q.zero) return BGS_ERR_LAUNCH;
  q.flags = g_halo_flags;
  g.ws3 = reinterpret_cast<const __bf16*>(w3split);
  g.bias3 = bias3;
  g.res = residual;
  g.y = y;
  g.Cout3 = Cout3;
  g.KC3 = bfx_kc(Cmid);
  g.relu3 = relu3;
  bgs_internal_census_bump(BGS_CENSUS_FUSED_C3);
  hipLaunchKernelGGL(conv3x3_c3_fused_bfx_kernel, dim3((unsigned)(8 * p.chunk)), dim3(kThreads), 0, (hipStream_t)stream, g);
  BGS_RETURN_LAUNCH_STATUS();
}

// ... with `mask` [N,H,W,Cout] or NULL: y = mask > 0 ? y : 0 in the epilogue — the DATA GRADIENT of a
// 3x3 / stride 1 / pad 1 conv is the same conv of dy with the flipped, transposed filter
// (x := dy [N,H,W,Cout_fwd], wsplit := split of wt [Cin_fwd][3][3][Cout_fwd], y := dx), and the mask is
// the ReLU backward of the forward conv's input (variant 4 only).
extern "C" int bgs_conv3x3_halo_nhwc_f32_bfx_ex(const float* x, const void* wsplit, const float* bias,
                                                const float* mask, float* y, int N, int H, int W,
                                                int Cin, int Cout, int relu, int planes, void* workspace,
                                                size_t workspace_bytes, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return BGS_ERR_INVALID_ARG;
  if (!x || !wsplit || !y) return BGS_ERR_INVALID_ARG;
  if (planes != 1 && planes != 3) return BGS_ERR_INVALID_ARG;
  if (Cin % 16 != 0) return BGS_ERR_UNSUPPORTED;
  if (mask && g_halo_variant != 4) return BGS_ERR_UNSUPPORTED;
  if ((uintptr_t)x % 16 != 0 || (uintptr_t)wsplit % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  HaloBfxArgs q;
  q.ns = planes;
  ConvArgs& p = q.c;
  p.x = x; p.w = nullptr; p.bias = bias; p.res = nullptr; p.mask = mask; p.y = y;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1;
  p.Ho = H; p.Wo = W; p.M = (int)M; p.K = 9 * Cin; p.relu = relu; p.res_mode = 0;
  p.partial = nullptr; p.kt_per_split = 0;
  q.ws = reinterpret_cast<const __bf16*>(wsplit);
  q.KC = bfx_kc(p.K);
  // the small maps (where the plan below would slice K): 8 x 8-pixel workgroups that own the whole reduction, patch
  // planes in LDS, ONE barrier per 32-channel chunk (conv3x3_planes.hip); -1 = not eligible / declined.  Not under a
  // forced slice count / variant / pixel tile (the tuning hook's A/B arms and the tests that assert them).
  bgs_internal_conv3x3_planes_clear_last();
  if (q.ns == 3 && g_halo_force_splits < 0 && g_halo_variant == 4 && !g_halo_pf && !g_halo_padded && g_halo_geom < 0 && !g_ablate) {
    const int rc = bgs_internal_conv3x3_planes(p, q.ws, q.KC, (hipStream_t)stream);
    if (rc >= 0) {
      g_halo_last_variant = 9;
      g_halo_last_nb = 0;
      g_halo_last_splits = 1;
      return rc;
    }
  }
  const int geom = (g_halo_variant == 4 && q.ns == 3) ? halo_bfx_geom(H, W) : 0;   // (the bf16 mode keeps 8 x 16)
  int gth, gtw;
  halo_geom_dims(geom, gth, gtw);
  g_halo_last_geom = geom;
  q.tiles_y = (H + gth - 1) / gth;
  q.tiles_x = (W + gtw - 1) / gtw;
  p.tiles_m = N * q.tiles_y * q.tiles_x;
  int nb;
  int want = halo_bfx_plan(M, p.tiles_m, Cin, Cout, nb);
  if (!workspace) want = 1;
  while (want > 1 && (size_t)want * (size_t)M * Cout * sizeof(float) > workspace_bytes) --want;
  const int cchunks = Cin / 16;
  int splits = 1;
  q.chunks_per_split = cchunks;
  if (want > 1) {
    q.chunks_per_split = (cchunks + want - 1) / want;
    splits = (cchunks + q.chunks_per_split - 1) / q.chunks_per_split;
    p.partial = reinterpret_cast<float*>(workspace);
  }
  p.tiles_n = (Cout + 64 * nb - 1) / (64 * nb);
  p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;
  g_halo_last_nb = nb;
  g_halo_last_splits = splits;
  g_halo_last_wide_units = g_halo_last_tail_units = 0;
  // ---- variant 7: whole rounds of 16 x 16-pixel units, then the left-over image rows on variant 4 (8 x 16 units)
  if (g_halo_variant == 4 && geom == 0 && !g_halo_padded && !g_halo_pf && splits == 1 && nb == 2 && Cout % 128 == 0 &&
      halo_wide_mode() > 0 && (long long)M * Cin < 0xffffffffLL) {
    constexpr int kSlots = 512;                                 // two workgroups per CU
    const int ty7 = (H + 15) / 16, tx7 = (W + 15) / 16;         // (tx7 == q.tiles_x: both tiles are 16 pixels wide)
    const int tn7 = Cout / 128;
    const long long units = (long long)N * ty7 * tx7 * tn7;
    long long rows_a = (long long)N * ty7;                      // tile rows (image-major) given to the wide launch
    if (halo_wide_mode() == 1) {
      const long long full = units / kSlots * kSlots;           // whole rounds
      rows_a = full / ((long long)tx7 * tn7);
      // (whatever is left goes to the variant-4 launch, however little: inside the wide launch even ONE unit past a
      //  whole round costs a full 144-step unit duration, ~3x what the 128-pixel units need for the same rows)
    }
    // (the bf16 mode, NS = 1: 1/6 of the MFMAs per filter byte-equivalent — its step is bound by patch staging and
    //  barriers, which two waves per SIMD hide worse than three: 0.209 vs 0.186 ms at the P2 level, profiles/r9a)
    if (rows_a > 0 && (halo_wide_mode() == 2 || (units >= kSlots && q.ns == 3))) {
      q.zero = zero_page_device();
      if (!q.zero) return BGS_ERR_LAUNCH;
      q.flags = g_halo_flags | halo_nt_flags();
      HaloBfxArgs qa = q;
      qa.tiles_y = ty7;
      qa.tiles_x = tx7;
      qa.c.tiles_m = N * ty7 * tx7;
      qa.c.tiles_n = tn7;
      qa.tile_base = 0;
      qa.tile_count = (int)(rows_a * tx7 * tn7);
      qa.c.chunk = (qa.tile_count + 7) / 8;
      g_halo_last_variant = 7;
      g_halo_last_wide_units = qa.tile_count;
      bgs_internal_census_bump(BGS_CENSUS_HALO_WIDE);
#define BGS_W7(A_) hipLaunchKernelGGL((conv3x3_halo_bfx7_kernel<3, A_>), dim3((unsigned)(8 * qa.c.chunk)), dim3(kThreads), 0, (hipStream_t)stream, qa)
      if (q.ns == 3 && halo_fb32_mode() && !g_ablate)
        hipLaunchKernelGGL((conv3x3_halo_bfx7_kernel<3, 0, true>), dim3((unsigned)(8 * qa.c.chunk)), dim3(kThreads), 0, (hipStream_t)stream, qa);
      else if (q.ns == 1)
        hipLaunchKernelGGL((conv3x3_halo_bfx7_kernel<1>), dim3((unsigned)(8 * qa.c.chunk)), dim3(kThreads), 0, (hipStream_t)stream, qa);
      else if (g_ablate == 1) BGS_W7(1);
      else if (g_ablate == 2) BGS_W7(2);
      else if (g_ablate == 4) BGS_W7(4);
      else if (g_ablate == 6) BGS_W7(6);
      else if (g_ablate == 7) BGS_W7(7);
      else BGS_W7(0);
#undef BGS_W7
      if (rows_a < (long long)N * ty7) {
        // the rest: image n_a from pixel row 16 r_a on, and every later image — contiguous in variant 4's tile order
        const int n_a = (int)(rows_a / ty7), r_a = (int)(rows_a - (long long)n_a * ty7);
        const long long tm4_begin = ((long long)n_a * q.tiles_y + 2 * r_a) * q.tiles_x;
        q.tile_base = (int)(tm4_begin * p.tiles_n);
        q.tile_count = p.tiles_m * p.tiles_n - q.tile_base;
        p.chunk = (q.tile_count + 7) / 8;
        g_halo_last_tail_units = q.tile_count;
        bgs_internal_census_bump(BGS_CENSUS_HALO_BFX4);
        if (q.ns == 1)
          hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 1>), dim3((unsigned)(8 * p.chunk)), dim3(kThreads), 0, (hipStream_t)stream, q);
        else
          hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3>), dim3((unsigned)(8 * p.chunk)), dim3(kThreads), 0, (hipStream_t)stream, q);
      }
      BGS_RETURN_LAUNCH_STATUS();
    }
  }
  dim3 grid((unsigned)(8 * p.chunk), 1u, (unsigned)splits);
  const bool v4 = g_halo_variant == 4;
  g_halo_last_variant = v4 ? 4 : (g_halo_variant == 1 && q.ns == 3 ? 1 : 2);
  if (v4) {
    q.zero = zero_page_device();
    if (!q.zero) return BGS_ERR_LAUNCH;
    q.flags = g_halo_flags | halo_nt_flags();
    bgs_internal_census_bump(BGS_CENSUS_HALO_BFX4);
    if (q.ns == 1) {
      if (nb == 1) hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<1, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
      else hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (geom == 1) {
      if (nb == 1) hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<1, 3, false, 10, 12>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
      else hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3, false, 10, 12>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (geom == 2) {
      if (nb == 1) hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<1, 3, false, 5, 21>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
      else hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3, false, 5, 21>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (nb == 1) {
      hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<1, 3>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (g_halo_padded) {
      hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3, false, 8, 16, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (g_halo_pf) {
      hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else {
      hipLaunchKernelGGL((conv3x3_halo_bfx4_kernel<2, 3>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    }
  } else if (g_halo_variant == 1 && q.ns == 3) {
    if (nb == 1)
      hipLaunchKernelGGL(conv3x3_halo_bfx_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    else
      hipLaunchKernelGGL(conv3x3_halo_bfx_kernel<2>, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
  } else {
    if (q.ns == 1) {
      if (nb == 1)
        hipLaunchKernelGGL((conv3x3_halo_bfx3_kernel<1, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
      else
        hipLaunchKernelGGL((conv3x3_halo_bfx3_kernel<2, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    } else if (nb == 1) {
      hipLaunchKernelGGL((conv3x3_halo_bfx3_kernel<1, 3>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
#ifdef BGS_ABLATE
    } else if (g_ablate) {
#define ABL_H(A_) case A_: hipLaunchKernelGGL((conv3x3_halo_bfx3_kernel<2, 3, A_>), grid, dim3(kThreads), 0, (hipStream_t)stream, q); break;
      switch (g_ablate) { ABL_H(1) ABL_H(2) ABL_H(3) ABL_H(4) ABL_H(5) ABL_H(6) ABL_H(8) ABL_H(16) ABL_H(18) ABL_H(26) ABL_H(30) ABL_H(14) default: return BGS_ERR_UNSUPPORTED; }
#undef ABL_H
#endif
    } else {
      hipLaunchKernelGGL((conv3x3_halo_bfx3_kernel<2, 3>), grid, dim3(kThreads), 0, (hipStream_t)stream, q);
    }
  }
  if (splits > 1) {
    if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
    return bgs_internal_conv_splitk_epilogue(p, splits, (hipStream_t)stream);
  }
  BGS_RETURN_LAUNCH_STATUS();
}
