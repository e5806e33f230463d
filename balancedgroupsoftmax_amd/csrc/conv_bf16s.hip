// bf16 STORAGE mode of cfg[4]: convolution / linear with bf16 activations in HBM (gfx950).
//
// The reference trains its X101 configurations under `Fp16OptimizerHook` + `wrap_fp16_model`
// (mmdet/core/fp16/hooks.py:11-127, decorators.py:8-160): every activation of the trunk is a HALF
// tensor in memory, products are half x half with fp32 accumulation, master weights stay fp32.
// The bf16 mode of conv_bfx.hip (NS = 1) reproduces the ARITHMETIC but keeps fp32 tensors in HBM
// and rounds the operands inside the kernel; this file is the storage half of that mode: the A
// operand arrives as bf16 NHWC, the result leaves as bf16 (or fp32 for the consumers outside the
// trunk: FPN laterals), bias / accumulate / residual add / ReLU in fp32.
//
// Why it pays: the bf16-mode 1x1 convs are bound by the CU's memory path, not by the matrix pipe
// (one MFMA per product: 46 launches of M = 8400, K = Cout = 1024 at 50-60 us = 300-350 TFLOP/s of
// 2500).  A bf16 A operand
//   * needs no conversion between LDS and the MFMA (the fragment is ONE ds_read_b128),
//   * halves the A bytes per K step, so a stage holds K = 32 in the bytes the fp32 ring spends on
//     K = 16: 16 KB per 32 MFMAs instead of 12 KB per 16, and HALF the barriers per MFMA,
//   * halves the epilogue's stores and residual loads and the next layer's HBM / L2 reads.
//
// conv_bf16s_kernel<P1X1, YBF>: 128 x 128 tile, EIGHT waves (4 x 2, each 32 x 64), K = 32 per stage:
//   * stage = A 128 rows x 64 B (32 bf16; the four 16-byte chunks XOR-swizzled by (row >> 2) & 3) +
//     B two 16-deep blocks x 128 rows x 32 B (the hi plane of bgs_conv_bfx_split_weights,
//     [K/16][Cout][16]; halves swapped on odd 8-row groups) = 16 KB; THREE stages (48 KB, three
//     workgroups = 24 waves per CU); 16 DMA pieces of 1 KB per stage = two per wave (A piece w,
//     B piece w), `global_load_lds_dwordx4`, one counted `s_waitcnt vmcnt(2)` + one raw `s_barrier`
//     per stage;
//   * out-of-range operands (image border, K tail, rows past M / Cout, stages past the end) come
//     from a zero page, so every wave issues the same number of DMAs per stage;
//   * epilogue through an LDS transpose in two halves of 64 rows: YBF: eight consecutive channels
//     per thread = one 16-byte bf16 store (+ one 16-byte residual load); fp32 out: four channels.
#include <stdlib.h>

#include "conv_args.h"

using namespace bgs_conv;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) unsigned g_bf16s_zero_page[16];

struct Bf16sArgs {
  const unsigned* zero;
  const __bf16* x;       // [N, H, W, Cin] bf16
  const __bf16* ws;      // hi plane [KC][Cout][16]
  const float* bias;     // [Cout] or null
  const void* res;       // residual: bf16 (res_bf16) or fp32; mode 1 same shape, 2 nearest-2x upsampled
  void* y;               // [N, Ho, Wo, Cout] bf16 (YBF) or fp32
  int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
  int M, K, KC;
  int relu, res_mode, res_bf16;
  int tiles_m, tiles_n, chunk;
  // tile order inside an XCD's band of row tiles (mpx row tiles per XCD): sub-bands of `sb` row tiles, inside a
  // sub-band column-tile-major — the workgroups in flight on an XCD (96) then share ONE sub-band of A (<= ~2 MB:
  // L2-resident, re-read by every column tile as L2 hits) instead of spanning a dozen row tiles whose A rows plus
  // the whole filter exceed the 4 MB of the XCD's L2.  sb == 0: row-tile-major (column tiles fastest).
  int mpx, sb;
};

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) {
  return __builtin_bit_cast(float, u & 0xffff0000u);
}

// Epilogue of a 128 x 128 tile held by eight waves (4 x 2, each 32 x 64 as two 32 x 32 accumulators):
// LDS transpose in two halves of 64 rows (scratch 64 x 132 floats at the start of `lds`, which no wave reads
// any more: callers have drained their DMAs / loads), then bias + residual + ReLU and 16-byte stores.
template <bool YBF>
__device__ __forceinline__ void bf16s_epilogue(const Bf16sArgs& p, const f32x16 (&acc)[2], unsigned char* lds,
                                               int m0, int n0, int wm, int wn, int lane, int tid) {
  // ---- epilogue through LDS, two halves of 64 rows: scratch 64 x 132 floats
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int LD = 132;
  constexpr int CPT = YBF ? 8 : 4;                       // channels per thread
  constexpr int TPR = 128 / CPT, RPP = 512 / TPR;        // threads per row, rows per pass
  const int c0 = (tid % TPR) * CPT, r0 = tid / TPR;
  const int j = n0 + c0;
  float bias[CPT];
#pragma unroll
  for (int t = 0; t < CPT; ++t) bias[t] = 0.f;
  if (p.bias && j < p.Cout) {
#pragma unroll
    for (int t = 0; t < CPT; t += 4) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + j + t);
      bias[t] = bv[0]; bias[t + 1] = bv[1]; bias[t + 2] = bv[2]; bias[t + 3] = bv[3];
    }
  }
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                     // ring (h = 0) / previous half (h = 1) no longer read
    if ((wm >> 1) == h) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (wm & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          scratch[i * LD + wn * 64 + b * 32 + (lane & 31)] = acc[b][r];
        }
    }
    __syncthreads();
    if (j >= p.Cout) continue;
#pragma unroll
    for (int ps = 0; ps < 64 / RPP; ++ps) {
      const int i = r0 + ps * RPP;
      const int m = m0 + h * 64 + i;
      if (m >= p.M) break;
      float v[CPT];
#pragma unroll
      for (int t = 0; t < CPT; t += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(scratch + i * LD + c0 + t);
        v[t] = s4[0] + bias[t]; v[t + 1] = s4[1] + bias[t + 1];
        v[t + 2] = s4[2] + bias[t + 2]; v[t + 3] = s4[3] + bias[t + 3];
      }
      if (p.res_mode) {
        size_t rrow = (size_t)m * p.Cout;
        if (p.res_mode == 2) {
          const int n = m / hw;
          const int rem = m - n * hw;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          rrow = (((size_t)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout;
        }
        if (p.res_bf16) {
          const __bf16* rp = reinterpret_cast<const __bf16*>(p.res) + rrow + j;
          if constexpr (CPT == 8) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              v[2 * t] += bf16_lo(rv[t]);
              v[2 * t + 1] += bf16_hi(rv[t]);
            }
          } else {
            const u32x2 rv = *reinterpret_cast<const u32x2*>(rp);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              v[2 * t] += bf16_lo(rv[t]);
              v[2 * t + 1] += bf16_hi(rv[t]);
            }
          }
        } else {
          const float* rp = reinterpret_cast<const float*>(p.res) + rrow + j;
#pragma unroll
          for (int t = 0; t < CPT; t += 4) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + t);
            v[t] += rv[0]; v[t + 1] += rv[1]; v[t + 2] += rv[2]; v[t + 3] += rv[3];
          }
        }
      }
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < CPT; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      if constexpr (YBF) {
        __bf16* yp = reinterpret_cast<__bf16*>(p.y) + (size_t)m * p.Cout + j;
        *reinterpret_cast<u32x4*>(yp) = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),
                                              pack_bf16(v[CPT - 4], v[CPT - 3]), pack_bf16(v[CPT - 2], v[CPT - 1])};
      } else {
        float* yp = reinterpret_cast<float*>(p.y) + (size_t)m * p.Cout + j;
        *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  }
}

// workgroup -> (row tile, column tile); false = no tile (grid padding)
__device__ __forceinline__ bool bf16s_tile(const Bf16sArgs& p, int& tm, int& tn) {
  if (p.sb == 0) {
    const int vtile = (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3));
    if (vtile >= p.tiles_m * p.tiles_n) return false;
    tm = vtile / p.tiles_n;
    tn = vtile - tm * p.tiles_n;
    return true;
  }
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int m_first = x * p.mpx;
  const int mc = min(p.mpx, p.tiles_m - m_first);          // row tiles of this XCD
  const int per = p.sb * p.tiles_n;
  const int j = i / per, r = i - j * per;
  const int ms = min(p.sb, mc - j * p.sb);                 // row tiles of sub-band j
  if (ms <= 0) return false;
  tn = r / ms;
  if (tn >= p.tiles_n) return false;
  tm = m_first + j * p.sb + (r - tn * ms);
  return true;
}

// NST: ring stages (NST - 1 in flight).  The loop is bound by Little's law, not by a bandwidth: operand bytes in
// flight per CU / the ~1.5-2 us an L2-missing DMA takes under load (tools/l2_stream_bench.hip) — NST = 3 keeps
// 32 KB per workgroup in flight (48 KB of LDS: three workgroups per CU), NST = 4 keeps 48 KB (64 KB: two).
// PF (with NST = 4): the fragments of stage kt + 1 are read from LDS while stage kt is multiplied from registers —
// the ds_read latency leaves the per-stage dependency chain (barrier -> DMA issue -> ds_read -> wait -> MFMA).
// ABL (only instantiated != 0 under -DBGS_ABLATE, tools/bf16s_ablate.py): timing-only variants of the NST = 3
// loop without 1: the MFMAs, 2: the DMA issue (after the prologue), 4: the fragment ds_reads, 8: the barrier.
template <bool P1X1, bool YBF, int NST = 3, bool PF = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_bf16s_kernel(Bf16sArgs p) {
  const unsigned* __restrict__ zero_page = p.zero;
  constexpr int A_BYTES = 128 * 64, B_BLOCK = 128 * 32, STAGE = A_BYTES + 2 * B_BLOCK;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);               // 0..7
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n;
  if (!bf16s_tile(p, tile_m, tile_n)) return;            // workgroup-uniform
  const int m0 = tile_m * 128, n0 = tile_n * 128;
  const int nk = p.KC >> 1;                              // KC is even (K padded to 32)

  // ---- A DMA role: row 16 wave + (lane >> 2); lane -> physical chunk, logical = physical ^ ((row >> 2) & 3)
  const int arow = wave * 16 + (lane >> 2);
  const int aq = (lane & 3) ^ ((arow >> 2) & 3);
  int a_hi0, a_wi0;
  const __bf16* a_base;
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
  // ---- B DMA role: block wave >> 2, rows 32 (wave & 3) + (lane >> 1)
  const int bkb = wave >> 2;
  const int brow_d = (wave & 3) * 32 + (lane >> 1);
  const int bhalf_d = (lane & 1) ^ ((brow_d >> 3) & 1);
  const bool b_ok = n0 + brow_d < p.Cout;
  const __bf16* b_ptr = p.ws + ((size_t)bkb * p.Cout + (b_ok ? n0 + brow_d : 0)) * 16 + bhalf_d * 8;
  int kg = aq * 8;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  const __bf16* a_ptr = nullptr;
  if (P1X1) a_ptr = a_base + ((size_t)a_hi0 * p.W + a_wi0) * p.Cin + kg;
  const size_t b_step = (size_t)p.Cout * 32;             // two 16-deep blocks per stage
  const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page);
  int kt_issue = 0, slot_issue = 0;                      // slot_issue = (kt_issue % NST) * STAGE, kept by rotation
  auto issue = [&]() {
    unsigned char* st = lds + slot_issue;
    slot_issue = slot_issue + STAGE == NST * STAGE ? 0 : slot_issue + STAGE;
    const bool live = kt_issue < nk;
    const __bf16* asrc;
    if (P1X1) {
      asrc = (live && a_ok && kg < p.K) ? a_ptr : zp;
      a_ptr += 32;
      kg += 32;
    } else {
      const int hi = a_hi0 + kr, wi = a_wi0 + ks;
      const bool ok = live && a_ok && kg < p.K && hi >= 0 && wi >= 0 && hi < p.H && wi < p.W;
      asrc = ok ? a_base + ((size_t)hi * p.W + wi) * p.Cin + kc : zp;
      kg += 32;
      kc += 32;
      while (kc >= p.Cin) {
        kc -= p.Cin;
        if (++ks == p.S) {
          ks = 0;
          ++kr;
        }
      }
    }
    if (!(ABL & 2) || kt_issue < NST - 1) {
      glds16(asrc, st + wave * 1024);
      glds16((live && b_ok) ? b_ptr : zp, st + A_BYTES + wave * 1024);
    }
    b_ptr += b_step;
    ++kt_issue;
  };

  // ---- fragment roles: wave tile = rows 32 wm .., columns 64 wn .. (two 32-wide sub-tiles)
  const int frow = lane & 31, fk = lane >> 5;
  const int ar = wm * 32 + frow;
  const int ac = (ar >> 2) & 3;
  const int a_off0 = ar * 64 + ((fk ^ ac) << 4);         // k block 0: chunks 0, 1
  const int a_off1 = ar * 64 + (((2 + fk) ^ ac) << 4);   // k block 1: chunks 2, 3
  const int br = wn * 64 + frow;                         // + 32 b keeps the 8-row-group parity
  const int b_off = A_BYTES + br * 32 + ((fk ^ ((br >> 3) & 1)) << 4);

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue();
  int slot_read = 0;
  if (PF) {
    static_assert(!PF || NST == 4, "fragment prefetch needs the 4-stage ring");
    bf16x8 fa[2], fb[2][2];
    auto read_frags = [&](const unsigned char* st) {
      fa[0] = *reinterpret_cast<const bf16x8*>(st + a_off0);
      fa[1] = *reinterpret_cast<const bf16x8*>(st + a_off1);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        fb[0][b] = *reinterpret_cast<const bf16x8*>(st + b_off + b * 1024);
        fb[1][b] = *reinterpret_cast<const bf16x8*>(st + b_off + B_BLOCK + b * 1024);
      }
    };
    // stage 0 landed -> registers
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nk > 0) read_frags(lds);
    slot_read = STAGE;
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt + 1 has landed once at most one younger stage (2 DMAs) is in flight; every wave has its stage
      // kt fragments in registers, so slot kt % 4 may be refilled (by stage kt + 4... issued below as kt + 3's
      // successor: the ring keeps three stages in flight)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue();
      const bf16x8 ca0 = fa[0], ca1 = fa[1], cb00 = fb[0][0], cb01 = fb[0][1], cb10 = fb[1][0], cb11 = fb[1][1];
      if (kt + 1 < nk) read_frags(lds + slot_read);        // stage kt + 1 -> registers, under the MFMAs below
      slot_read = slot_read + STAGE == NST * STAGE ? 0 : slot_read + STAGE;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca0, cb00, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca0, cb01, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca1, cb10, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca1, cb11, acc[1], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bf16s_epilogue<YBF>(p, acc, lds, m0, n0, wm, wn, lane, tid);
    return;
  }
  for (int kt = 0; kt < nk; ++kt) {
    // NST - 2 younger stages (2 DMAs each) may still be in flight
    if (ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NST == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    const unsigned char* st = lds + slot_read;
    slot_read = slot_read + STAGE == NST * STAGE ? 0 : slot_read + STAGE;
    bf16x8 fa0, fa1;
    if (ABL & 4) {
      fa0 = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, 1u, 2u, 3u});
      fa1 = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, 5u, 6u, 7u});
    } else {
      fa0 = *reinterpret_cast<const bf16x8*>(st + a_off0);
      fa1 = *reinterpret_cast<const bf16x8*>(st + a_off1);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      bf16x8 fb0, fb1;
      if (ABL & 4) {
        fb0 = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, (unsigned)b, 2u, 3u});
        fb1 = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, (unsigned)b, 6u, 7u});
      } else {
        fb0 = *reinterpret_cast<const bf16x8*>(st + b_off + b * 1024);
        fb1 = *reinterpret_cast<const bf16x8*>(st + b_off + B_BLOCK + b * 1024);
      }
      if (ABL & 1) {
        asm volatile("" ::"v"(fa0), "v"(fb0), "v"(fa1), "v"(fb1));
      } else {
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[b], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the (zero-page) tail DMAs

  bf16s_epilogue<YBF>(p, acc, lds, m0, n0, wm, wn, lane, tid);
}

// The same tile with the operands staged THROUGH REGISTERS (global_load_dwordx4 -> VGPR -> ds_write_b128)
// instead of by LDS-DMA.  bf16 operands need no conversion on the way, so the register path is a pure
// copy: two 16-byte loads + two 16-byte LDS writes per thread and stage, placed exactly where the DMA
// would put them (lane-linear 1 KB pieces, the swizzle in the source address).  Two LDS stage buffers;
// the loads of stage kt + 2 are in flight while stage kt is multiplied (raw s_barrier: a plain
// __syncthreads() would wait for them).  Why: the LDS-DMA path of a CU sustains about one 1 KB piece per
// ~64 cycles (16 B/clk: profiles/r5i_bf16s_ab.txt — 0.75 us per 16 KB stage at two workgroups per CU),
// a quarter of what the vector-memory return path delivers into registers.
template <bool P1X1, bool YBF>
__global__ __launch_bounds__(512, 2) void conv_bf16s_reg_kernel(Bf16sArgs p) {
  const unsigned* __restrict__ zero_page = p.zero;
  constexpr int NB = 2, A_BYTES = 128 * 64, B_BLOCK = 128 * 32, STAGE = A_BYTES + 2 * B_BLOCK;
  constexpr int LDS_BYTES = NB * STAGE > 64 * 132 * 4 ? NB * STAGE : 64 * 132 * 4;   // epilogue scratch
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n;
  if (!bf16s_tile(p, tile_m, tile_n)) return;
  const int m0 = tile_m * 128, n0 = tile_n * 128;
  const int nk = p.KC >> 1;

  const int arow = wave * 16 + (lane >> 2);
  const int aq = (lane & 3) ^ ((arow >> 2) & 3);
  int a_hi0, a_wi0;
  const __bf16* a_base;
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
  const int bkb = wave >> 2;
  const int brow_d = (wave & 3) * 32 + (lane >> 1);
  const int bhalf_d = (lane & 1) ^ ((brow_d >> 3) & 1);
  const bool b_ok = n0 + brow_d < p.Cout;
  const __bf16* b_ptr = p.ws + ((size_t)bkb * p.Cout + (b_ok ? n0 + brow_d : 0)) * 16 + bhalf_d * 8;
  int kg = aq * 8;
  int kc, kr, ks;
  {
    const int rs = kg / p.Cin;
    kc = kg - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  const __bf16* a_ptr = nullptr;
  if (P1X1) a_ptr = a_base + ((size_t)a_hi0 * p.W + a_wi0) * p.Cin + kg;
  const size_t b_step = (size_t)p.Cout * 32;
  const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page);
  int kt_issue = 0;
  u32x4 ra, rb;
  auto gload = [&]() {
    const bool live = kt_issue < nk;
    const __bf16* asrc;
    if (P1X1) {
      asrc = (live && a_ok && kg < p.K) ? a_ptr : zp;
      a_ptr += 32;
      kg += 32;
    } else {
      const int hi = a_hi0 + kr, wi = a_wi0 + ks;
      const bool ok = live && a_ok && kg < p.K && hi >= 0 && wi >= 0 && hi < p.H && wi < p.W;
      asrc = ok ? a_base + ((size_t)hi * p.W + wi) * p.Cin + kc : zp;
      kg += 32;
      kc += 32;
      while (kc >= p.Cin) {
        kc -= p.Cin;
        if (++ks == p.S) {
          ks = 0;
          ++kr;
        }
      }
    }
    // issued HERE (volatile asm keeps program order; hipcc otherwise sinks the loads below the MFMAs and
    // waits for them at the top of the next iteration); the consumer waits with an explicit vmcnt
    const __bf16* bsrc = (live && b_ok) ? b_ptr : zp;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra) : "v"(asrc) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rb) : "v"(bsrc) : "memory");
    b_ptr += b_step;
    ++kt_issue;
  };
  const int w_off = wave * 1024 + lane * 16;
  auto lwrite = [&](int buf) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra), "+v"(rb) : : "memory");
    *reinterpret_cast<u32x4*>(lds + buf * STAGE + w_off) = ra;
    *reinterpret_cast<u32x4*>(lds + buf * STAGE + A_BYTES + w_off) = rb;
  };

  const int frow = lane & 31, fk = lane >> 5;
  const int ar = wm * 32 + frow;
  const int ac = (ar >> 2) & 3;
  const int a_off0 = ar * 64 + ((fk ^ ac) << 4);
  const int a_off1 = ar * 64 + (((2 + fk) ^ ac) << 4);
  const int br = wn * 64 + frow;
  const int b_off = A_BYTES + br * 32 + ((fk ^ ((br >> 3) & 1)) << 4);

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  gload();                 // stage 0
  lwrite(0);
  gload();                 // stage 1 stays in registers
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int kt = 0; kt < nk; ++kt) {
    lwrite((kt + 1) & 1);  // stage kt + 1 (its buffer was last read in iteration kt - 1)
    gload();               // stage kt + 2: in flight under this iteration's MFMAs
    const unsigned char* st = lds + (kt & 1) * STAGE;
    const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(st + a_off0);
    const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(st + a_off1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(st + b_off + b * 1024);
      const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(st + b_off + B_BLOCK + b * 1024);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[b], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes and reads are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16s_epilogue<YBF>(p, acc, lds, m0, n0, wm, wn, lane, tid);
}

int g_bf16s_order = -1;      // BGS_BF16S_ORDER: 0 = row-tile-major, 1 = L2-sized sub-bands (auto), n > 1 = n row tiles per sub-band
int bf16s_order() {
  if (g_bf16s_order < 0) {
    const char* e = getenv("BGS_BF16S_ORDER");
    g_bf16s_order = e ? atoi(e) : 0;
    if (g_bf16s_order < 0 || g_bf16s_order > 4096) g_bf16s_order = 0;
  }
  return g_bf16s_order;
}

int g_bf16s_variant = -1;    // 0 = LDS-DMA ring (conv_bf16s_kernel), 1 = register-staged (conv_bf16s_reg_kernel)
int bf16s_variant() {
  if (g_bf16s_variant < 0) {
    const char* e = getenv("BGS_BF16S_VARIANT");
    g_bf16s_variant = e ? atoi(e) : 0;
    if (g_bf16s_variant < 0 || g_bf16s_variant > 3) g_bf16s_variant = 0;
  }
  return g_bf16s_variant;
}

const unsigned* bf16s_zero_page() {
  static const unsigned* ptr = nullptr;
  if (!ptr) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_bf16s_zero_page)) == hipSuccess) ptr = (const unsigned*)a;
  }
  return ptr;
}

int g_bf16s_launches = 0;

}  // namespace

// x bf16 [N, H, W, Cin]; w_hi = plane 0 of bgs_conv_bfx_split_weights (bf16(w), [ceil(K/32)*2][Cout][16]);
// residual_mode 0 | 1 (same shape) | 2 (nearest-2x upsampled: [N, Ho/2, Wo/2, Cout]); residual_bf16 /
// y_bf16: element type of `residual` / `y` (1 = bf16, 0 = fp32).
extern "C" int bgs_conv2d_nhwc_bf16s(const void* x, const void* w_hi, const float* bias,
                                     const void* residual, int residual_mode, int residual_bf16,
                                     void* y, int y_bf16, int N, int H, int W, int Cin, int Cout,
                                     int R, int S, int stride, int pad, int relu,
                                     bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!x || !w_hi || !y) return BGS_ERR_INVALID_ARG;
  if (residual_mode < 0 || residual_mode > 2 || (residual_mode != 0 && !residual))
    return BGS_ERR_INVALID_ARG;
  if (Cin % 8 != 0 || Cout % 8 != 0) return BGS_ERR_UNSUPPORTED;   // 16-byte chunks of bf16
  if (((uintptr_t)x | (uintptr_t)w_hi | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) % 16 != 0)
    return BGS_ERR_INVALID_ARG;
  Bf16sArgs p;
  p.zero = bf16s_zero_page();
  if (!p.zero) return BGS_ERR_LAUNCH;
  p.x = reinterpret_cast<const __bf16*>(x);
  p.ws = reinterpret_cast<const __bf16*>(w_hi);
  p.bias = bias;
  p.res = residual_mode ? residual : nullptr;
  p.y = y;
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
  p.KC = 2 * ((p.K + 31) / 32);
  p.relu = relu;
  p.res_mode = residual_mode;
  p.res_bf16 = residual_bf16;
  p.tiles_m = (int)((M + 127) / 128);
  p.tiles_n = (Cout + 127) / 128;
  p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;
  p.mpx = (p.tiles_m + 7) / 8;
  p.sb = 0;
  dim3 grid((unsigned)(8 * p.chunk));
  const int order = bf16s_order();
  if (order) {
    // sub-band = as many 128-row tiles as keep its A rows within ~2 MB (order > 1: that many tiles, forced)
    const long long row_bytes = (long long)p.K * 2;
    long long sb = order > 1 ? order : (2ll << 20) / (128 * row_bytes);
    if (sb < 1) sb = 1;
    if (sb > p.mpx) sb = p.mpx;
    p.sb = (int)sb;
    const int nsb = (p.mpx + p.sb - 1) / p.sb;
    grid = dim3((unsigned)(8 * nsb * p.sb * p.tiles_n));
  }
  hipStream_t st = (hipStream_t)stream;
  const bool p1x1 = R == 1 && S == 1 && pad == 0;
  ++g_bf16s_launches;
  bgs_internal_census_bump(BGS_CENSUS_BF16S);
#define BGS_BF16S_LAUNCH(KERNEL_)                                                                  \
  do {                                                                                            \
    if (y_bf16) {                                                                                 \
      if (p1x1) hipLaunchKernelGGL((KERNEL_<true, true>), grid, dim3(512), 0, st, p);             \
      else hipLaunchKernelGGL((KERNEL_<false, true>), grid, dim3(512), 0, st, p);                 \
    } else {                                                                                      \
      if (p1x1) hipLaunchKernelGGL((KERNEL_<true, false>), grid, dim3(512), 0, st, p);            \
      else hipLaunchKernelGGL((KERNEL_<false, false>), grid, dim3(512), 0, st, p);                \
    }                                                                                             \
  } while (0)
  if (bf16s_variant() == 1) BGS_BF16S_LAUNCH(conv_bf16s_reg_kernel);
  else if (bf16s_variant() == 2) {
    if (y_bf16) {
      if (p1x1) hipLaunchKernelGGL((conv_bf16s_kernel<true, true, 4>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((conv_bf16s_kernel<false, true, 4>), grid, dim3(512), 0, st, p);
    } else {
      if (p1x1) hipLaunchKernelGGL((conv_bf16s_kernel<true, false, 4>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((conv_bf16s_kernel<false, false, 4>), grid, dim3(512), 0, st, p);
    }
  } else if (bf16s_variant() == 3) {
    if (y_bf16) {
      if (p1x1) hipLaunchKernelGGL((conv_bf16s_kernel<true, true, 4, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((conv_bf16s_kernel<false, true, 4, true>), grid, dim3(512), 0, st, p);
    } else {
      if (p1x1) hipLaunchKernelGGL((conv_bf16s_kernel<true, false, 4, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((conv_bf16s_kernel<false, false, 4, true>), grid, dim3(512), 0, st, p);
    }
  } else
#ifdef BGS_ABLATE
  if (getenv("BGS_BF16S_ABLATE") && atoi(getenv("BGS_BF16S_ABLATE")) && p1x1 && y_bf16) {
#define BS_ABL(A_) case A_: hipLaunchKernelGGL((conv_bf16s_kernel<true, true, 3, false, A_>), grid, dim3(512), 0, st, p); break;
    switch (atoi(getenv("BGS_BF16S_ABLATE"))) { BS_ABL(1) BS_ABL(2) BS_ABL(4) BS_ABL(8) BS_ABL(3) BS_ABL(5) BS_ABL(6) BS_ABL(7) BS_ABL(12) BS_ABL(14) BS_ABL(15)
      default: return BGS_ERR_UNSUPPORTED; }
#undef BS_ABL
  } else
#endif
  BGS_BF16S_LAUNCH(conv_bf16s_kernel);
#undef BGS_BF16S_LAUNCH
  BGS_RETURN_LAUNCH_STATUS();
}

// tuning / test hook: 0 = LDS-DMA operand ring (default), 1 = register-staged operands.  Process-wide.
extern "C" void bgs_conv_bf16s_tuning(int variant) { g_bf16s_variant = (variant >= 0 && variant <= 3) ? variant : 0; }

// ---------------------------------------------------------------------------------------------
// Grouped 3x3 conv (conv2 of the ResNeXt bottleneck, resnext.py:47-57 + BN eval + ReLU) with bf16
// activations in and out; fp32 filter (rounded to bf16 once per wave), fp32 bias / accumulate.
// Same decomposition as grouped_conv.hip: a wave owns 16 output channels (one group, half a group
// or several whole groups with a block-diagonal filter), v_mfma_f32_16x16x16_bf16 per tap and
// 16-pixel sub-tile with the TRANSPOSED product (filter rows x pixel columns), so a lane ends up
// with four consecutive output channels of one pixel: one 8-byte bf16 store.
namespace {

typedef short bf16x4_bits __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4_bits gcs_pack_bf16(const f32x4 v) {
  return __builtin_bit_cast(bf16x4_bits, u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])});
}

// Stride-1 layers: LDS-resident patch.  Workgroup = 8 x 16 pixel tile x 64 channels; its 10 x 18
// pixel patch (180 pixels x 128 B = 23 KB, half of the fp32 kernel's) reaches LDS once by DMA.
//   * LDS layout: pixel PAIR pp at 256 pp bytes = 16 chunks of 16 B (chunk id = 8 * parity + channel
//     octet); chunk id sits in slot id ^ (pp % 9) — the pair's position within its 9-pair patch row, so the
//     address is AFFINE in the tile row (the first version keyed on pp & 15: hipcc materialised the 72
//     addresses of a wave in registers, 164 VGPRs, three waves per SIMD).  The DMA writes lane-linearly, so the swizzle is
//     applied to the SOURCE chunk each lane fetches.  A fragment read (ds_read_b64: 16 consecutive
//     pixels, the same channel quad; four quads per pixel = two adjacent chunks) touches every slot
//     exactly twice: 512 B in the minimal two LDS cycles.
constexpr int SGTH = 8, SGTW = 16, SGPH = SGTH + 2, SGPW = SGTW + 2, SGPIX = SGPH * SGPW;   // 180
constexpr int SGPAIRS = (SGPIX + 1) / 2;                                                      // 90
constexpr int SGDMA = (SGPAIRS + 3) / 4;                                                      // 23 pieces

template <int CG>
__global__ __launch_bounds__(256, CG >= 16 ? 3 : 5) void grouped_conv3x3_bf16s_lds_kernel(
    const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    __bf16* __restrict__ y, const unsigned* __restrict__ zero_page, int N, int H, int W, int C,
    int tiles_y, int tiles_x, int relu) {
  constexpr int KH = CG >= 16 ? CG / 16 : 1;       // 16-channel K slabs per tap
  __shared__ __attribute__((aligned(1024))) unsigned char lds[SGDMA * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, j = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int c0 = blockIdx.y * 64;                  // first channel of the slab
  const int h0 = ty * SGTH - 1, w0 = tx * SGTW - 1;

  // ---- patch DMA: piece d (1 KB) = pixel pairs 4 d .. 4 d + 3; lane -> (pair, slot)
  const __bf16* xn = x + (size_t)n * H * W * C + c0;
  for (int d = wave; d < SGDMA; d += 4) {
    const int pp = 4 * d + (lane >> 4);
    const int id = (lane & 15) ^ (pp % (SGPW / 2)); // source chunk of this slot (key: pair within the patch row)
    const int P = 2 * pp + (id >> 3);
    const int pr = P / SGPW, pc = P - pr * SGPW;
    const int hi = h0 + pr, wi = w0 + pc;
    const bool in = P < SGPIX && hi >= 0 && hi < H && wi >= 0 && wi < W;
    const __bf16* src = in ? xn + ((size_t)hi * W + wi) * C + 8 * (id & 7)
                           : reinterpret_cast<const __bf16*>(zero_page);
    glds16(src, lds + d * 1024);
  }

  // ---- this wave's filter fragments (registers, bf16) while the patch is in flight
  const int ct16 = c0 + wave * 16;
  const int n_out = ct16 + i;
  const int grp_first = ct16 / CG;
  const int in0 = (CG >= 16) ? grp_first * CG : ct16;
  bool w_live = true;
  int w_off = 4 * j;
  if (CG < 16) {
    const int g_in = (in0 + 4 * j) / CG, g_out = n_out / CG;
    w_live = g_in == g_out;
    w_off = (in0 + 4 * j) - g_in * CG;
  }
  bf16x4_bits bv[9][KH];
  const float* wrow = w + (size_t)n_out * 9 * CG;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      if (w_live) f = *reinterpret_cast<const f32x4*>(wrow + tap * CG + kh * 16 + w_off);
      bv[tap][kh] = gcs_pack_bf16(f);
    }
  f32x4 acc[SGTH];
#pragma unroll
  for (int a = 0; a < SGTH; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int qbase = (in0 - c0) / 4 + j;            // channel quad (within the slab) per K slab
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
      for (int a = 0; a < SGTH; ++a) {
        // pair index written as (row) * 9 + (column pair): the row term is a compile-time constant
        const int cp = (i + s2) >> 1, par = (i + s2) & 1;
        const int pp = (a + r) * (SGPW / 2) + cp;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          const int q = qbase + kh * 4;
          const int id = (par << 3) | (q >> 1);
          // swizzle key = the pair's position within its patch row, (i + s2) >> 1: independent of the
          // tile row a, so the eight reads of a tap differ by the compile-time offset 9 * 256 * a
          const bf16x4_bits av = *reinterpret_cast<const bf16x4_bits*>(
              lds + pp * 256 + ((id ^ cp) << 4) + ((q & 1) << 3));
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bv[r * 3 + s2][kh], av, acc[a], 0, 0, 0);
        }
      }
    }
  }
  const int c_out = ct16 + 4 * j;
  f32x4 bsv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bsv = *reinterpret_cast<const f32x4*>(bias + c_out);
  const int wo = tx * SGTW + i;
#pragma unroll
  for (int a = 0; a < SGTH; ++a) {
    const int ho = ty * SGTH + a;
    if (ho >= H || wo >= W) continue;
    f32x4 v = acc[a] + bsv;
    if (relu) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) v[tt] = fmaxf(v[tt], 0.f);
    }
    *reinterpret_cast<bf16x4_bits*>(y + (((size_t)n * H + ho) * W + wo) * C + c_out) = gcs_pack_bf16(v);
  }
}

// Any stride (the three stride-2 blocks of a ResNeXt trunk, and shapes the LDS kernel does not
// take): direct 8-byte loads per tap, 64 pixels x 16 channels per wave.
template <int CG>
__global__ __launch_bounds__(256) void grouped_conv3x3_bf16s_direct_kernel(
    const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    __bf16* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, int stride, int relu) {
  constexpr int KH = CG >= 16 ? CG / 16 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, j = lane >> 4;
  const int M = N * Ho * Wo;
  const int m0 = blockIdx.x * 64;
  const int ct = blockIdx.y * 4 + wave;
  if (ct * 16 >= C) return;
  const int n_out = ct * 16 + i;
  const int grp_first = (ct * 16) / CG;
  const int in0 = (CG >= 16) ? grp_first * CG : ct * 16;
  bool w_live = true;
  int w_off = 4 * j;
  if (CG < 16) {
    const int g_in = (in0 + 4 * j) / CG, g_out = n_out / CG;
    w_live = g_in == g_out;
    w_off = (in0 + 4 * j) - g_in * CG;
  }
  int hi0[4], wi0[4];
  const __bf16* xb[4];
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
        f32x4 f = {0.f, 0.f, 0.f, 0.f};
        if (w_live) f = *reinterpret_cast<const f32x4*>(wrow + (r * 3 + s) * CG + kh * 16 + w_off);
        const bf16x4_bits bvp = gcs_pack_bf16(f);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int hi = hi0[a] + r, wi = wi0[a] + s;
          bf16x4_bits av = {0, 0, 0, 0};
          if (ok[a] && hi >= 0 && wi >= 0 && hi < H && wi < W)
            av = *reinterpret_cast<const bf16x4_bits*>(xb[a] + ((size_t)hi * W + wi) * C + in0 +
                                                       kh * 16 + 4 * j);
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bvp, av, acc[a], 0, 0, 0);
        }
      }
    }
  }
  const int c_out = ct * 16 + 4 * j;
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
    *reinterpret_cast<bf16x4_bits*>(y + (size_t)m * C + c_out) = gcs_pack_bf16(v);
  }
}

// 3x3 / stride-2 / pad-1 max pooling of the fp32 stem output into the bf16 trunk (resnet.py:452);
// max commutes with the (monotone) rounding, so this equals rounding the fp32 pool.
__global__ __launch_bounds__(256) void maxpool3x3s2_f32_to_bf16_kernel(const float* __restrict__ x,
                                                                       __bf16* __restrict__ y, int N,
                                                                       int H, int W, int C, int Ho,
                                                                       int Wo) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
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
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + hi) * W + wi) * C + c4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = fmaxf(m[u], v[u]);
      }
    }
    *reinterpret_cast<bf16x4_bits*>(y + i * 4) = gcs_pack_bf16(m);
  }
}

}  // namespace

// x, y bf16 NHWC; w [C][3][3][C / groups] fp32 (folded BN); stride 1 | 2, pad 1.
extern "C" int bgs_grouped_conv3x3_nhwc_bf16s(const void* x, const float* w, const float* bias, void* y,
                                              int N, int H, int W, int C, int groups, int stride,
                                              int relu, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || (stride != 1 && stride != 2))
    return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (C % groups != 0 || C % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int cg = C / groups;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias) % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long long M = (long long)N * Ho * Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  __bf16* yb = reinterpret_cast<__bf16*>(y);
  bgs_internal_census_bump(BGS_CENSUS_GROUPED_BF16S);
  if (stride == 1 && C % 64 == 0) {
    const unsigned* zero = bf16s_zero_page();
    if (!zero) return BGS_ERR_LAUNCH;
    const int tiles_y = (H + SGTH - 1) / SGTH, tiles_x = (W + SGTW - 1) / SGTW;
    dim3 grid((unsigned)(N * tiles_y * tiles_x), (unsigned)(C / 64));
#define BGS_GS_LDS(CG_)                                                                              \
  hipLaunchKernelGGL((grouped_conv3x3_bf16s_lds_kernel<CG_>), grid, dim3(256), 0, st, xb, w, bias, yb, \
                     zero, N, H, W, C, tiles_y, tiles_x, relu)
    if (cg == 4) BGS_GS_LDS(4);
    else if (cg == 8) BGS_GS_LDS(8);
    else if (cg == 16) BGS_GS_LDS(16);
    else if (cg == 32) BGS_GS_LDS(32);
    else return BGS_ERR_UNSUPPORTED;
#undef BGS_GS_LDS
    BGS_RETURN_LAUNCH_STATUS();
  }
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((C / 16 + 3) / 4));
#define BGS_GS_DIRECT(CG_)                                                                          \
  hipLaunchKernelGGL((grouped_conv3x3_bf16s_direct_kernel<CG_>), grid, dim3(256), 0, st, xb, w, bias, \
                     yb, N, H, W, C, Ho, Wo, stride, relu)
  if (cg == 4) BGS_GS_DIRECT(4);
  else if (cg == 8) BGS_GS_DIRECT(8);
  else if (cg == 16) BGS_GS_DIRECT(16);
  else if (cg == 32) BGS_GS_DIRECT(32);
  else return BGS_ERR_UNSUPPORTED;
#undef BGS_GS_DIRECT
  BGS_RETURN_LAUNCH_STATUS();
}

// nn.MaxPool2d(3, 2, 1) of the fp32 stem output, written as bf16 (entry of the bf16 trunk).
extern "C" int bgs_maxpool3x3s2_nhwc_f32_to_bf16(const float* x, void* y, int N, int H, int W, int C,
                                                 bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !x || !y) return BGS_ERR_INVALID_ARG;
  if (C % 4 != 0 || ((uintptr_t)x | (uintptr_t)y) % 16 != 0) return BGS_ERR_UNSUPPORTED;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  size_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(maxpool3x3s2_f32_to_bf16_kernel, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, x, reinterpret_cast<__bf16*>(y), N, H, W, C, Ho, Wo);
  BGS_RETURN_LAUNCH_STATUS();
}
