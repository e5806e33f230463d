// 1x1 convolution / linear layer in bf16x6 arithmetic (conv_bfx.hip: fp32 tensors, every product from the
// exact three-way bf16 split of both operands, fp32 accumulate) on a 128 x (32 NBW) workgroup tile whose FOUR
// WAVES ARE STACKED ALONG M: wave w owns rows 32 w .. 32 w + 31 and ALL columns of the tile.
//
// Layers: the 1x1 convolutions of the ResNet bottlenecks, the projection shortcuts, the FPN laterals and the
// FC heads (mmdet/models/backbones/resnet.py:220-266, necks/fpn.py:101-141,
// bbox_heads/convfc_bbox_head.py:132-168) — 54 launches and 43 % of the cfg[1] step on the 64 x 64 operand ring
// (conv_igemm_bfx_dma_kernel), which runs the matrix pipe 0.30-0.39 busy.
//
// Why a different decomposition (profiles/r5_pmc_final_tree.md §2): the 64 x 64 ring issues
// ~190 instructions per K step and workgroup-wave for SIX MFMAs (the A split of a 32 x 16 fragment, the DMA
// address selects, waits and the barrier are paid per 32 x 32 output block), and moves 10 KB through the CU's
// vector-memory path per 24 MFMAs.  Here one wave multiplies a 32-row A fragment against the whole 128-column
// filter slice:
//   * 24 MFMAs per wave and barrier (12 for NBW = 2) for ONE fragment split (44 VALU), two A DMA pieces, three
//     B pieces and 14 ds_read_b128: ~3.5 instructions per MFMA instead of ~30;
//   * an A row block is read and split by exactly one wave: its LDS region is WAVE-PRIVATE (no barrier orders
//     it, the issuing wave's own vmcnt does), so the raw fragment of step k + 1 is read and split in the shadow
//     of step k's MFMAs — the split never sits between a barrier and the matrix pipe;
//   * 20 KB of DMA per 96 MFMAs (the 64 x 64 ring: 10 KB per 24): half the bytes per MFMA through TA / TD;
//   * operands arrive by `global_load_lds_dwordx4` only (no VGPR staging): A raw fp32, rows of 64 B with the
//     16-byte quads XOR-swizzled through the SOURCE addresses, B = the pre-split filter planes, rows of 32 B
//     with the halves swapped on odd 8-row groups (both layouts are conv_bfx.hip's, conflict-free by PMC);
//     rows past M / Cout are CLAMPED to the last valid row (their products are never stored): no zero page,
//     no selects in the loop;
//   * NST = 3: ring of three 20 KB stages (two workgroups per CU), counted vmcnt: stage k + 2 is issued at the
//     top of step k.  NST = 2: two stages (40 KB: three workgroups per CU), everything one step ahead,
//     vmcnt(0) per step (the schedule of the halo kernel's filter slices);
//   * same product order inside a K step and ascending K: results are BIT-IDENTICAL to the 64 x 64 ring
//     (tests/test_gpu_det_ops.py::test_bfx_wide_*);
//   * epilogue (bias, residual same-shape / nearest-2x-upsampled, ReLU, ReLU-backward mask, split-K slabs)
//     through an LDS transpose in two halves of 64 rows: 16-byte residual loads and stores.
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
// x (4 consecutive k) -> three planes of 4 packed bf16 each (conv_bfx.hip's split3: same values, same bits)
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

struct WideArgs {
  ConvArgs c;            // x, bias, res, mask, y, H, W, Cin, Ho, Wo, Cout, stride, M, K, relu, res_mode, partial, ..
  const __bf16* ws;      // split weights [3][KC][Cout][16]
  int KC;
  int flags;             // bit 0: static wave priority by residency slot; bit 1: staggered start; bit 2: the step's
                         // DMA pieces are issued between the two MFMA halves instead of behind the barrier
};

constexpr int W_A_WAVE = 32 * 64;            // a wave's private raw-A block: 32 rows x 16 fp32
constexpr int W_A_BYTES = 4 * W_A_WAVE;      // 8 KB

// ABL (only instantiated != 0 under -DBGS_ABLATE, tools/wide_ablate.py): timing-only variants that drop one
// component — 1: MFMAs, 2: DMA issue after the prologue, 4: the epilogue's global loads / stores, 8: the whole
// epilogue, 16: the A split.
// NBW = 4: 128 x 128 tile.  NBW = 2: 128 x 64 — twice the tiles (the stride-16 / 32 maps and the 64-channel layers,
// where 128 x 128 leaves the chip half empty) at 0.7 of the ring's operand bytes per MFMA and half its A splits.
template <int NBW, int NST, int ABL = 0>
__global__ __launch_bounds__(kThreads, NBW == 2 ? (NST == 3 ? 3 : 5) : (NST == 3 ? 2 : 3)) void conv1x1_bfx_wide_kernel(WideArgs q) {
  const ConvArgs& p = q.c;
  constexpr int BN = 32 * NBW;
  constexpr int B_PLANE = BN * 32;                 // one bf16 plane of a K step: BN rows x 32 B
  constexpr int B_BYTES = 3 * B_PLANE;
  constexpr int STAGE = W_A_BYTES + B_BYTES;       // NBW = 4: 20 KB; NBW = 2: 14 KB
  static_assert(NBW == 4 || NBW == 2, "one B piece = 32 rows: 12 (NBW = 4) or 6 (NBW = 2) pieces per stage");
  constexpr int EPI_BYTES = 64 * (BN + 4) * 4;
  constexpr int LDS_BYTES = NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;                  // workgroup-uniform
  const int m0 = (vtile / p.tiles_n) * 128, n0 = (vtile % p.tiles_n) * BN;
  const int nk_all = p.K >> 4;                                 // K % 16 == 0 (checked by the launcher)
  const int kt_begin = p.partial ? (int)blockIdx.z * p.kt_per_split : 0;
  const int kt_end = p.partial ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
  const int nk = kt_end - kt_begin;                            // >= 2 (launcher)
  // Co-resident workgroups start together and every step takes every workgroup the same time, so the DMA-issue,
  // fragment-read and MFMA phases of the workgroups of a CU coincide and their costs ADD (component ablation,
  // profiles/r8b_wide_tile_ablation.txt: fpn.lat0 134 us = skeleton 21 + MFMA 57 + DMA 20 + epilogue 39).  Block b
  // runs on XCD b % 8 and, within the XCD, the dispatcher fills the CUs round-robin: (b >> 8) is the residency
  // slot of a first-round workgroup.  A static priority per slot lets one workgroup's MFMA phase finish first, so
  // that its memory phase lies beside the other's MFMA phase from then on; a staggered start sets the same offset.
  {
    const int rslot = (int)((blockIdx.x >> 8) % 3);
    if (q.flags & 1) {
      if (rslot == 1) __builtin_amdgcn_s_setprio(1);
      else if (rslot == 2) __builtin_amdgcn_s_setprio(2);
    }
    if ((q.flags & 2) && rslot) {
      if (rslot == 1) __builtin_amdgcn_s_sleep(6);
      else __builtin_amdgcn_s_sleep(12);
    }
  }
  const bool late_issue = (q.flags & 4) != 0;

  // ---- A DMA role: piece j covers rows 16 j + (lane >> 2) of the wave's block; lane -> physical quad,
  //      logical quad = physical ^ ((row >> 2) & 3) (the XOR swizzle of conv_bfx.hip's ring, on the source side)
  const float* a_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rw = 16 * j + (lane >> 2);
    const int aq = (lane & 3) ^ ((rw >> 2) & 3);
    int m = m0 + wave * 32 + rw;
    m = m < p.M ? m : p.M - 1;                                 // clamped: rows past M are never stored
    const int hw = p.Ho * p.Wo;
    const int n = m / hw;
    const int rem = m - n * hw;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_ptr[j] = p.x + (((size_t)n * p.H + (size_t)ho * p.stride) * p.W + (size_t)wo * p.stride) * p.Cin +
               (size_t)kt_begin * 16 + aq * 4;
  }
  // ---- B DMA role.  NBW = 4: wave w carries rows 32 w .. 32 w + 31 of the three planes (three pieces).  NBW = 2:
  //      six pieces (plane j / 2, rows 32 (j % 2) ..): wave w takes piece w and, w < 2, piece w + 4.
  //      Halves swapped on odd 8-row groups.
  const size_t b_plane = (size_t)q.KC * p.Cout * 16;           // elements between planes
  const size_t b_step = (size_t)p.Cout * 16;                   // elements between K steps
  const int nb_mine = NBW == 4 ? 3 : (wave < 2 ? 2 : 1);       // wave-uniform
  const __bf16* b_ptr[NBW == 4 ? 3 : 2];
  int b_dst[NBW == 4 ? 3 : 2];
#pragma unroll
  for (int i = 0; i < (NBW == 4 ? 3 : 2); ++i) {
    const int plane = NBW == 4 ? i : ((wave + 4 * i) >> 1);
    const int rblk = NBW == 4 ? wave : (wave & 1);
    const int row = rblk * 32 + (lane >> 1);
    const int half = (lane & 1) ^ ((row >> 3) & 1);
    int nrow = n0 + row;
    nrow = nrow < p.Cout ? nrow : p.Cout - 1;                  // clamped: columns past Cout are never stored
    b_ptr[i] = q.ws + (size_t)(plane < 3 ? plane : 0) * b_plane + ((size_t)kt_begin * p.Cout + nrow) * 16 + half * 8;
    b_dst[i] = W_A_BYTES + plane * B_PLANE + rblk * 1024;
  }

  auto issue_a = [&](int slot) {                               // the wave's two raw-A pieces of the next step
    unsigned char* dst = lds + slot * STAGE + wave * W_A_WAVE;
    glds16(a_ptr[0], dst);
    glds16(a_ptr[1], dst + 1024);
    a_ptr[0] += 16;
    a_ptr[1] += 16;
  };
  auto issue_b = [&](int slot) {                               // the wave's filter pieces of the next step
    unsigned char* dst = lds + slot * STAGE;
#pragma unroll
    for (int i = 0; i < (NBW == 4 ? 3 : 2); ++i) {
      if (i < nb_mine) glds16(b_ptr[i], dst + b_dst[i]);
      b_ptr[i] += b_step;
    }
  };
  // s_waitcnt vmcnt(n_fixed + k * nb_mine) with the wave's own piece count
  auto wait_vm = [&](int fixed, int kb) {
    const int n = fixed + kb * nb_mine;
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
  };

  // ---- fragment roles
  const int frow = lane & 31, fk = lane >> 5;
  const int ac = (frow >> 2) & 3;
  const int a_off0 = wave * W_A_WAVE + frow * 64 + (((2 * fk) ^ ac) << 4);
  const int a_off1 = wave * W_A_WAVE + frow * 64 + (((2 * fk + 1) ^ ac) << 4);
  const int b_off = W_A_BYTES + frow * 32 + ((fk ^ ((frow >> 3) & 1)) << 4);   // + 1024 b: same 8-row-group parity

  f32x16 acc[NBW];
#pragma unroll
  for (int b = 0; b < NBW; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  auto read_raw = [&](int slot, f32x4& r0, f32x4& r1) {
    const unsigned char* st = lds + slot * STAGE;
    r0 = *reinterpret_cast<const f32x4*>(st + a_off0);
    r1 = *reinterpret_cast<const f32x4*>(st + a_off1);
  };
  auto split_frag = [&](const f32x4 r0, const f32x4 r1, bf16x8 (&fa)[3]) {
    if (ABL & 16) {
      fa[0] = __builtin_bit_cast(bf16x8, r0);
      fa[1] = __builtin_bit_cast(bf16x8, r1);
      fa[2] = __builtin_bit_cast(bf16x8, r0 + r1);
      return;
    }
    u32x2 h0, m0_, l0, h1, m1, l1;
    split3(r0, h0, m0_, l0);
    split3(r1, h1, m1, l1);
    fa[0] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    fa[1] = __builtin_bit_cast(bf16x8, u32x4{m0_[0], m0_[1], m1[0], m1[1]});
    fa[2] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
  };
  // B fragments and MFMAs in two column halves (blocks 0,1 then 2,3): 24 instead of 48 fragment registers
  // live, the product order of every accumulator unchanged
  auto read_b = [&](int slot, int half, bf16x8 (&fb)[3][2]) {
    const unsigned char* st = lds + slot * STAGE + b_off + half * 2048;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[s][b] = *reinterpret_cast<const bf16x8*>(st + s * B_PLANE + b * 1024);
  };
  auto mma = [&](const bf16x8 (&fa)[3], const bf16x8 (&fb)[3][2], int half) {
    // products (i, j), i + j <= 2, smallest terms first (conv_bfx.hip's order); consecutive MFMAs alternate
    // between the two accumulators of the half
    if (ABL & 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(fb[i][b]));
      }
      return;
    }
#pragma unroll
    for (int t = 2; t >= 0; --t)
#pragma unroll
      for (int i = 0; i <= t; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[2 * half + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[t - i][b], acc[2 * half + b], 0, 0, 0);
  };

  bf16x8 fa0[3], fa1[3];
  int kt_a = 0, kt_b = 0;                                      // K steps whose A / B pieces have been issued
  int slot = 0;                                                // ring slot of step k (NST == 3)
  if (NST == 3) {
    issue_a(0); issue_b(0);
    issue_a(1); issue_b(1);
    kt_a = kt_b = 2;
    wait_vm(2, 2);                                             // my A pieces of step 0 (the two oldest; 2 + 2 nb younger)
  } else {
    issue_a(0); issue_b(0); issue_a(1);
    kt_a = 2; kt_b = 1;
    wait_vm(2, 1);                                             // my A pieces of step 0 (B(0) and A(1) younger)
  }
  {
    f32x4 r0, r1;
    read_raw(0, r0, r1);
    split_frag(r0, r1, fa0);
  }

  // One K step: B(k) and raw A(k + 1) are read behind the step's barrier; the MFMAs of step k run on the
  // fragment split one step earlier while the VALU splits the next one.
  auto step = [&](int k, const bf16x8 (&fa)[3], bf16x8 (&fa_next)[3]) {
    int s_cur, s_nxt;
    if (NST == 3) {
      // stage k complete, A of stage k + 1 landed (mine); lgkmcnt(0): the fragment reads of step k - 1 have
      // RETURNED before this wave arrives — behind the barrier other waves (and this one) refill what they read
      wait_vm(0, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // .. and everyone's: B(k) is visible; B(k - 1)'s slot is free
      asm volatile("" ::: "memory");
      s_cur = slot;
      s_nxt = slot + 1 == 3 ? 0 : slot + 1;
      slot = s_nxt;
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // B(k), A(k + 1): issued one step ago; reads returned
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      s_cur = k & 1;
      s_nxt = s_cur ^ 1;
    }
    auto issue_next = [&]() {
      if (ABL & 2) return;
      if (NST == 3) {
        const int s_iss = s_nxt + 1 == 3 ? 0 : s_nxt + 1;      // slot of stage k + 2 = slot of stage k - 1
        if (kt_a < nk) {
          issue_a(s_iss);
          issue_b(s_iss);
          ++kt_a;
        }
      } else {
        if (kt_b < nk) {                                       // B(k + 1) -> the slot B(k - 1) left
          issue_b(s_nxt);
          ++kt_b;
        }
        if (kt_a < nk) {                                       // A(k + 2) -> the block raw A(k) left one step ago
          issue_a(s_cur);
          ++kt_a;
        }
      }
    };
    if (!late_issue) issue_next();
    bf16x8 fb0[3][2], fb1[3][2];
    f32x4 r0, r1;
    read_b(s_cur, 0, fb0);
    read_raw(s_nxt, r0, r1);
    if (NBW == 4) read_b(s_cur, 1, fb1);
    mma(fa, fb0, 0);
    if (late_issue) issue_next();
    split_frag(r0, r1, fa_next);
    if (NBW == 4) mma(fa, fb1, 1);
  };
  auto last_step = [&](int k, const bf16x8 (&fa)[3]) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 fb0[3][2], fb1[3][2];
    read_b(NST == 3 ? slot : (k & 1), 0, fb0);
    if (NBW == 4) read_b(NST == 3 ? slot : (k & 1), 1, fb1);
    mma(fa, fb0, 0);
    if (NBW == 4) mma(fa, fb1, 1);
  };

  int k = 0;
  for (; k + 2 <= nk - 1; k += 2) {
    step(k, fa0, fa1);
    step(k + 1, fa1, fa0);
  }
  if (k < nk - 1) {
    step(k, fa0, fa1);
    last_step(k + 1, fa1);
  } else {
    last_step(k, fa0);
  }

  if (ABL & 8) {
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[b][r]));
    return;
  }
  // ---- epilogue through an LDS transpose, two halves of 64 rows (waves 0,1 then 2,3); every thread then
  //      handles four consecutive channels of a row: one 16-byte residual load, one 16-byte store
  float* scratch = reinterpret_cast<float*>(lds);
  constexpr int LD = BN + 4, TPR = BN / 4, RPP = kThreads / TPR;
  const int c4 = (tid % TPR) * 4, r0e = tid / TPR;
  const int j = n0 + c4;
  const bool jok = j < p.Cout;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && !p.partial && jok) bias = *reinterpret_cast<const f32x4*>(p.bias + j);
  float* dst = p.partial ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : p.y;
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                           // ring (h = 0) / previous half (h = 1) no longer read
    if ((wave >> 1) == h) {
#pragma unroll
      for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          scratch[i * LD + b * 32 + (lane & 31)] = acc[b][r];
        }
    }
    __syncthreads();
    if (!jok) continue;
#pragma unroll
    for (int ps = 0; ps < 64 / RPP; ++ps) {
      const int i = r0e + ps * RPP;
      const int m = m0 + h * 64 + i;
      if (m >= p.M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + i * LD + c4);
      if (ABL & 4) {
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        continue;
      }
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

int g_wide_mode = -1;        // -1: read BGS_BFX_WIDE once; 0 off; 1 auto; 2 every eligible layer
int g_wide_nst = 0;          // 0 auto | 2 | 3
int g_wide_nbw = 0;          // 0 auto | 2 (128 x 64 tile) | 4 (128 x 128)
int g_wide_narrow = 1;       // automatic mode: 0 = never pick the 128 x 64 tile (BGS_BFX_WIDE_NARROW, A/B)
int g_wide_splitk = -1;      // -1 auto | 1..16
int g_wide_last = 0;         // bit 0: the wide kernel ran; bits 4..7: NST; bits 8..: K slices
int g_wide_ablate = 0;       // -DBGS_ABLATE builds only
int g_wide_flags = 0;        // WideArgs::flags (A/B: bits 16.. of the tuning hook's mode)

}  // namespace

// Called by conv_bfx.hip's launcher for bf16x6 (three planes) 1x1 / pad 0 problems.  Returns -1 when the layer is
// not eligible (the caller takes the 64 x 64 ring), else a BGS status.
void bgs_internal_conv1x1_bfx_wide_clear_last() { g_wide_last = 0; }

// forced_only: take the layer only under mode 2 (called ahead of the filter-resident kernel, which keeps its
// measured layers in the automatic mode)
int bgs_internal_conv1x1_bfx_wide(const bgs_conv::ConvArgs& pc, const void* wsplit, int KC, void* workspace,
                                  size_t workspace_bytes, int forced_only, hipStream_t st) {
  if (g_wide_mode < 0) {
    const char* e = getenv("BGS_BFX_WIDE");
    g_wide_mode = e ? atoi(e) : 1;
    if (const char* n = getenv("BGS_BFX_WIDE_NST")) g_wide_nst = atoi(n);
    if (const char* s = getenv("BGS_BFX_WIDE_SPLITK")) g_wide_splitk = atoi(s);
    if (const char* s = getenv("BGS_BFX_WIDE_NARROW")) g_wide_narrow = atoi(s);
  }
  g_wide_last = 0;
  if (!g_wide_mode || (forced_only && g_wide_mode != 2)) return -1;
  if (pc.R != 1 || pc.S != 1 || pc.pad != 0 || pc.rowmap) return -1;
  if ((pc.K & 15) || pc.K < 64 || pc.Cout < 64 || pc.M < 128) return -1;
  if (pc.res_mode == 3) return -1;
  const uintptr_t al = (uintptr_t)pc.y | (uintptr_t)pc.res | (uintptr_t)pc.mask | (uintptr_t)pc.bias |
                       (uintptr_t)pc.x | (uintptr_t)wsplit | (uintptr_t)workspace;
  if ((pc.Cout & 3) || (al & 15)) return -1;
  WideArgs q;
  q.c = pc;
  ConvArgs& p = q.c;
  q.ws = reinterpret_cast<const __bf16*>(wsplit);
  q.KC = KC;
  q.flags = g_wide_flags;
  int nbw = g_wide_nbw == 2 ? 2 : 4;
  if (p.Cout < 128) nbw = 2;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.Cout + 32 * nbw - 1) / (32 * nbw);
  long long tiles = (long long)p.tiles_m * p.tiles_n;
  const int nk = p.K >> 4;
  // K slices: one launch wants >= ~2 workgroups per CU; a slice keeps >= 8 K steps
  int want = 1;
  if (g_wide_mode == 1) {
    // auto: the layers where the wide tile was measured ahead of the 64 x 64 ring, per layer
    // (tools/wide_ab.py, profiles/r8a_wide_tile_ab.txt: fpn.lat0 147 -> 124 us, l2.b0.c1 77 -> 66, l2.ds 76 -> 64,
    // l3.b0.c1 / l3.ds / fpn.lat1 74 -> 67) AND inside the step (rocprofv3 of the graph-replayed cfg[1] step,
    // profiles/r8a_wide_tile_ab.txt, in-step note in DESIGN 4.20): the large grids with a reduction of >= 256.  The deep reductions on the
    // stride-16 / 32 maps need K sliced to fill the chip and lose the gain to the slab reduction launch
    // (+11 us each in the step); K <= 128, the 528-tile K = 256 layers and the tiny grids stay on the ring.
    const long long tiles4 = (long long)p.tiles_m * ((p.Cout + 127) / 128);
    const bool big = p.Cout >= 128 && tiles4 >= 500 && (long long)p.K * tiles4 >= 256000 && p.K >= 256;
    // 128 x 64 tiles, three-stage ring (three workgroups per CU): ahead of the 64 x 64 ring by 3 - 8 % where one
    // launch still offers >= 500 of them with a deep reduction, or >= 1000 with K >= 256 (profiles/r8k_wide_narrow_ab.txt:
    // l2.c1 39.7 -> 36.8 us, l4.b0.c1 / l4.ds 71.7 -> 67.4, l4.c3 41.7 -> 39.4, fc_reg 77.6 -> 71.6, l1.c1 39.9 -> 38.7,
    // l3.c3 39.6 -> 38.8); behind it on K <= 128 and on the 264-tile layers.  The two-stage form (five per CU) loses
    // everywhere.
    const long long tiles2 = (long long)p.tiles_m * ((p.Cout + 63) / 64);
    const bool narrow = !big && p.K < 8192 && g_wide_narrow && g_wide_nbw != 4 &&
                        ((tiles2 >= 500 && p.K >= 512) || (tiles2 >= 1000 && p.K >= 256));
    // the first shared FC (K = 12544, 64 tiles x 8 K slices): 170 -> 147 us against the register-staged 128 x 128
    // kernel it ran on, both with their slab reduction (profiles/r8k_wide_narrow_ab.txt)
    const bool deep = p.Cout >= 128 && p.K >= 8192 && tiles4 >= 32 && workspace != nullptr;
    if (!big && !narrow && !deep) return -1;
    if (narrow) {
      nbw = 2;
      p.tiles_n = (p.Cout + 63) / 64;
      tiles = tiles2;
    }
  }
  if (tiles < 384) want = (int)((640 + tiles - 1) / tiles);
  if (want > nk / 8) want = nk / 8;
  if (want > 8) want = 8;
  if (g_wide_splitk >= 1) want = g_wide_splitk;
  if (want > nk / 2) want = nk / 2;
  if (want < 1) want = 1;
  if (!workspace) want = 1;
  while (want > 1 && (size_t)want * (size_t)p.M * p.Cout * sizeof(float) > workspace_bytes) --want;
  int splits = 1;
  p.partial = nullptr;
  p.kt_per_split = 0;
  for (; want > 1; --want) {                                   // every slice (the last one too) holds >= 2 K steps
    const int per = (nk + want - 1) / want;
    const int n = (nk + per - 1) / per;
    if (n > 1 && per >= 2 && nk - (n - 1) * per >= 2) {
      p.kt_per_split = per;
      splits = n;
      p.partial = reinterpret_cast<float*>(workspace);
      break;
    }
  }
  if (nk < 2) return -1;
  p.chunk = (int)((tiles + 7) / 8);
  dim3 grid((unsigned)(8 * p.chunk), 1u, (unsigned)splits);
  // ring depth: three stages (two workgroups per CU) unless the grid offers a third workgroup per CU
  int nst = tiles * splits > 512 ? 2 : 3;
  if (nbw == 2) nst = 3;
  if (g_wide_nst == 2 || g_wide_nst == 3) nst = g_wide_nst;
  g_wide_last = 1 | (nst << 4) | (splits << 8) | (nbw == 2 ? 0x100000 : 0);
  bgs_internal_census_bump(BGS_CENSUS_BFX_WIDE);
#ifdef BGS_ABLATE
  if (g_wide_ablate) {
#define ABL_W(A_) case A_: if (nst == 3) hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<4, 3, A_>), grid, dim3(kThreads), 0, st, q); \
                           else hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<4, 2, A_>), grid, dim3(kThreads), 0, st, q); break;
    switch (g_wide_ablate) { ABL_W(1) ABL_W(2) ABL_W(3) ABL_W(4) ABL_W(8) ABL_W(9) ABL_W(10) ABL_W(11) ABL_W(16) ABL_W(5) default: return BGS_ERR_UNSUPPORTED; }
#undef ABL_W
  } else
#endif
  if (nbw == 2) {
    if (nst == 3) hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<2, 3>), grid, dim3(kThreads), 0, st, q);
    else hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<2, 2>), grid, dim3(kThreads), 0, st, q);
  } else if (nst == 3) hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<4, 3>), grid, dim3(kThreads), 0, st, q);
  else hipLaunchKernelGGL((conv1x1_bfx_wide_kernel<4, 2>), grid, dim3(kThreads), 0, st, q);
  if (splits > 1) {
    if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
    return bgs_internal_conv_splitk_epilogue(p, splits, st);
  }
  BGS_RETURN_LAUNCH_STATUS();
}

// workspace the wide kernel wants for a layer (0: none, or not eligible)
size_t bgs_internal_conv1x1_bfx_wide_workspace(long long M, int Cout, int K) {
  if (M < 128 || Cout < 64 || (K & 15) || K < 64) return 0;
  const int bn = (g_wide_nbw == 2 || Cout < 128) ? 64 : 128;
  const long long tiles = ((M + 127) / 128) * ((Cout + bn - 1) / bn);
  const int nk = K >> 4;
  int want = 1;
  if (tiles < 384) want = (int)((640 + tiles - 1) / tiles);
  if (want > nk / 8) want = nk / 8;
  if (want > 8) want = 8;
  if (g_wide_splitk >= 1) want = g_wide_splitk > 16 ? 16 : g_wide_splitk;
  if (want < 1) want = 1;
  return want > 1 ? (size_t)want * (size_t)M * Cout * sizeof(float) : 0;
}

// tuning / test hook: mode 0 off | 1 auto | 2 every eligible layer; nst 0 auto | 2 | 3; splitk -1 auto | 1..16
extern "C" void bgs_conv_bfx_wide_tuning(int mode, int nst, int splitk) {
  g_wide_ablate = (mode >> 8) & 0xff;          // timing-only ablation modes (-DBGS_ABLATE builds)
  g_wide_flags = (mode >> 16) & 0xff;          // WideArgs::flags
  mode &= 0xff;
  g_wide_mode = mode;
  g_wide_nst = nst & 15;                      // bits 4..7 of `nst`: tile width, 0 auto | 2: 128 x 64 | 4: 128 x 128
  g_wide_nbw = (nst >> 4) & 15;
  g_wide_splitk = splitk;
}

// bit 0: the last bf16x6 1x1 launch took the wide kernel; bits 4..7: ring stages; bits 8..: K slices
extern "C" int bgs_conv_bfx_wide_last_launch(void) { return g_wide_last; }
