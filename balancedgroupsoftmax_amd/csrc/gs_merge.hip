// Inference score merge of Balanced Group Softmax for gfx950.
//
// Replaces GSBBoxHeadWith0._merge_score (mmdet/models/bbox_heads/gs_bbox_head_with0.py:239-273):
// the reference runs B softmax kernels, 2 zero-fills of [N, C], B-1 index_puts, a broadcast
// multiply and 2 strided copies.  Here one pass, one 4-wave workgroup per RoI row
// (gs_rowwave.h): the row is read once into LDS, every bin is normalised in place by one wave,
// and the [N, C] score row is gathered from LDS and written once, coalesced:
//     scores[r, 0] = p_0[r, 0]
//     scores[r, c] = p_0[r, 1] * p_b[r, k]     with column cls2col[c] = s_b + k,  k >= 1
// ("others" column 0 of every foreground bin is dropped; rows do not sum to 1.)
// Algorithmic bytes per RoI: W*4 read + C*4 written = 9,868 B (cls2col stays in L2).
#include <stdlib.h>

#include "bgs_common.h"
#include "gs_rowwave.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / BGS_WAVE;
constexpr int kMaxGrid = 2048;

// Round 5 (HBM-bound, 9,868 B / RoI): at R = 65,536 the first version ran 2.93 TB/s = 0.37 of 8 TB/s
// (profiles/r9a) — every row paid a global round trip in front of its first barrier, re-read the class -> column table
// (4.9 KB from L2) and left through ordinary stores.  Now, as in gs_loss_rowwave_kernel: the NEXT row of the workgroup
// is fetched into registers while the current one is normalised, the table entries of the (at most kColsPerThread)
// classes a thread always writes live in registers for the whole launch, and the scores leave with the non-temporal
// hint.  Same arithmetic in the same order: bit-identical scores.
constexpr int kColsPerThread = 8;            // C <= 2048 on the register path (every LVIS table: C = 1231)

template <int VEC, bool PF>
__global__ __launch_bounds__(kBlock) void gs_merge_rowwave_kernel(
    const float* __restrict__ logits, bgs::BinGeom geom, const int32_t* __restrict__ cls2col,
    int N, int C, int B, int W, int wpad, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][wpad]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  const int bg_col = geom.start[0];                              // p_0[:, 0]
  const int fg_col = geom.len[0] > 1 ? geom.start[0] + 1 : -1;   // p_0[:, 1]
  int cols[kColsPerThread];
  if (PF) {
#pragma unroll
    for (int i = 0; i < kColsPerThread; ++i) {
      const int c = tid + kBlock * i;
      const int col = c < C ? cls2col[c] : -1;
      cols[i] = (col >= 0 && col < W) ? col : -1;
    }
  }
  const int c0 = tid * VEC, c1 = (tid + kBlock) * VEC;
  float pf0[VEC], pf1[VEC];
  auto prefetch = [&](int r) {
    const float* g = logits + (size_t)r * W;
    if (c0 < W) bgs::load_vec<VEC>(g + c0, pf0);
    if (c1 < W) bgs::load_vec<VEC>(g + c1, pf1);
  };
  if (PF && (int)blockIdx.x < N) prefetch(blockIdx.x);
  int par = 0;
  for (int r = blockIdx.x; r < N; r += gridDim.x, par ^= 1) {
    float* row = smem + (size_t)par * wpad;
    if (PF) {
      if (c0 < W) bgs::store_vec<VEC>(row + c0, pf0);
      if (c1 < W) bgs::store_vec<VEC>(row + c1, pf1);
      if (r + (int)gridDim.x < N) prefetch(r + gridDim.x);      // in flight under this row's softmaxes
    } else {
      bgs::stage_row<VEC>(logits + (size_t)r * W, row, W, tid, kBlock);
    }
    __syncthreads();
    for (int b = wave; b < B; b += kWaves) {
      const int n = geom.len[b];
      float* seg = row + geom.start[b];
      float m, S;
      bgs::bin_softmax_inplace(seg, n, lane, m, S);
      bgs::bin_scale_inplace(seg, n, lane, 1.f / S);
    }
    __syncthreads();
    const float pfg = fg_col >= 0 ? row[fg_col] : 0.f;
    float* out = scores + (size_t)r * C;
    if (PF) {
#pragma unroll
      for (int i = 0; i < kColsPerThread; ++i) {
        const int c = tid + kBlock * i;
        if (c >= C) break;
        const int col = cols[i];
        float sc = 0.f;
        if (col >= 0) sc = (col == bg_col) ? row[col] : pfg * row[col];
        __builtin_nontemporal_store(sc, out + c);
      }
    } else {
      for (int c = tid; c < C; c += kBlock) {
        const int col = cls2col[c];
        float sc = 0.f;
        if (col >= 0 && col < W) sc = (col == bg_col) ? row[col] : pfg * row[col];
        out[c] = sc;
      }
    }
  }
}

// Round 6: a row per WAVE with a wave-private LDS row (gs_rowwave.h) — no workgroup barrier in the row loop, 32
// independent rows per CU in LDS and 32 more on their way in registers — and the scores leave in 16-byte stores:
// the score row of RoI r starts at byte 4 r C of the output, 16-byte aligned only every fourth row (C = 1231), so
// a wave writes the ALIGNED 16-byte pieces of the flat output that its row covers (piece i of the row = flat floats
// [4 i - a, 4 i - a + 4) of the row, a = (base / 4 + r C) mod 4) and finishes the at most 3 + 3 floats of the first
// and last piece with dword stores.  A wave's rows are r0 + 4 k grid, so a is the SAME for every row of the wave
// and the 4 KV columns (class -> column table) its lanes gather live in registers for the whole launch.  The
// probabilities come from bin_probs_registers (each bin read once, written once; the operations and their order of
// the kernel above: bit-identical scores).
template <int KV, bool NTL = false>
__global__ __launch_bounds__(kBlock, KV <= 5 ? 8 : 4) void gs_merge_wavepriv_kernel(
    const float* __restrict__ logits, bgs::BinGeom geom, const int32_t* __restrict__ cls2col,
    int N, int C, int B, int W, int zero_slot, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [kWaves][W] + read slack + one 0.f (zero_slot)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  float* row = smem + (size_t)wave * W;
  const int nw = gridDim.x * kWaves;                        // a multiple of 4
  const int nq = W >> 2;
  const int bg_col = geom.start[0];
  const int fg_col = geom.len[0] > 1 ? geom.start[0] + 1 : -1;
  int r = blockIdx.x * kWaves + wave;
  if (tid == 0) smem[zero_slot] = 0.f;                      // what a class without a column reads (score 0)
  const int zcol = zero_slot - wave * W;                    // ... as an index from this wave's row
  // flat float offset of the row's first score from a 16-byte boundary: the same for r, r + nw, r + 2 nw, ...
  const int a = (int)((((uintptr_t)scores >> 2) + (size_t)r * (size_t)C) & 3);
  // piece i = lane + 64 k holds classes 4 i - a + {0..3}; pieces [i_lo, i_hi) lie inside the row and leave in one
  // 16-byte store, the (at most two) pieces cut by the row's ends are finished by lanes 0..7 with dword stores
  const int i_lo = a ? 1 : 0, i_hi = (C + a) >> 2;
  uint32_t cols[KV][2];                                     // two 16-bit LDS indices per register (64 VGPRs: 8 waves / SIMD)
#pragma unroll
  for (int k = 0; k < KV; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * (lane + BGS_WAVE * k) - a + j;
      int col = zcol;
      if (c >= 0 && c < C) {
        const int t = cls2col[c];
        if (t >= 0 && t < W) col = t;
      }
      if (j & 1) cols[k][j >> 1] |= (uint32_t)col << 16;
      else cols[k][j >> 1] = (uint32_t)col;
    }
  }
  // edge elements: lanes 0..3 the classes of the cut first piece, lanes 4..7 those of the cut last piece
  int edge_c = -1, edge_col = zcol;
  if (lane < 8) {
    const int c = lane < 4 ? (a ? lane : -1) : 4 * i_hi - a + (lane - 4);
    if (c >= 0 && c < C && (lane >= 4 || c < 4 - a)) {
      edge_c = c;
      const int t = cls2col[c];
      if (t >= 0 && t < W) edge_col = t;
    }
  }
  const int edge = edge_c >= 0 ? (edge_c << 16) | edge_col : -1;      // one register in the row loop
  const bool bg_in_piece0 = lane == 0 && a == 0;            // class 0 = the background column, not scaled by p_fg
  float pf[KV][4];
  auto prefetch = [&](int rr) {
    const float* g = logits + (size_t)rr * W;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int q = lane + BGS_WAVE * k;
      if (q < nq) {
        if (NTL) bgs::load_vec_nt<4>(g + 4 * q, pf[k]);
        else bgs::load_vec<4>(g + 4 * q, pf[k]);
      }
    }
  };
  if (r < N) prefetch(r);
  __syncthreads();                                          // the zero slot (once per launch)
  for (; r < N; r += nw) {
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int q = lane + BGS_WAVE * k;
      if (q < nq) bgs::store_vec<4>(row + 4 * q, pf[k]);
    }
    if (r + nw < N) prefetch(r + nw);
    bgs::wave_phase();
    for (int b = 0; b < B; ++b) bgs::bin_probs_registers(row + geom.start[b], geom.len[b], lane);
    bgs::wave_phase();
    const float pfg = fg_col >= 0 ? row[fg_col] : 0.f;
    const float pbg = row[bg_col];
    float* out = scores + (size_t)r * C - a;                // 16-byte aligned
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int i = lane + BGS_WAVE * k;
      float sc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t cw = cols[k][j >> 1];
        asm volatile("" : "+v"(cw));                        // keep the unpacking in the loop (hoisted, it costs 10 VGPRs)
        sc[j] = pfg * row[(j & 1) ? cw >> 16 : cw & 0xffffu];
      }
      if (k == 0 && bg_in_piece0) sc[0] = pbg;
      if (i >= i_lo && i < i_hi) bgs::store_vec_nt<4>(out + 4 * i, sc);
    }
    if (edge >= 0) {
      const int ec = edge >> 16;
      const float p = row[edge & 0xffff];
      __builtin_nontemporal_store(ec == 0 ? pbg : pfg * p, out + a + ec);
    }
    bgs::wave_phase();
  }
}

int g_merge_pf = -1;      // BGS_GS_MERGE_PF=0: the round-1 form (A/B)
int g_merge_mode = -1;    // bgs_gs_merge_tuning / BGS_GS_MERGE_WAVEPRIV: 1 (default) = 2 = row-per-wave kernel from
                          // kMergeWavePrivMinRows rows | 3 = .. with non-temporal row loads | 0 = the 4-wave-per-row kernel always
constexpr int kMergeWavePrivMinRows = 4096;
int g_merge_min_rows = kMergeWavePrivMinRows;

template <int VEC>
void launch_merge(int grid, hipStream_t st, const float* logits, const bgs::BinGeom& geom,
                  const int32_t* c2c, int N, int C, int B, int W, float* scores) {
  const int wpad = (W + 3) & ~3;
  if (g_merge_pf < 0) {
    const char* e = getenv("BGS_GS_MERGE_PF");
    g_merge_pf = (e && atoi(e) == 0) ? 0 : 1;
  }
  const bool pf = g_merge_pf && W <= 2 * kBlock * VEC && C <= kColsPerThread * kBlock;
  if (pf)
    hipLaunchKernelGGL((gs_merge_rowwave_kernel<VEC, true>), dim3(grid), dim3(kBlock),
                       sizeof(float) * 2 * (size_t)wpad, st, logits, geom, c2c, N, C, B, W, wpad, scores);
  else
    hipLaunchKernelGGL((gs_merge_rowwave_kernel<VEC, false>), dim3(grid), dim3(kBlock),
                       sizeof(float) * 2 * (size_t)wpad, st, logits, geom, c2c, N, C, B, W, wpad, scores);
}

}  // namespace

// tuning / test hook: mode 1 (default) = row-per-wave kernel for N >= min_rows (< 0: default 4096) | 0 = never
extern "C" void bgs_gs_merge_tuning(int mode, int min_rows) {
  g_merge_mode = mode < 0 || mode > 3 ? 1 : mode;
  g_merge_min_rows = min_rows < 0 ? kMergeWavePrivMinRows : min_rows;
}

extern "C" int bgs_gs_merge_score(const float* logits, const int64_t* host_pred_slice,
                                  const int32_t* cls2col, int N, int C, int B, int W,
                                  float* scores_out, bgs_stream_t stream) {
  if (N < 0 || C <= 0 || B <= 0 || W <= 0) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS) return BGS_ERR_UNSUPPORTED;
  if (N == 0) return BGS_OK;
  if (!logits || !host_pred_slice || !cls2col || !scores_out) return BGS_ERR_INVALID_ARG;
  bgs::BinGeom geom;
  int tiles = 0;
  const int grc = bgs::make_bin_geom(host_pred_slice, B, W, &geom, &tiles);
  if (grc != BGS_OK) return grc;
  // the score merge is only defined for tables whose bins tile the logits (bin 0 = {bg, fg})
  if (!tiles || W > 8000) return BGS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (g_merge_mode < 0) {
    const char* e = getenv("BGS_GS_MERGE_WAVEPRIV");
    g_merge_mode = (e && atoi(e) == 0) ? 0 : 1;
  }
  bool bins_fit = true;
  for (int b = 0; b < B; ++b) bins_fit = bins_fit && geom.len[b] <= BGS_WAVE * bgs::kSweep;
  if (g_merge_mode >= 1 && N >= g_merge_min_rows && bins_fit && W % 4 == 0 && W <= 2048 && C <= W &&
      (uintptr_t)logits % 16 == 0 && (uintptr_t)scores_out % 4 == 0) {
    const int zero_slot = kWaves * W + bgs::row_read_slack(geom, B, W);
    const size_t lds = sizeof(float) * ((size_t)zero_slot + 4);
    int per_cu = (int)((160u * 1024u) / lds);
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    const int want = (N + kWaves - 1) / kWaves;
    const int grid = want < 256 * per_cu ? want : 256 * per_cu;
    const int kv = ((C + 3) / 4 + 1 + BGS_WAVE - 1) / BGS_WAVE;      // pieces of the score row incl. the shifted tail
    const int kvw = (W / 4 + BGS_WAVE - 1) / BGS_WAVE;
    const int kvm = kv > kvw ? kv : kvw;
    // non-temporal row loads: mode 3, A/B only (slower or equal at every size, profiles/r10c_gs_stream_ab.txt)
    const bool ntl = g_merge_mode == 3;
#define GS_MW(KV_)                                                                                                    \
  do {                                                                                                                \
    if (ntl)                                                                                                          \
      hipLaunchKernelGGL((gs_merge_wavepriv_kernel<KV_, true>), dim3(grid), dim3(kBlock), lds, st, logits, geom,      \
                         cls2col, N, C, B, W, zero_slot, scores_out);                                                 \
    else                                                                                                              \
      hipLaunchKernelGGL((gs_merge_wavepriv_kernel<KV_, false>), dim3(grid), dim3(kBlock), lds, st, logits, geom,     \
                         cls2col, N, C, B, W, zero_slot, scores_out);                                                 \
  } while (0)
    if (kvm <= 2) GS_MW(2);
    else if (kvm <= 5) GS_MW(5);
    else GS_MW(8);
#undef GS_MW
    BGS_RETURN_LAUNCH_STATUS();
  }
  const int grid = N < kMaxGrid ? N : kMaxGrid;
  if (W % 4 == 0 && (uintptr_t)logits % 16 == 0)
    launch_merge<4>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  else if (W % 2 == 0 && (uintptr_t)logits % 8 == 0)
    launch_merge<2>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  else
    launch_merge<1>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  BGS_RETURN_LAUNCH_STATUS();
}
