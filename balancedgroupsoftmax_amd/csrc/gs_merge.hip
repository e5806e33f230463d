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

int g_merge_pf = -1;      // BGS_GS_MERGE_PF=0: the round-1 form (A/B)

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
  const int grid = N < kMaxGrid ? N : kMaxGrid;
  if (W % 4 == 0 && (uintptr_t)logits % 16 == 0)
    launch_merge<4>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  else if (W % 2 == 0 && (uintptr_t)logits % 8 == 0)
    launch_merge<2>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  else
    launch_merge<1>(grid, st, logits, geom, cls2col, N, C, B, W, scores_out);
  BGS_RETURN_LAUNCH_STATUS();
}
