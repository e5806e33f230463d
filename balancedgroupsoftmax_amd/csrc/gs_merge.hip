// Inference score merge of Balanced Group Softmax for gfx950.
//
// Replaces GSBBoxHeadWith0._merge_score (mmdet/models/bbox_heads/gs_bbox_head_with0.py:239-273):
// the reference runs B softmax kernels, 2 zero-fills of [N, C], B-1 index_puts, a broadcast
// multiply and 2 strided copies.  Here one pass: the row is read once (registers), per-bin
// softmax as in the loss kernel (gs_rowblock.h), the normalised row goes through LDS once and
// the [N, C] score row is written once, coalesced:
//     scores[r, 0] = p_0[r, 0]
//     scores[r, c] = p_0[r, 1] * p_b[r, k]     with column cls2col[c] = s_b + k,  k >= 1
// ("others" column 0 of every foreground bin is dropped; rows do not sum to 1.)
#include "bgs_common.h"
#include "gs_rowblock.h"

namespace {

constexpr int kMaxGrid = 2048;

template <int VEC, int KPT>
__global__ __launch_bounds__(1024) void gs_merge_rowblock_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ pslice,
    const int32_t* __restrict__ cls2col, int N, int C, int B, int W, int nchunks,
    float* __restrict__ scores) {
  __shared__ bgs::RowShared sh;
  extern __shared__ __attribute__((aligned(16))) float prow[];  // [W] normalised row
  bgs::RowLanes<VEC, KPT> L;
  bgs::init_row_lanes<VEC, KPT>(L, sh, pslice, B, W, nchunks);
  const int tid = threadIdx.x;
  const int nw = blockDim.x >> 6;
  int s0, n0;
  bgs::bin_range(pslice, 0, W, s0, n0);
  const int bg_col = s0;                         // p_0[:, 0]
  const int fg_col = n0 > 1 ? s0 + 1 : -1;       // p_0[:, 1]

  for (int r = blockIdx.x; r < N; r += gridDim.x) {
    const float* zr = logits + (size_t)r * W;
    float v[KPT][VEC], e[KPT][VEC];
    bgs::load_row<VEC, KPT>(L, zr, v);
    bgs::bin_max_pass<VEC, KPT>(L, sh, B, v);
    __syncthreads();
    bgs::bin_exp_sum_pass<VEC, KPT>(L, sh, B, nw, v, e);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      if (!L.valid[q]) continue;
      const int b0 = L.binid[q][0];
      const float invS0 = b0 >= 0 ? 1.f / bgs::lookup_sum(sh, b0, nw) : 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int b = L.binid[q][j];
        float invS = invS0;
        if (b != b0 && b >= 0) invS = 1.f / bgs::lookup_sum(sh, b, nw);
        prow[L.col0[q] + j] = b >= 0 ? e[q][j] * invS : 0.f;
      }
    }
    __syncthreads();
    const float pfg = fg_col >= 0 ? prow[fg_col] : 0.f;
    float* out = scores + (size_t)r * C;
    for (int c = tid; c < C; c += blockDim.x) {
      const int col = cls2col[c];
      float sc = 0.f;
      if (col >= 0 && col < W) sc = (col == bg_col) ? prow[col] : pfg * prow[col];
      out[c] = sc;
    }
    // prow / red_sum are rewritten only after the next row's first two barriers... prow is
    // written after barrier #2 of the next row, red_sum after barrier #1: both safe.  red_max
    // is written before barrier #1 of the next row but nobody reads it after barrier #2.
  }
}

template <int VEC, int KPT>
void launch_merge(int grid, int block, hipStream_t st, const float* logits, const int64_t* ps,
                  const int32_t* c2c, int N, int C, int B, int W, int nchunks, float* scores) {
  hipLaunchKernelGGL((gs_merge_rowblock_kernel<VEC, KPT>), dim3(grid), dim3(block),
                     sizeof(float) * (size_t)W, st, logits, ps, c2c, N, C, B, W, nchunks, scores);
}

}  // namespace

extern "C" int bgs_gs_merge_score(const float* logits, const int64_t* pred_slice,
                                  const int32_t* cls2col, int N, int C, int B, int W,
                                  float* scores_out, bgs_stream_t stream) {
  if (N < 0 || C <= 0 || B <= 0 || W <= 0) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS) return BGS_ERR_UNSUPPORTED;
  if (N == 0) return BGS_OK;
  if (!logits || !pred_slice || !cls2col || !scores_out) return BGS_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  int vec = 1;
  if (W % 4 == 0 && (uintptr_t)logits % 16 == 0) vec = 4;
  else if (W % 2 == 0 && (uintptr_t)logits % 8 == 0) vec = 2;
  const int nchunks = W / vec;
  int kpt = 1;
  int block = ((nchunks + 63) / 64) * 64;
  if (block > 1024) {
    kpt = 2;
    block = (((nchunks + 1) / 2 + 63) / 64) * 64;
  }
  if (block > 1024 || (size_t)W * sizeof(float) > 48 * 1024) return BGS_ERR_UNSUPPORTED;
  const int grid = N < kMaxGrid ? N : kMaxGrid;
#define BGS_MERGE_LAUNCH(V, K) \
  launch_merge<V, K>(grid, block, st, logits, pred_slice, cls2col, N, C, B, W, nchunks, scores_out)
  if (vec == 4 && kpt == 1) BGS_MERGE_LAUNCH(4, 1);
  else if (vec == 4) BGS_MERGE_LAUNCH(4, 2);
  else if (vec == 2 && kpt == 1) BGS_MERGE_LAUNCH(2, 1);
  else if (vec == 2) BGS_MERGE_LAUNCH(2, 2);
  else if (kpt == 1) BGS_MERGE_LAUNCH(1, 1);
  else BGS_MERGE_LAUNCH(1, 2);
#undef BGS_MERGE_LAUNCH
  BGS_RETURN_LAUNCH_STATUS();
}
