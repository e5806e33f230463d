// Row-per-workgroup machinery shared by the group-softmax loss and score-merge kernels.
//
// One workgroup owns one RoI row at a time (grid-stride over rows).  Thread t holds the
// VEC consecutive logits of chunk t (+ chunk t+blockDim when KPT == 2) in registers, so a
// wave reads 64*VEC*4 contiguous bytes per load instruction (global_load_dwordx4 for VEC=4).
// For the shipped 5-bin LVIS tables W = 1236 -> 309 float4 chunks -> 320 threads (5 waves).
//
// Bins are contiguous column ranges, so inside a wave a bin is a contiguous lane range and
// only 1-2 of the B bins intersect any one wave.  Which bins those are never changes from
// row to row: the per-element bin id and the per-wave bin mask are computed once per
// workgroup.  Per row: masked wave all-reduce (DPP) of the bins a wave touches -> LDS ->
// one barrier -> every thread combines the <= 16 per-wave partials of the bins it needs.
#pragma once

#include <math.h>

#include "bgs_common.h"

namespace bgs {

constexpr int kMaxWavesPerRow = 16;  // blockDim <= 1024

struct RowShared {
  float red_max[BGS_MAX_BINS][kMaxWavesPerRow];
  float red_sum[BGS_MAX_BINS][kMaxWavesPerRow];
  int tgt[2][BGS_MAX_BINS];     // double-buffered by row parity (read until the row's end)
  float coef[2][BGS_MAX_BINS];
};

template <int VEC, int KPT>
struct RowLanes {
  int binid[KPT][VEC];  // bin of every element this thread owns; -1 = outside every bin / row
  unsigned wave_bins;   // bit b set <=> some lane of this wave owns an element of bin b (uniform)
  int col0[KPT];        // first column of chunk q
  bool valid[KPT];      // chunk q lies inside the row
};

// Bin geometry clamped into [0, W] so that a corrupt pred_slice can never index out of bounds.
__device__ __forceinline__ void bin_range(const int64_t* __restrict__ pslice, int b, int W, int& s,
                                          int& n) {
  s = (int)pslice[2 * b];
  n = (int)pslice[2 * b + 1];
  s = min(max(s, 0), W);
  n = min(max(n, 0), W - s);
}

template <int VEC, int KPT>
__device__ __forceinline__ void init_row_lanes(RowLanes<VEC, KPT>& L, RowShared& sh,
                                               const int64_t* __restrict__ pslice, int B, int W,
                                               int nchunks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int c = tid + q * (int)blockDim.x;
    L.valid[q] = c < nchunks;
    L.col0[q] = c * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) L.binid[q][j] = -1;
  }
  L.wave_bins = 0u;
  for (int b = 0; b < B; ++b) {
    int s, n;
    bin_range(pslice, b, W, s, n);
    bool mine = false;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int col = L.col0[q] + j;
        if (L.valid[q] && col >= s && col < s + n && col < W) {
          L.binid[q][j] = b;
          mine = true;
        }
      }
    }
    if (__ballot(mine) != 0ull) L.wave_bins |= (1u << b);
  }
  // (bin, wave) pairs that never intersect keep the reduction identities forever
  if (lane == 0) {
    for (int b = 0; b < B; ++b) {
      if (!((L.wave_bins >> b) & 1u)) {
        sh.red_max[b][wave] = -INFINITY;
        sh.red_sum[b][wave] = 0.f;
      }
    }
  }
}

template <int VEC, int KPT>
__device__ __forceinline__ void load_row(const RowLanes<VEC, KPT>& L, const float* __restrict__ zr,
                                         float (&v)[KPT][VEC]) {
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    if (L.valid[q]) {
      load_vec<VEC>(zr + L.col0[q], v[q]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[q][j] = 0.f;
    }
  }
}

// Pass 1: per-bin row maximum -> sh.red_max[b][wave].  Caller issues the barrier.
template <int VEC, int KPT>
__device__ __forceinline__ void bin_max_pass(const RowLanes<VEC, KPT>& L, RowShared& sh, int B,
                                             const float (&v)[KPT][VEC]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = 0; b < B; ++b) {
    if ((L.wave_bins >> b) & 1u) {  // wave-uniform
      float pm = -INFINITY;
#pragma unroll
      for (int q = 0; q < KPT; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) pm = fmaxf(pm, L.binid[q][j] == b ? v[q][j] : -INFINITY);
      pm = wave_max(pm);
      if (lane == 0) sh.red_max[b][wave] = pm;
    }
  }
}

__device__ __forceinline__ float lookup_max(const RowShared& sh, int b, int nw) {
  float m = sh.red_max[b][0];
  for (int w = 1; w < nw; ++w) m = fmaxf(m, sh.red_max[b][w]);
  return m;
}

__device__ __forceinline__ float lookup_sum(const RowShared& sh, int b, int nw) {
  float s = sh.red_sum[b][0];
  for (int w = 1; w < nw; ++w) s += sh.red_sum[b][w];
  return s;
}

// After the barrier that follows bin_max_pass: e = exp(v - max_of_own_bin), and the per-bin
// sums of e -> sh.red_sum[b][wave].  Caller issues the barrier.
template <int VEC, int KPT>
__device__ __forceinline__ void bin_exp_sum_pass(const RowLanes<VEC, KPT>& L, RowShared& sh, int B,
                                                 int nw, const float (&v)[KPT][VEC],
                                                 float (&e)[KPT][VEC]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int b0 = L.binid[q][0];
    const float m0 = b0 >= 0 ? lookup_max(sh, b0, nw) : 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int b = L.binid[q][j];
      float m = m0;
      if (b != b0 && b >= 0) m = lookup_max(sh, b, nw);  // chunk straddles a bin boundary
      e[q][j] = b >= 0 ? __expf(v[q][j] - m) : 0.f;
    }
  }
  for (int b = 0; b < B; ++b) {
    if ((L.wave_bins >> b) & 1u) {
      float ps = 0.f;
#pragma unroll
      for (int q = 0; q < KPT; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) ps += (L.binid[q][j] == b) ? e[q][j] : 0.f;
      ps = wave_sum(ps);
      if (lane == 0) sh.red_sum[b][wave] = ps;
    }
  }
}

}  // namespace bgs
