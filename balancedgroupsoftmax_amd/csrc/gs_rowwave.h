// Row-per-WAVE machinery shared by the group-softmax loss and score-merge kernels (gfx950).
//
// The op is 9.9 KB of HBM traffic per row, so to stay under the HBM roofline a row may cost
// at most a few hundred wave-instructions.  The first version spread a row over 5 waves with
// 4 logits per thread held in registers; per-thread fixed costs (reductions, LDS lookups, bin
// bookkeeping, and 16-byte stores that HIP's float4 split into 4 dword stores) dominated:
// ~4000 wave-instructions per row, VALU-issue bound at 1.3-1.7 TB/s (profiles/r1a, r1b).
// Now: a workgroup of 4 waves owns a row; the row is staged once in LDS (16-byte coalesced
// global loads, ds_write_b128), and each BIN is swept by ONE wave with bin-aligned indexing
// (lane j handles columns s_b + j, s_b + j + 64, ...): all per-bin quantities (max, sum,
// 1/avg, target column) are wave-uniform scalars, there are no selection chains and no
// boundary special cases.  The bins of a row go round-robin over the 4 waves (they are
// independent), the gradient overwrites the row in LDS and leaves through ds_read_b128 +
// 16-byte coalesced global stores.  Two barriers per row, LDS rows double-buffered.
#pragma once

#include <math.h>

#include "bgs_common.h"

namespace bgs {

// Bin geometry travels BY VALUE in the kernel-argument segment (the host copy of
// pred_slice_with0.pt): no global-memory round trip before the first row can be touched.
struct BinGeom {
  int start[BGS_MAX_BINS];
  int len[BGS_MAX_BINS];
};

// Host side: validate + convert the [B,2] int64 (start, length) table.
// *tiles = 1 when the bins are non-empty, ascending and cover [0, W) exactly (what
// tools/lvis_analyse.py:39-54 produces); the wave kernels require it.
inline int make_bin_geom(const int64_t* host_pred_slice, int B, int W, BinGeom* out, int* tiles) {
  if (!host_pred_slice || B <= 0 || B > BGS_MAX_BINS) return BGS_ERR_INVALID_ARG;
  for (int b = 0; b < BGS_MAX_BINS; ++b) {
    out->start[b] = W;
    out->len[b] = 0;
  }
  int64_t expect = 0;
  int ok = 1;
  for (int b = 0; b < B; ++b) {
    const int64_t s = host_pred_slice[2 * b], n = host_pred_slice[2 * b + 1];
    if (s < 0 || n < 0 || s + n > (int64_t)W) return BGS_ERR_INVALID_ARG;
    if (s != expect || n < 1) ok = 0;
    expect = s + n;
    out->start[b] = (int)s;
    out->len[b] = (int)n;
  }
  if (expect != W) ok = 0;
  if (tiles) *tiles = ok;
  return BGS_OK;
}

constexpr float kLog2e = 1.4426950408889634f;

// Sweeps are unrolled kSweep-fold: the LDS reads of a step are issued back to back (clamped
// index + select instead of a divergent bound check), so a lone wave is not serialised on
// the ~100-cycle LDS latency of every single element.
constexpr int kSweep = 6;  // 6 * 64 = 384 columns per step >= the largest LVIS bin (371)

// One bin of one row held in LDS: returns the bin max m and the sum S of exp(z - m), and
// replaces z by e = exp(z - m) in place.  All 64 lanes of the wave must call it.
__device__ __forceinline__ void bin_softmax_inplace(float* __restrict__ seg, int n, int lane,
                                                    float& m, float& S) {
  float pm = -INFINITY;
  for (int j0 = lane; j0 < n; j0 += BGS_WAVE * kSweep) {
    float x[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) x[u] = seg[min(j0 + BGS_WAVE * u, n - 1)];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) pm = fmaxf(pm, x[u]);  // duplicates of seg[n-1] are harmless
  }
  m = wave_max(pm);
  float ps = 0.f;
  for (int j0 = lane; j0 < n; j0 += BGS_WAVE * kSweep) {
    float x[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) x[u] = seg[min(j0 + BGS_WAVE * u, n - 1)];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      const int j = j0 + BGS_WAVE * u;
      const float e = __builtin_amdgcn_exp2f((x[u] - m) * kLog2e);  // v_exp_f32
      if (j < n) {
        seg[j] = e;
        ps += e;
      }
    }
  }
  S = wave_sum(ps);
}

// Fused per-bin loss + gradient for bins of at most 64*kSweep columns (every shipped LVIS
// table): the bin is read from LDS ONCE into registers, max / exp / sum / gradient are
// computed there and the gradient is written back ONCE.  Reads past the end of the bin stay
// inside the row buffer (the caller pads it by 64*kSweep floats) and are masked out.
// Returns the row's loss term  coef * (logsumexp - z[tgt]).
template <bool WRITE_GRAD>
__device__ __forceinline__ float bin_loss_registers(float* __restrict__ seg, int n, int lane,
                                                    float coef, int tgt) {
  float x[kSweep];
  bool ok[kSweep];
#pragma unroll
  for (int u = 0; u < kSweep; ++u) {
    const int j = lane + BGS_WAVE * u;
    ok[u] = j < n;
    x[u] = seg[j];
  }
  const float zt = seg[tgt];
  float pm = -INFINITY;
#pragma unroll
  for (int u = 0; u < kSweep; ++u) pm = fmaxf(pm, ok[u] ? x[u] : -INFINITY);
  const float m = wave_max(pm);
  float ps = 0.f;
#pragma unroll
  for (int u = 0; u < kSweep; ++u) {
    x[u] = ok[u] ? __builtin_amdgcn_exp2f((x[u] - m) * kLog2e) : 0.f;
    ps += x[u];
  }
  const float S = wave_sum(ps);
  if (WRITE_GRAD) {
    const float k = coef / S;
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      const int j = lane + BGS_WAVE * u;
      if (ok[u]) seg[j] = x[u] * k - (j == tgt ? coef : 0.f);
    }
  }
  return coef * ((m + logf(S)) - zt);
}

// seg[j] = seg[j] * k - (j == tgt ? c : 0)   (gradient of one bin, in place)
__device__ __forceinline__ void bin_grad_inplace(float* __restrict__ seg, int n, int lane, float k,
                                                 float c, int tgt) {
  for (int j0 = lane; j0 < n; j0 += BGS_WAVE * kSweep) {
    float x[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) x[u] = seg[min(j0 + BGS_WAVE * u, n - 1)];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      const int j = j0 + BGS_WAVE * u;
      if (j < n) seg[j] = x[u] * k - (j == tgt ? c : 0.f);
    }
  }
}

// seg[j] *= k
__device__ __forceinline__ void bin_scale_inplace(float* __restrict__ seg, int n, int lane,
                                                  float k) {
  for (int j0 = lane; j0 < n; j0 += BGS_WAVE * kSweep) {
    float x[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) x[u] = seg[min(j0 + BGS_WAVE * u, n - 1)];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      const int j = j0 + BGS_WAVE * u;
      if (j < n) seg[j] = x[u] * k;
    }
  }
}

// ---- row-per-wave-PRIVATE form (round 6) --------------------------------------------------------
// A wave owns a row end to end: its LDS row is touched by no other wave, so the only ordering the
// phases need is the wave's own program order (the LDS serves one wave's instructions in issue
// order).  wave_phase() keeps the COMPILER from moving a lane's LDS accesses across the point where
// other lanes' data is consumed; it emits no instruction that waits on another wave.
__device__ __forceinline__ void wave_phase() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Floats a wave reads past the end of ITS row in bin_loss_registers / bin_probs_registers (the sweep
// always issues kSweep reads of 64 lanes from a bin's start): the next wave's row absorbs them inside
// a workgroup, the last wave's row needs this much slack behind it.
inline int row_read_slack(const BinGeom& g, int B, int W) {
  int over = 0;
  for (int b = 0; b < B; ++b) {
    const int end = g.start[b] + BGS_WAVE * kSweep;
    if (g.len[b] > 0 && end - W > over) over = end - W;
  }
  return (over + 3) & ~3;
}

// One bin (at most 64*kSweep columns) of a row in LDS -> its softmax, in place: p = exp(z - m) * (1 / S) with the
// operations and their order of bin_softmax_inplace + bin_scale_inplace (bit-identical probabilities), but the
// bin is read ONCE and written ONCE.
__device__ __forceinline__ void bin_probs_registers(float* __restrict__ seg, int n, int lane) {
  float x[kSweep];
  bool ok[kSweep];
#pragma unroll
  for (int u = 0; u < kSweep; ++u) {
    const int j = lane + BGS_WAVE * u;
    ok[u] = j < n;
    x[u] = seg[j];
  }
  float pm = -INFINITY;
#pragma unroll
  for (int u = 0; u < kSweep; ++u) pm = fmaxf(pm, ok[u] ? x[u] : -INFINITY);
  const float m = wave_max(pm);
  float ps = 0.f;
#pragma unroll
  for (int u = 0; u < kSweep; ++u) {
    x[u] = __builtin_amdgcn_exp2f((x[u] - m) * kLog2e);
    if (ok[u]) ps += x[u];
  }
  const float k = 1.f / wave_sum(ps);
#pragma unroll
  for (int u = 0; u < kSweep; ++u) {
    if (ok[u]) seg[lane + BGS_WAVE * u] = x[u] * k;
  }
}

// global row -> LDS row, cooperatively by `nthreads` threads (VEC floats per thread per step)
template <int VEC>
__device__ __forceinline__ void stage_row(const float* __restrict__ g, float* __restrict__ row,
                                          int W, int tid, int nthreads) {
  for (int c = tid * VEC; c < W; c += nthreads * VEC) {
    float t[VEC];
    load_vec<VEC>(g + c, t);
    store_vec<VEC>(row + c, t);
  }
}

template <int VEC>
__device__ __forceinline__ void unstage_row(const float* __restrict__ row, float* __restrict__ g,
                                            int W, int tid, int nthreads) {
  for (int c = tid * VEC; c < W; c += nthreads * VEC) {
    float t[VEC];
    load_vec<VEC>(row + c, t);
    store_vec<VEC>(g + c, t);
  }
}

}  // namespace bgs
