// Device-side label remap + "others" sampling for Balanced Group Softmax (gfx950).
//
// Replaces GSBBoxHeadWith0._remap_labels (mmdet/models/bbox_heads/gs_bbox_head_with0.py:91-112)
// and _sample_others (:63-89; Reweight variant gs_bbox_head_with0_reweight.py:57-87).
//
// The reference does, per bin, 2x nonzero() (device sync), a D2H copy of the index list,
// np.random.choice(replace=False) on the host, an H2D copy and an index_put, then a
// .sum().item() sync for the avg_factor — >= 17 host round trips per loss() call.
// Here: ONE launch, one workgroup per bin, no host involvement:
//   bl[r]  = label2binlabel[b, labels[r]]                       (int64 gather, bit-exact)
//   n_fg   = #{bl > 0};  k = int(n_fg * ratio);  M = N - n_fg
//   n_fg == 0 -> w = 0;   k >= M -> w = 1;   else  w = fg OR (row is among the k smallest
//   32-bit counter-based random keys (bgs::gs_key) of the non-fg rows)  == uniform sampling of exactly k
//   rows without replacement (ties broken by row index), found by a 4-pass radix select.
//   avg    = max(sum_r w[r], 1)
// Fixed-shape batches: `row_weights` (the detector's label_weights) marks padding slots with a
// value <= 0 — the reference's sampler returns FEWER RoIs instead (two_stage.py:200-210), so such
// rows are excluded from everything: not counted in N, never sampled, weight 0 in every bin.
#include "bgs_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kNW = kThreads / BGS_WAVE;

struct PrepShared {
  int hist[256];
  int wsum_i[kNW];
  double wsum_d[kNW];
  unsigned prefix;  // selected high bits so far
  int rem;          // rank still to be resolved inside the current prefix
  int tie_row_bound;
};

__device__ __forceinline__ int block_sum_i(int v, int* sm) {
  v = bgs::wave_sum_i(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = 0;
  for (int w = 0; w < kNW; ++w) r += sm[w];
  return r;
}

__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  v = bgs::wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < kNW; ++w) r += sm[w];
  return r;
}

__global__ __launch_bounds__(kThreads) void gs_prepare_kernel(
    const int64_t* __restrict__ labels, const int64_t* __restrict__ l2b,
    const float* __restrict__ cls_weight, int cw_stride, const float* __restrict__ row_weights,
    int N, int C, int B, double ratio,
    uint64_t seed, const uint64_t* __restrict__ seed_offset, int32_t* __restrict__ bl_out,
    float* __restrict__ w_out, float* __restrict__ avg_out) {
  __shared__ PrepShared sh;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* map = l2b + (size_t)b * C;
  if (seed_offset) seed += 0x2545F4914F6CDD1Dull * seed_offset[0];  // device-side draw counter
  const unsigned salt = bgs::gs_bin_salt(seed, (uint32_t)b);          // keys: bgs::gs_key(salt, row)

  // pass 0: bin labels + foreground count (+ number of real rows)
  int nfg_local = 0, nreal_local = 0;
  for (int r = tid; r < N; r += kThreads) {
    int64_t y = labels[r];
    y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
    const int64_t bl = map[y];
    if (bl_out) bl_out[(size_t)b * N + r] = (int32_t)bl;
    const bool real = !row_weights || row_weights[r] > 0.f;
    nfg_local += (real && bl > 0) ? 1 : 0;
    nreal_local += real ? 1 : 0;
  }
  const int n_fg = block_sum_i(nfg_local, sh.wsum_i);
  const int n_bg = block_sum_i(nreal_local, sh.wsum_i) - n_fg;

  // mode: 0 = all zero, 1 = all one, 2 = sampled
  int mode;
  int k = 0;
  if (b == 0) {
    mode = 1;  // bin 0 (bg vs fg): weight = ones (gs_bbox_head_with0.py:100-102)
  } else if (n_fg == 0) {
    mode = 0;
  } else {
    k = (int)((double)n_fg * ratio);  // int(fg_num * self.others_sample_ratio)
    mode = (k >= n_bg) ? 1 : 2;
  }

  unsigned T = 0u;       // threshold key
  int tie_bound = N;     // rows with key == T and row < tie_bound are selected
  if (mode == 2 && k > 0) {
    if (tid == 0) {
      sh.prefix = 0u;
      sh.rem = k;  // rank (1-based) of the largest selected key among the non-fg rows
    }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) sh.hist[tid] = 0;
      __syncthreads();
      const unsigned prefix = sh.prefix;
      const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int r = tid; r < N; r += kThreads) {
        int64_t y = labels[r];
        y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
        if (map[y] > 0 || (row_weights && !(row_weights[r] > 0.f))) continue;
        const unsigned key = bgs::gs_key(salt, (uint32_t)r);
        if ((key & himask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 0xFFu], 1);
      }
      __syncthreads();
      if (wave == 0) {
        const int rem = sh.rem;
        const int h0 = sh.hist[4 * lane], h1 = sh.hist[4 * lane + 1], h2 = sh.hist[4 * lane + 2],
                  h3 = sh.hist[4 * lane + 3];
        const int s = h0 + h1 + h2 + h3;
        int incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(incl, off, BGS_WAVE);
          if (lane >= off) incl += t;
        }
        const int excl = incl - s;
        if (excl < rem && rem <= incl) {  // exactly one lane
          int c = excl, d = 4 * lane;
          if (rem > c + h0) { c += h0; ++d;
            if (rem > c + h1) { c += h1; ++d;
              if (rem > c + h2) { c += h2; ++d; } } }
          sh.prefix = prefix | ((unsigned)d << shift);
          sh.rem = rem - c;  // rank inside digit d (>= 1)
          if (pass == 3) sh.hist[0] = sh.hist[d] == rem - c ? 1 : 0;  // all ties selected?
        }
      }
      __syncthreads();
    }
    T = sh.prefix;
    const int need = sh.rem;           // number of rows with key == T to select (>= 1)
    const bool all_ties = sh.hist[0] == 1;
    if (!all_ties) {
      // more rows share the threshold key than we may take: keep the `need` lowest row ids.
      // (probability ~ N / 2^32 per call; a serial scan by one thread is fine.)
      if (tid == 0) {
        int got = 0, bound = N;
        for (int r = 0; r < N; ++r) {
          int64_t y = labels[r];
          y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
          if (map[y] > 0 || (row_weights && !(row_weights[r] > 0.f))) continue;
          if (bgs::gs_key(salt, (uint32_t)r) == T) {
            if (++got == need) { bound = r + 1; break; }
          }
        }
        sh.tie_row_bound = bound;
      }
      __syncthreads();
      tie_bound = sh.tie_row_bound;
    }
  }

  // weights + their sum
  const float* cw = (cls_weight && b >= 1) ? cls_weight + (size_t)(b - 1) * cw_stride : nullptr;
  double wsum = 0.0;
  for (int r = tid; r < N; r += kThreads) {
    int64_t y = labels[r];
    y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
    const int64_t bl = map[y];
    float w;
    if (mode == 0 || (row_weights && !(row_weights[r] > 0.f))) {
      w = 0.f;  // (padding slot, or) the reference returns zeros BEFORE the class-weight multiply (reweight.py:65-66)
    } else {
      bool sel = true;
      if (mode == 2) {
        sel = bl > 0;
        if (!sel && k > 0) {
          const unsigned key = bgs::gs_key(salt, (uint32_t)r);
          sel = (key < T) || (key == T && r < tie_bound);
        }
      }
      w = sel ? 1.f : 0.f;
      if (cw) {
        int64_t idx = bl < 0 ? 0 : (bl >= cw_stride ? (int64_t)cw_stride - 1 : bl);
        w *= cw[idx];
      }
    }
    w_out[(size_t)b * N + r] = w;
    wsum += (double)w;
  }
  const double total = block_sum_d(wsum, sh.wsum_d);
  if (tid == 0) avg_out[b] = fmaxf((float)total, 1.f);
}

}  // namespace

extern "C" int bgs_gs_prepare(const int64_t* labels, const int64_t* label2binlabel,
                              const float* cls_weight, int cls_weight_stride,
                              const float* row_weights, int N, int C, int B,
                              double others_sample_ratio, uint64_t seed,
                              const uint64_t* seed_offset, int32_t* bin_labels_out,
                              float* weights_out, float* avg_out, bgs_stream_t stream) {
  if (N < 0 || C <= 0 || B <= 0) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS) return BGS_ERR_UNSUPPORTED;
  if (!label2binlabel || !avg_out || (N > 0 && (!labels || !weights_out)))
    return BGS_ERR_INVALID_ARG;
  if (cls_weight && cls_weight_stride <= 0) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs_prepare_kernel, dim3(B), dim3(kThreads), 0, (hipStream_t)stream, labels,
                     label2binlabel, cls_weight, cls_weight_stride, row_weights, N, C, B,
                     others_sample_ratio,
                     seed, seed_offset, bin_labels_out, weights_out, avg_out);
  BGS_RETURN_LAUNCH_STATUS();
}
