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
//   n_fg == 0 -> w = 0;   k >= M -> w = 1;   else  w = fg OR (a keyed pseudo-random permutation of
//   [0, M) maps the row's position among the non-fg rows below k: bgs::gs_perm)  == uniform
//   sampling of exactly k rows without replacement, O(1) per row after a prefix count.
//   avg    = max(sum_r w[r], 1)
// Fixed-shape batches: `row_weights` (the detector's label_weights) marks padding slots with a
// value <= 0 — the reference's sampler returns FEWER RoIs instead (two_stage.py:200-210), so such
// rows are excluded from everything: not counted in N, never sampled, weight 0 in every bin.
#include "bgs_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kNW = kThreads / BGS_WAVE;

struct PrepShared {
  int wsum_i[kNW];
  double wsum_d[kNW];
  int wtot[kNW];   // candidates per wave of the current chunk (block-wide prefix count)
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
  const unsigned salt = bgs::gs_bin_salt(seed, (uint32_t)b);          // key of the bin's permutation

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

  // weights + their sum.  Sampled bins: the candidates (real, non-foreground rows) are numbered in
  // row order by a block-wide prefix count, chunk by chunk; row r is drawn iff the keyed
  // permutation of [0, n_bg) maps its position below k (bgs::gs_perm): exactly k rows.
  const float* cw = (cls_weight && b >= 1) ? cls_weight + (size_t)(b - 1) * cw_stride : nullptr;
  double wsum = 0.0;
  int base = 0;                      // candidates in the chunks before this one (block-uniform)
  for (int r0 = 0; r0 < N; r0 += kThreads) {
    const int r = r0 + tid;
    const bool in = r < N;
    int64_t bl = 0;
    bool real = false;
    if (in) {
      int64_t y = labels[r];
      y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
      bl = map[y];
      real = !row_weights || row_weights[r] > 0.f;
    }
    const bool cand = in && real && !(bl > 0);
    bool sel = true;
    if (mode == 2) {                 // block-uniform
      const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) sh.wtot[wave] = __popcll(m);
      __syncthreads();
      int off = 0, tot = 0;
      for (int v = 0; v < kNW; ++v) {
        const int t = sh.wtot[v];
        off += v < wave ? t : 0;
        tot += t;
      }
      __syncthreads();               // wtot is rewritten by the next chunk
      sel = bl > 0;
      if (cand && k > 0) sel = bgs::gs_perm(salt, (uint32_t)(base + off + before), (uint32_t)n_bg) < (uint32_t)k;
      base += tot;
    }
    if (!in) continue;
    float w;
    if (mode == 0 || !real) {
      w = 0.f;  // (padding slot, or) the reference returns zeros BEFORE the class-weight multiply (reweight.py:65-66)
    } else {
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
