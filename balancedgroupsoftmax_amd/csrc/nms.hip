// Batched greedy NMS entirely on the device, for gfx950 (MI355X).
//
// Replaces ops.nms (mmdet/ops/nms/nms_wrapper.py:8-49) -> nms_cuda
// (mmdet/ops/nms/src/nms_kernel.cu:23-131): the reference computes a 64x64-tiled suppression
// bitmask on the GPU, copies it to the host (512 KB D2H + sync per call), runs the greedy
// scan on the CPU and copies the kept indices back — 10 times per training iteration
// (5 FPN levels x 2 images, mmdet/models/anchor_heads/rpn_head.py:92).
// Here all P problems (image x level) go through ONE pair of launches and nothing leaves
// the device:
//   1. nms_mask_kernel   grid (cb, cb, P): same 64x64 bitmask tiles (upper triangle only),
//      legacy "+1" IoU (nms_kernel.cu:13-21);
//   2. the greedy scan, boxes consumed in chunks of 64, kept indices emitted in ascending
//      (= score) order:
//      nms_scan_wide_kernel (problems of more than 256 boxes: the RPN's 10 x <= 2000, the 1230
//      per-class problems at test time): one 16-wave workgroup per problem.  Wave w owns the "removed" words w, w+16, ... (wave-uniform
//      registers).  Per chunk c the wave owning word c runs the 64-step bit recurrence on the
//      diagonal word with v_readlane (lane r holds row r's word: scalar code, no memory), publishes
//      the 64 "kept" bits through LDS, and after ONE barrier every wave folds its own words:
//      lane r contributes row r's word if r was kept, OR-reduced over the wave with DPP.  The
//      mask words a wave needs (row = lane, word = owned) do not depend on any decision, so they
//      are prefetched four chunks ahead into registers: the serial chain is ~0.4 us per chunk
//      (recurrence + barrier + reduce) instead of 11 us (staging loads, then 2 x 64 dependent LDS
//      reads, all exposed in a single wave).
//      nms_scan_kernel (<= 256 boxes): one wave per problem, rows staged through LDS.
// Boxes must already be sorted by descending score per problem (the callers' topk does that).
// iou_mode 0: suppress when IoU >  thr (nms_kernel.cu:60);  1: IoU >= thr (nms_cpu.cpp:55).
#include <stdlib.h>

#include "bgs_common.h"

namespace {

constexpr int kTile = 64;


// "IoU(a, b) > thr" (mode 0) / ">= thr" (mode 1) with the SAME decision as the division, mostly without it: the
// mask kernel is bound by its 64 fp32 divisions per lane (40 us for the RPN's ten problems).  v = fl(inter / u)
// differs from the real ratio by at most 2^-24 relative, so outside the band |inter - thr u| <= 2^-21 thr u the
// decision is the sign of inter - thr u — which the fp32 product and difference give reliably at that margin;
// inside the band (and for degenerate unions, u <= 0) the division decides, exactly as before.
__device__ __forceinline__ bool iou_exceeds(const float* a, const float* b, float thr, int mode) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = width * height;
  const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  const float u = sa + sb - inter;
  const float m = thr * u;
  const float d = inter - m;
  if (u > 0.f && m > 0.f && fabsf(d) > 4.8e-7f * m) return d > 0.f;
  const float v = inter / u;
  return mode ? (v >= thr) : (v > thr);
}

// boxes [P, nmax, 5] (x1,y1,x2,y2,score), counts [P]; mask [P, nmax, cb] u64.
// Four 64 x 64 tiles per workgroup (one per wave: column tiles 4 blockIdx.x + wave of row tile blockIdx.y), every
// wave staging its own column tile in its own LDS slice and only synchronising with itself (measured equal to the
// one-wave-per-workgroup form: the kernel is bound by the IoU arithmetic, not by its 10,240 dispatches).
constexpr int kMaskWaves = 4;
__global__ __launch_bounds__(kTile * kMaskWaves) void nms_mask_kernel(const float* __restrict__ boxes,
                                                                      const int* __restrict__ counts,
                                                                      int nmax, int cb, float thr,
                                                                      int iou_mode,
                                                                      unsigned long long* __restrict__ mask) {
  const int p = blockIdx.z;
  const int n = min(counts[p], nmax);
  const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x / kTile;
  const int row_start = blockIdx.y, col_start = blockIdx.x * kMaskWaves + wave;
  __shared__ float cbx_all[kMaskWaves][kTile * 4];
  float* cbx = cbx_all[wave];
  if (col_start >= cb || row_start > col_start) return;  // only j > i can be suppressed by i (wave-uniform)
  if (row_start * kTile >= n || col_start * kTile >= n) return;
  const int row_size = min(n - row_start * kTile, kTile);
  const int col_size = min(n - col_start * kTile, kTile);
  const float* pb = boxes + (size_t)p * nmax * 5;
  if (lane < col_size) {
    const float* s = pb + (size_t)(col_start * kTile + lane) * 5;
    cbx[lane * 4 + 0] = s[0];
    cbx[lane * 4 + 1] = s[1];
    cbx[lane * 4 + 2] = s[2];
    cbx[lane * 4 + 3] = s[3];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slice is this wave's own: no workgroup barrier
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < row_size) {
    const int i = row_start * kTile + lane;
    const float* s = pb + (size_t)i * 5;
    const float cur[4] = {s[0], s[1], s[2], s[3]};
    unsigned long long t = 0ull;
    // diagonal tiles carry the full symmetric word (every j != i): bits j > i are "i suppresses
    // j", bits j < i are "i is suppressed by j" (IoU is symmetric) — the scan's parallel
    // fixed-point iteration reads the latter, the serial recurrence ignores them
    const int self = (row_start == col_start) ? lane : -1;
    for (int j = 0; j < col_size; ++j) {
      if (j == self) continue;
      if (iou_exceeds(cur, cbx + j * 4, thr, iou_mode)) t |= 1ull << j;
    }
    mask[((size_t)p * nmax + i) * cb + col_start] = t;
  }
}

// one wave per problem.  keep [P, nmax] int32 (ascending), keep_count [P].
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ counts, int nmax,
                                                      int cb, int max_keep,
                                                      int* __restrict__ keep,
                                                      int* __restrict__ keep_count) {
  const int p = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = min(counts[p], nmax);
  const unsigned long long* pm = mask + (size_t)p * nmax * cb;
  int* pk = keep + (size_t)p * nmax;
  __shared__ unsigned long long rows[kTile][kTile + 1];  // rows of the current chunk x words
  // lane j (j < cb, cb <= 64) keeps the running removed-bits of column block j
  unsigned long long remv = 0ull;
  int nkeep = 0;
  const int nchunks = (n + kTile - 1) / kTile;
  for (int c = 0; c < nchunks && nkeep < max_keep; ++c) {
    const int base = c * kTile;
    const int rows_here = min(n - base, kTile);
    // stage the chunk's mask rows (words c..cb-1; words < c are never written) in LDS
    for (int r = 0; r < rows_here; ++r) {
      if (lane >= c && lane < cb) rows[r][lane] = pm[(size_t)(base + r) * cb + lane];
    }
    __syncthreads();
    // greedy recurrence inside the chunk on the diagonal word (all lanes compute it redundantly)
    unsigned long long dead = __shfl(remv, c, 64);  // removed bits of this chunk so far
    unsigned long long kept = 0ull;
    for (int r = 0; r < rows_here; ++r) {
      if (!((dead >> r) & 1ull)) {
        kept |= 1ull << r;
        dead |= rows[r][c];
      }
    }
    // fold the kept rows into the running removed words of the later blocks
    if (lane > c && lane < cb) {
      unsigned long long acc = 0ull;
      for (int r = 0; r < rows_here; ++r)
        if ((kept >> r) & 1ull) acc |= rows[r][lane];
      remv |= acc;
    }
    // emit kept indices in ascending order: lane r writes at its rank among the kept bits
    if (lane < rows_here && ((kept >> lane) & 1ull)) {
      const int rank = __popcll(kept & ((1ull << lane) - 1ull));
      if (nkeep + rank < max_keep) pk[nkeep + rank] = base + lane;
    }
    nkeep += __popcll(kept);
    __syncthreads();
  }
  if (lane == 0) keep_count[p] = min(nkeep, max_keep);
}

#ifndef BGS_NO_DPP
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
  v |= dpp_u32<0xb1>(v);   // quad_perm:[1,0,3,2]
  v |= dpp_u32<0x4e>(v);   // quad_perm:[2,3,0,1]
  v |= dpp_u32<0x124>(v);  // row_ror:4
  v |= dpp_u32<0x128>(v);  // row_ror:8
  v |= dpp_u32<0x142>(v);  // row_bcast:15
  v |= dpp_u32<0x143>(v);  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#else
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
#endif
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  const uint32_t lo = wave_or_u32((uint32_t)v), hi = wave_or_u32((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// one 16-wave workgroup per problem; WPW = words owned per wave (cb <= 16 * WPW).
template <int WPW>
__global__ __launch_bounds__(1024) void nms_scan_wide_kernel(
    const unsigned long long* __restrict__ mask, const int* __restrict__ counts, int nmax, int cb,
    int max_keep, int* __restrict__ keep, int* __restrict__ keep_count) {
  constexpr int NW = 16, PF = 4;
  __shared__ unsigned long long s_kept[2];
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, w = bgs::uniform(threadIdx.x >> 6);
  const int n = min(counts[p], nmax);
  const unsigned long long* pm = mask + (size_t)p * nmax * cb;
  int* pk = keep + (size_t)p * nmax;
  const int nchunks = (n + kTile - 1) / kTile;
  unsigned long long remv[WPW];
#pragma unroll
  for (int q = 0; q < WPW; ++q) remv[q] = 0ull;
  // pre[d][q]: word w + 16q of row 64 * c + lane, for the chunk c that maps to slot d
  unsigned long long pre[PF][WPW];
  auto load_chunk = [&](int c, unsigned long long (&dst)[WPW]) {
    const int i = c * kTile + lane;
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
      const int j = w + NW * q;
      // (words < c of a row are never written by the mask kernel, and never needed)
      dst[q] = (c < nchunks && i < n && j < cb && j >= c) ? pm[(size_t)i * cb + j] : 0ull;
    }
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) load_chunk(d, pre[d]);
  int nkeep = 0;
  for (int c0 = 0; c0 < nchunks && nkeep < max_keep; c0 += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int c = c0 + d;
      if (c >= nchunks || nkeep >= max_keep) break;        // workgroup-uniform
      unsigned long long cur[WPW];
#pragma unroll
      for (int q = 0; q < WPW; ++q) cur[q] = pre[d][q];
      const int rows_here = min(n - c * kTile, kTile);
#pragma unroll
      for (int q = 0; q < WPW; ++q) {
        if (w + NW * q != c) continue;                     // wave-uniform: the owner of word c
        // greedy decision inside the chunk as a fixed point instead of a 64-step recurrence
        // (a lone wave issues ~1 scalar instruction per 5 cycles: the readlane loop cost 2.7 us
        // per chunk): kept = alive & ~{r : some kept s < r suppresses r}.  Row r's word holds
        // in its bits below r exactly those s.  After t rounds the first t rows are final, so
        // the iteration reaches the (unique) greedy solution; suppression chains are short and it
        // typically stops after 2-4 rounds of ~8 instructions.
        const unsigned long long rows_mask = rows_here == 64 ? ~0ull : ((1ull << rows_here) - 1ull);
        const unsigned long long alive = ~remv[q] & rows_mask;
        const unsigned long long below = cur[q] & ((1ull << lane) - 1ull);
        unsigned long long kept = alive;
        for (int it = 0; it <= kTile; ++it) {
          const unsigned long long sup = __ballot((below & kept) != 0ull);
          const unsigned long long next = alive & ~sup;
          if (next == kept) break;
          kept = next;
        }
        if (lane == 0) s_kept[c & 1] = kept;
      }
      // refill this slot for chunk c + PF only now: issued before the recurrence, its s_waitcnt
      // would also wait for these brand-new loads (a full memory round trip per chunk)
      load_chunk(c + PF, pre[d]);
      __syncthreads();
      const unsigned long long kept = s_kept[c & 1];
      const unsigned long long mine = ((kept >> lane) & 1ull) ? ~0ull : 0ull;
#pragma unroll
      for (int q = 0; q < WPW; ++q) {
        const int j = w + NW * q;
        if (j > c && j < cb) remv[q] |= wave_or_u64(cur[q] & mine);
      }
      if (w == 0 && lane < rows_here && ((kept >> lane) & 1ull)) {
        const int rank = __popcll(kept & ((1ull << lane) - 1ull));
        if (nkeep + rank < max_keep) pk[nkeep + rank] = c * kTile + lane;
      }
      nkeep += __popcll(kept);
    }
  }
  if (threadIdx.x == 0) keep_count[p] = min(nkeep, max_keep);
}

}  // namespace

extern "C" size_t bgs_nms_workspace_bytes(int P, int nmax) {
  if (P <= 0 || nmax <= 0) return 0;
  const size_t cb = (size_t)(nmax + kTile - 1) / kTile;
  return (size_t)P * nmax * cb * sizeof(unsigned long long);
}

extern "C" int bgs_nms_batched(const float* boxes, const int* counts, int P, int nmax,
                               float iou_thr, int iou_mode, int max_keep, int* keep,
                               int* keep_count, void* workspace, bgs_stream_t stream) {
  if (P < 0 || nmax <= 0) return BGS_ERR_INVALID_ARG;
  if (P == 0) return BGS_OK;
  if (!boxes || !counts || !keep || !keep_count || !workspace) return BGS_ERR_INVALID_ARG;
  const int cb = (nmax + kTile - 1) / kTile;
  if (cb > 64) return BGS_ERR_UNSUPPORTED;  // nmax <= 4096 (RPN uses nms_pre = 2000)
  if (max_keep <= 0 || max_keep > nmax) max_keep = nmax;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_mask_kernel, dim3((cb + kMaskWaves - 1) / kMaskWaves, cb, P), dim3(kTile * kMaskWaves), 0, st, boxes, counts, nmax, cb,
                     iou_thr, iou_mode, mask);
  // problems of more than 256 boxes -> 16-wave workgroups (RPN: 0.44 -> 0.16 ms per call; the 1230
  // per-class problems of 1000 boxes at test time: 1.59 -> 1.35 ms); tiny ones -> one wave each
  const char* env = getenv("BGS_NMS_SCAN");           // tests: 1 = narrow, 2 = wide
  const int force = env ? atoi(env) : 0;
  const bool wide = force ? force == 2 : cb > 4;
  if (!wide)
    hipLaunchKernelGGL(nms_scan_kernel, dim3(P), dim3(64), 0, st, mask, counts, nmax, cb, max_keep,
                       keep, keep_count);
  else if (cb <= 16)
    hipLaunchKernelGGL(nms_scan_wide_kernel<1>, dim3(P), dim3(1024), 0, st, mask, counts, nmax, cb,
                       max_keep, keep, keep_count);
  else if (cb <= 32)
    hipLaunchKernelGGL(nms_scan_wide_kernel<2>, dim3(P), dim3(1024), 0, st, mask, counts, nmax, cb,
                       max_keep, keep, keep_count);
  else
    hipLaunchKernelGGL(nms_scan_wide_kernel<4>, dim3(P), dim3(1024), 0, st, mask, counts, nmax, cb,
                       max_keep, keep, keep_count);
  BGS_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------
// The two gathers of RPNHead.get_bboxes_single around the NMS (rpn_head.py:92-103): the kept boxes of
// every (image, level) problem into fixed-shape rows (slots past keep_count score -1), and the final
// per-image top `max_num` selection — each ONE launch instead of six element-wise / gather / where
// launches (the proposal tail is pure launch latency: ~12 launches of ~4.5 us).
namespace {

__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes,
                                                         const int* __restrict__ keep,
                                                         const int* __restrict__ keep_n, int total,
                                                         int nmax, float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int r = i / nmax, slot = i - r * nmax;
  int k = keep[i];
  k = k < 0 ? 0 : (k > nmax - 1 ? nmax - 1 : k);
  const float* src = boxes + ((size_t)r * nmax + k) * 5;
  float* dst = out_boxes + (size_t)i * 5;
  const float s = src[4];
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
  dst[3] = src[3];
  dst[4] = s;
  out_scores[i] = slot < keep_n[r] ? s : -1.f;
}

__global__ __launch_bounds__(256) void gather_boxes_kernel(const float* __restrict__ flat,
                                                           const long long* __restrict__ idx,
                                                           const float* __restrict__ scores, int N, int T,
                                                           int num, float* __restrict__ props,
                                                           unsigned char* __restrict__ valid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * num) return;
  const int n = i / num;
  long long t = idx[i];
  t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
  const float* src = flat + ((size_t)n * T + t) * 5;
  float* dst = props + (size_t)i * 5;
#pragma unroll
  for (int c = 0; c < 5; ++c) dst[c] = src[c];
  valid[i] = scores[i] >= 0.f ? 1 : 0;
}


// The proposal tail of RPNHead.get_bboxes_single (rpn_head.py:99-103: `proposals = torch.cat(mlvl_proposals)`,
// `scores.topk(num)`, `proposals[topk_inds]`) in ONE launch instead of eleven (gather of the kept boxes, a memset,
// the eight launches of a 10,000-element radix select + sort, gather of the selected boxes: 83 us of the cfg[1]
// step, profiles/r6y_detector_prof_summary.md).  Each level's kept boxes already ARE in descending score order
// (NMS keeps positions of a score-sorted list in ascending order), so the per-image top `num` over the L levels
// is an L-way merge: the output rank of kept entry (level l, slot j) is
//     j + sum over l' != l of #{kept entries of l' that precede it},
// "precede" = larger score, or equal score and smaller concatenated index (= lower level) — the order of
// bgs_topk_sorted_f32's composites — found by a binary search of each other level's kept scores.  A workgroup
// stages the L kept-score lists of its image in LDS (L * nmax floats) and ranks its share of the entries.
// props [N, num, 5]: entries of rank < num, in rank order; slots past the number of kept boxes: zeros, valid = 0.
__device__ __forceinline__ unsigned merge_key_of(float f) {   // == key_of of csrc/topk.hip
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void nms_merge_select_kernel(const float* __restrict__ boxes,
                                                                const int* __restrict__ keep,
                                                                const int* __restrict__ keep_n, int L, int nmax,
                                                                int num, int parts, float* __restrict__ props,
                                                                unsigned char* __restrict__ valid) {
  extern __shared__ unsigned s_key[];                      // [L][nmax] kept scores of the image (as keys), descending per level
  __shared__ int s_n[16];
  const int n = blockIdx.x / parts, part = blockIdx.x - n * parts, tid = threadIdx.x;
  if (tid < L) {
    int c = keep_n[n * L + tid];
    s_n[tid] = c < 0 ? 0 : (c > nmax ? nmax : c);
  }
  __syncthreads();
  for (int e = tid; e < L * nmax; e += 1024) {
    const int l = e / nmax, j = e - l * nmax;
    float sc = -1.f;
    if (j < s_n[l]) {
      int k = keep[(size_t)(n * L + l) * nmax + j];
      k = k < 0 ? 0 : (k > nmax - 1 ? nmax - 1 : k);
      sc = boxes[((size_t)(n * L + l) * nmax + k) * 5 + 4];
    }
    // ranked by the order-preserving uint32 image of the float (the order bgs_topk_sorted_f32 sorts by): a TOTAL
    // order, so a NaN score (which sorts above +inf upstream) cannot make two entries claim one rank or leave a
    // slot below `total` unwritten, as float comparisons (false both ways against a NaN) could
    s_key[e] = merge_key_of(sc);
  }
  __syncthreads();
  int total = 0;
  for (int l = 0; l < L; ++l) total += s_n[l];
  // this workgroup's share of the (level, slot) entries
  for (int e = part * 1024 + tid; e < L * nmax; e += parts * 1024) {
    const int l = e / nmax, j = e - l * nmax;
    if (j >= s_n[l]) continue;
    const unsigned sc = s_key[e];
    int rank = j;
    for (int lo = 0; lo < L; ++lo) {
      if (lo == l) continue;
      const unsigned* v = s_key + lo * nmax;
      // number of entries of level `lo` that precede: v is non-increasing; lower levels win ties
      int a = 0, b = s_n[lo];
      while (a < b) {
        const int mid = (a + b) >> 1;
        const unsigned x = v[mid];
        const bool before = lo < l ? (x >= sc) : (x > sc);
        if (before) a = mid + 1;
        else b = mid;
      }
      rank += a;
    }
    if (rank < num) {
      int k = keep[(size_t)(n * L + l) * nmax + j];
      k = k < 0 ? 0 : (k > nmax - 1 ? nmax - 1 : k);
      const float* src = boxes + ((size_t)(n * L + l) * nmax + k) * 5;
      float* dst = props + ((size_t)n * num + rank) * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) dst[c] = src[c];
      valid[(size_t)n * num + rank] = 1;
    }
  }
  // slots nobody ranks into
  for (int r = (total < num ? total : num) + part * 1024 + tid; r < num; r += parts * 1024) {
    float* dst = props + ((size_t)n * num + r) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) dst[c] = 0.f;
    valid[(size_t)n * num + r] = 0;
  }
}

}  // namespace

// boxes [R, nmax, 5], keep [R, nmax] / keep_count [R] of bgs_nms_batched -> out_boxes [R, nmax, 5] =
// boxes[r, clamp(keep[r, slot])], out_scores [R, nmax] = that box's score for slot < keep_count[r], else -1.
extern "C" int bgs_nms_gather(const float* boxes, const int* keep, const int* keep_count, int R, int nmax,
                              float* out_boxes, float* out_scores, bgs_stream_t stream) {
  if (!boxes || !keep || !keep_count || !out_boxes || !out_scores || R <= 0 || nmax <= 0)
    return BGS_ERR_INVALID_ARG;
  if ((long long)R * nmax > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  const int total = R * nmax;
  hipLaunchKernelGGL(nms_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes, keep, keep_count, total, nmax, out_boxes, out_scores);
  BGS_RETURN_LAUNCH_STATUS();
}

// flat [N, T, 5], idx [N, num] int64 (row positions, e.g. of bgs_topk_sorted_f32), scores [N, num] ->
// props [N, num, 5] = flat[n, idx[n, j]], valid [N, num] uint8 = scores >= 0.
extern "C" int bgs_gather_boxes(const float* flat, const long long* idx, const float* scores, int N, int T,
                                int num, float* props, unsigned char* valid, bgs_stream_t stream) {
  if (!flat || !idx || !scores || !props || !valid || N <= 0 || T <= 0 || num <= 0)
    return BGS_ERR_INVALID_ARG;
  if ((long long)N * num > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gather_boxes_kernel, dim3((unsigned)((N * num + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, flat, idx, scores, N, T, num, props, valid);
  BGS_RETURN_LAUNCH_STATUS();
}

// boxes [N * L, nmax, 5] (each row sorted by descending score: the decoded pre-NMS candidates of one image and
// level), keep [N * L, nmax] / keep_count [N * L] of bgs_nms_batched -> props [N, num, 5] = the `num` best kept
// boxes of each image over its L levels in descending score order (ties: lower level, then NMS order),
// valid [N, num] uint8 (0 and a zero box past the number of kept boxes).  One launch; L <= 16, L * nmax * 4 bytes
// of LDS (<= 64 KB).  Replaces bgs_nms_gather + bgs_topk_sorted_f32 + bgs_gather_boxes on the proposal path
// (mmdet/models/anchor_heads/rpn_head.py:99-103).
extern "C" int bgs_nms_merge_select(const float* boxes, const int* keep, const int* keep_count, int N, int L,
                                    int nmax, int num, float* props, unsigned char* valid, bgs_stream_t stream) {
  if (!boxes || !keep || !keep_count || !props || !valid || N <= 0 || L <= 0 || nmax <= 0 || num <= 0)
    return BGS_ERR_INVALID_ARG;
  if (L > 16 || (size_t)L * nmax * sizeof(float) > 64 * 1024) return BGS_ERR_UNSUPPORTED;
  const int parts = (L * nmax + 1023) / 1024 < 8 ? (L * nmax + 1023) / 1024 : 8;
  hipLaunchKernelGGL(nms_merge_select_kernel, dim3((unsigned)(N * parts)), dim3(1024),
                     (size_t)L * nmax * sizeof(float), (hipStream_t)stream, boxes, keep, keep_count, L, nmax, num,
                     parts, props, valid);
  BGS_RETURN_LAUNCH_STATUS();
}
