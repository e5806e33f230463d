// RandomSampler for the RPN (mmdet/core/bbox/samplers/base_sampler.py:35-78 +
// random_sampler.py:19-53) in ONE launch per batch: from the assigner's result for all 268,569
// anchors of an image pick exactly min(int(num * pos_fraction), #pos) positives and
// min(num - #pos_sampled [, neg_pos_ub * max(#pos_sampled, 1)], #neg) negatives, uniformly without
// replacement, and emit them as byte masks.  The reference shuffles index lists with numpy on the
// host (4 synchronisations per image); the tensor-op form of this package (assign.py) needed a key
// tensor, two top-k's over 268k int64 keys and a dozen element-wise launches per image.
//
// One 1024-thread workgroup per image.  Every anchor gets a 32-bit key from a BIJECTIVE mixer of
// (index + per-draw offset): keys of different anchors differ, so "the k anchors with the smallest
// keys" is a uniform k-subset with no ties to break; the k-th smallest key is found by a 3-pass
// radix select (11 + 11 + 10 bits) with LDS histograms (uniform keys -> no bin contention).
#include "bgs_common.h"

namespace {

constexpr int kThreadsS = 1024;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32: every step is a bijection
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// Key sources of the radix select: the anchors of one class (keys recomputed from the index), or
// a short list of pre-filtered keys.
struct AnchorKeys {
  const int* assigned;
  int A, want;          // want 1: assigned > 0, 0: assigned == 0
  uint32_t offset;
  __device__ int size() const { return A; }
  __device__ bool get(int i, uint32_t& key) const {
    const int a = assigned[i];
    key = mix32((uint32_t)i + offset);
    return want ? (a > 0) : (a == 0);
  }
};
struct ListKeys {
  const uint32_t* keys;
  int n;
  __device__ int size() const { return n; }
  __device__ bool get(int i, uint32_t& key) const {
    key = keys[i];
    return true;
  }
};

// k-th smallest key (1-based k) of a source.  Returns through shared memory; all threads of the
// 1024-thread workgroup call it.  1 <= k <= number of keys in the source.
template <class Src>
__device__ uint32_t kth_smallest_key(const Src src, int k, int* hist, int* s_tmp) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0, pmask = 0;
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  int kk = k;
  const int n = src.size();
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int b = tid; b < 2048; b += kThreadsS) hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kThreadsS) {
      uint32_t key;
      if (!src.get(i, key)) continue;
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1);
    }
    __syncthreads();
    if (tid < 64) {   // one wave walks the bins upwards until the running count reaches kk
      int acc = 0, found = -1, before = 0;
      for (int base = 0; base < nb && found < 0; base += 64) {
        const int c = hist[base + tid];
        int inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(inc, off, 64);
          if (tid >= off) inc += t;
        }
        const int total = __shfl(inc, 63, 64);
        const unsigned long long m = __ballot(acc + inc >= kk);
        if (m) {
          const int lane = __ffsll((long long)m) - 1;
          found = base + lane;
          before = acc + __shfl(inc - c, lane, 64);
        }
        acc += total;
      }
      if (tid == 0) {
        s_tmp[0] = found;
        s_tmp[1] = kk - before;
      }
    }
    __syncthreads();
    prefix |= (uint32_t)s_tmp[0] << shifts[pass];
    pmask |= (uint32_t)(nb - 1) << shifts[pass];
    kk = s_tmp[1];
    __syncthreads();
  }
  return prefix;
}

// Three launches per batch (the single-workgroup-per-image form walked all 268,569 anchors five
// times from ONE CU: 320 us, 2.4 % of the training step):
//   scan    (G x N workgroups)  class counts + the keys that can matter: every positive's key and
//                                the negatives' keys below a conservative threshold T0 (expected
//                                8 * num of them, so the num-th smallest is among them with
//                                overwhelming probability), appended to per-image lists;
//   select  (N workgroups)       k_pos / k_neg, then the k-th smallest key of each class from the
//                                short lists — or, if a list overflowed or came up short (tiny
//                                n_neg, huge num), from the anchors themselves as before;
//   mark    (G x N workgroups)   pos_mask / neg_mask = key <= threshold of the anchor's class.
// Same keys, same thresholds, same result as the one-kernel form.
constexpr int kCand = 8192;      // list capacity per image and class
constexpr int kCtr = 8;          // counters per image: n_pos, n_neg, list_pos, list_neg, 4 x select

struct SampleWs {
  int* ctr;            // [N, kCtr]
  uint32_t* cand_pos;  // [N, kCand]
  uint32_t* cand_neg;  // [N, kCand]
};

__device__ __forceinline__ uint32_t image_offset(uint64_t seed, const long long* draw, int n) {
  const uint64_t d = draw ? (uint64_t)draw[0] : 0ull;
  uint64_t h = seed + 0x9E3779B97F4A7C15ull * (d + 1ull) + 0xD1B54A32D192ED03ull * ((uint64_t)n + 1ull);
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27;
  return (uint32_t)(h >> 16);
}

__global__ __launch_bounds__(256) void sample_scan_kernel(const int* __restrict__ assigned_all, int A,
                                                          uint32_t t0_neg, uint64_t seed,
                                                          const long long* __restrict__ draw,
                                                          SampleWs ws) {
  // Candidates are collected in LDS first (one LDS atomic per wave and list: ~100 cycles) and appended to the
  // image's lists with ONE global atomic per workgroup and list at the end; the first version reserved global
  // slots per wave and iteration — a returning device atomic (~1.5 us) in almost every one of the 32 iterations:
  // 41 us for a kernel that reads 2 MB.  (List order is arrival order either way; the select kernel takes the
  // k-th smallest KEY, so the result does not depend on it.)  Entries beyond the LDS capacity take the old path.
  constexpr int kLp = 256, kLn = 768;
  __shared__ int s_cnt[2];
  __shared__ int s_n[2], s_base[2];
  __shared__ uint32_t s_lp[kLp], s_ln[kLn];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int* assigned = assigned_all + (size_t)n * A;
  const uint32_t offset = image_offset(seed, draw, n);
  int* ctr = ws.ctr + n * kCtr;
  uint32_t* lp = ws.cand_pos + (size_t)n * kCand;
  uint32_t* ln = ws.cand_neg + (size_t)n * kCand;
  const int chunk = (A + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * chunk, hi = min(A, lo + chunk);
  if (tid < 2) {
    s_cnt[tid] = 0;
    s_n[tid] = 0;
  }
  __syncthreads();
  int c_pos = 0, c_neg = 0;
  const int lane = tid & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  // wave-uniform trip count (the ballots below need converged lanes)
  for (int i0 = lo; i0 < hi; i0 += 256) {
    const int i = i0 + tid;
    const int a = i < hi ? assigned[i] : -1;
    const uint32_t key = mix32((uint32_t)i + offset);
    const bool is_pos = a > 0, is_neg = a == 0;
    c_pos += is_pos;
    c_neg += is_neg;
    const bool cand_neg = is_neg && key <= t0_neg;
    const unsigned long long mp = __ballot(is_pos), mn = __ballot(cand_neg);
    if (mp) {
      const int first = __ffsll((long long)mp) - 1;
      int base = 0;
      if (lane == first) base = atomicAdd(&s_n[0], __popcll(mp));
      base = __shfl(base, first, 64);
      const int slot = base + __popcll(mp & below);
      if (is_pos) {
        if (slot < kLp) s_lp[slot] = key;
        else {                                           // LDS list full: straight to the image's list
          const int g = atomicAdd(&ctr[2], 1);
          if (g < kCand) lp[g] = key;
        }
      }
    }
    if (mn) {
      const int first = __ffsll((long long)mn) - 1;
      int base = 0;
      if (lane == first) base = atomicAdd(&s_n[1], __popcll(mn));
      base = __shfl(base, first, 64);
      const int slot = base + __popcll(mn & below);
      if (cand_neg) {
        if (slot < kLn) s_ln[slot] = key;
        else {
          const int g = atomicAdd(&ctr[3], 1);
          if (g < kCand) ln[g] = key;
        }
      }
    }
  }
  c_pos = bgs::wave_sum_i(c_pos);
  c_neg = bgs::wave_sum_i(c_neg);
  if ((tid & 63) == 0) {
    atomicAdd(&s_cnt[0], c_pos);
    atomicAdd(&s_cnt[1], c_neg);
  }
  __syncthreads();
  const int np = min(s_n[0], kLp), nn = min(s_n[1], kLn);
  if (tid == 0) {
    if (s_cnt[0]) atomicAdd(&ctr[0], s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&ctr[1], s_cnt[1]);
    s_base[0] = np ? atomicAdd(&ctr[2], np) : 0;
    s_base[1] = nn ? atomicAdd(&ctr[3], nn) : 0;
  }
  __syncthreads();
  for (int i = tid; i < np; i += 256)
    if (s_base[0] + i < kCand) lp[s_base[0] + i] = s_lp[i];
  for (int i = tid; i < nn; i += 256)
    if (s_base[1] + i < kCand) ln[s_base[1] + i] = s_ln[i];
}

// k-th smallest key (1-based) of a short list whose keys are all <= max_key: ONE 8192-bin histogram of the 13 bits
// below max_key's leading zeros, a workgroup-wide scan for the bin that holds rank k, and a one-wave rank of the
// (on average <= 1, at most 64) keys inside that bin.  *done = 0 when the bin holds more than 64 keys (the caller
// falls back to the three-pass select).  The three-pass form cost ~3.5 us per pass (a one-wave walk over up to
// 2048 bins) — 23 us for the two lists of an RPN image.
constexpr int kBinsL = 8192;
__device__ uint32_t kth_smallest_short(const uint32_t* __restrict__ keys, int n, int k, uint32_t max_key,
                                       int* hist, int* s_tmp, uint32_t* s_small, int* done) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bits = 32 - __clz((int)(max_key | 1u));
  const int shift = bits > 13 ? bits - 13 : 0;
  for (int b = tid; b < kBinsL; b += kThreadsS) hist[b] = 0;
  if (tid == 0) {
    s_tmp[2] = 0;
    s_tmp[3] = 0;
    s_tmp[4] = 0;                                           // "found": exactly one lane matched the rank
  }
  __syncthreads();
  for (int i = tid; i < n; i += kThreadsS) atomicAdd(&hist[keys[i] >> shift], 1);
  __syncthreads();
  // scan: thread t owns bins 8t .. 8t + 7 (ascending keys)
  int cnt[8], mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cnt[j] = hist[tid * 8 + j];
    mine += cnt[j];
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_tmp[8 + wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_tmp[8 + w];
  incl += base;
  const int excl = incl - mine;
  if (excl < k && k <= incl) {                              // exactly one thread
    int c = excl, bin = tid * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (k <= c + cnt[j]) {
        bin = tid * 8 + j;
        break;
      }
      c += cnt[j];
    }
    s_tmp[0] = bin;
    s_tmp[1] = k - c;                                       // rank inside the bin, 1-based
  }
  __syncthreads();
  const uint32_t bin = (uint32_t)s_tmp[0];
  const int kin = s_tmp[1];
  for (int i = tid; i < n; i += kThreadsS) {
    const uint32_t key = keys[i];
    if ((key >> shift) == bin) {
      const int slot = atomicAdd(&s_tmp[2], 1);
      if (slot < 64) s_small[slot] = key;
    }
  }
  __syncthreads();
  const int m = s_tmp[2];
  if (m <= 64 && tid < 64) {
    const uint32_t mykey = tid < m ? s_small[tid] : 0xffffffffu;
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += s_small[j] < mykey;
    // distinct keys (mix32 is a bijection of the anchor index): exactly one lane.  A key source with duplicates
    // would leave no lane at rank kin - 1 (ties share the lower rank): then `found` stays 0, *done = 0, and the
    // caller's three-pass select — which handles duplicates — runs instead of a stale threshold being returned
    if (tid < m && rank == kin - 1) {
      s_tmp[3] = (int)mykey;
      atomicAdd(&s_tmp[4], 1);
    }
  }
  __syncthreads();
  *done = (m <= 64 && s_tmp[4] == 1) ? 1 : 0;
  const uint32_t r = (uint32_t)s_tmp[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kThreadsS) void sample_select_kernel(
    const int* __restrict__ assigned_all, int A, int num, int n_exp_pos, float neg_pos_ub,
    uint32_t t0_neg, uint64_t seed, const long long* __restrict__ draw, SampleWs ws) {
  __shared__ int hist[kBinsL];
  __shared__ int s_tmp[8 + kThreadsS / 64];
  __shared__ uint32_t s_small[64];
  const int n = blockIdx.x;
  const int* assigned = assigned_all + (size_t)n * A;
  const uint32_t offset = image_offset(seed, draw, n);
  int* ctr = ws.ctr + n * kCtr;
  const int n_pos = ctr[0], n_neg = ctr[1], l_pos = ctr[2], l_neg = ctr[3];
  const int k_pos = min(n_exp_pos, n_pos);
  int n_exp_neg = num - k_pos;
  if (neg_pos_ub >= 0.f) {
    const int ub = (int)(neg_pos_ub * (float)max(k_pos, 1));
    n_exp_neg = min(n_exp_neg, ub);
  }
  const int k_neg = max(0, min(n_exp_neg, n_neg));
  // thresholds (only when a strict subset is wanted)
  uint32_t thr_pos = 0xffffffffu, thr_neg = 0xffffffffu;
  if (k_pos > 0 && k_pos < n_pos) {
    int done = 0;
    if (l_pos <= kCand) {     // the list holds every positive's key
      const uint32_t* lp = ws.cand_pos + (size_t)n * kCand;
      thr_pos = kth_smallest_short(lp, l_pos, k_pos, 0xffffffffu, hist, s_tmp, s_small, &done);
      if (!done) thr_pos = kth_smallest_key(ListKeys{lp, l_pos}, k_pos, hist, s_tmp);
    } else {
      thr_pos = kth_smallest_key(AnchorKeys{assigned, A, 1, offset}, k_pos, hist, s_tmp);
    }
  }
  if (k_neg > 0 && k_neg < n_neg) {
    // the list holds ALL negative keys <= t0_neg: its k-th smallest is the global one iff it has
    // at least k entries and none was dropped
    if (l_neg <= kCand && (l_neg >= k_neg || t0_neg == 0xffffffffu)) {
      const uint32_t* ln = ws.cand_neg + (size_t)n * kCand;
      int done = 0;
      thr_neg = kth_smallest_short(ln, l_neg, k_neg, t0_neg, hist, s_tmp, s_small, &done);
      if (!done) thr_neg = kth_smallest_key(ListKeys{ln, l_neg}, k_neg, hist, s_tmp);
    } else {
      thr_neg = kth_smallest_key(AnchorKeys{assigned, A, 0, offset}, k_neg, hist, s_tmp);
    }
  }
  if (threadIdx.x == 0) {
    ctr[4] = (int)thr_pos;
    ctr[5] = (int)thr_neg;
    ctr[6] = k_pos > 0;
    ctr[7] = k_neg > 0;
  }
}

__global__ __launch_bounds__(256) void sample_mark_kernel(const int* __restrict__ assigned_all, int A,
                                                          uint64_t seed,
                                                          const long long* __restrict__ draw,
                                                          SampleWs ws, uint8_t* __restrict__ pos_mask,
                                                          uint8_t* __restrict__ neg_mask) {
  const int n = blockIdx.y;
  const int* assigned = assigned_all + (size_t)n * A;
  uint8_t* pm = pos_mask + (size_t)n * A;
  uint8_t* nm = neg_mask + (size_t)n * A;
  const uint32_t offset = image_offset(seed, draw, n);
  const int* ctr = ws.ctr + n * kCtr;
  const uint32_t thr_pos = (uint32_t)ctr[4], thr_neg = (uint32_t)ctr[5];
  const bool any_pos = ctr[6] != 0, any_neg = ctr[7] != 0;
  const int chunk = (A + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * chunk, hi = min(A, lo + chunk);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const int a = assigned[i];
    const uint32_t key = mix32((uint32_t)i + offset);
    pm[i] = (a > 0 && any_pos && key <= thr_pos) ? 1 : 0;
    nm[i] = (a == 0 && any_neg && key <= thr_neg) ? 1 : 0;
  }
}

}  // namespace

extern "C" size_t bgs_sample_pos_neg_workspace_bytes(int N) {
  if (N <= 0) return 0;
  return (size_t)N * (kCtr * sizeof(int) + 2 * (size_t)kCand * sizeof(uint32_t));
}

extern "C" int bgs_sample_pos_neg(const int* assigned, int N, int A, int num, float pos_fraction,
                                  float neg_pos_ub, uint64_t seed, const long long* draw_counter,
                                  uint8_t* pos_mask, uint8_t* neg_mask, void* workspace,
                                  bgs_stream_t stream) {
  if (N < 0 || A <= 0 || num <= 0 || !(pos_fraction >= 0.f && pos_fraction <= 1.f))
    return BGS_ERR_INVALID_ARG;
  if (N == 0) return BGS_OK;
  if (!assigned || !pos_mask || !neg_mask || !workspace) return BGS_ERR_INVALID_ARG;
  const int n_exp_pos = (int)((double)num * (double)pos_fraction);
  SampleWs ws;
  ws.ctr = reinterpret_cast<int*>(workspace);
  ws.cand_pos = reinterpret_cast<uint32_t*>(ws.ctr + (size_t)N * kCtr);
  ws.cand_neg = ws.cand_pos + (size_t)N * kCand;
  // expected 8 * num negative keys below t0 when every anchor is a negative
  const double frac = 8.0 * (double)num / (double)A;
  const uint32_t t0_neg = frac >= 1.0 ? 0xffffffffu : (uint32_t)(frac * 4294967296.0);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws.ctr, 0, (size_t)N * kCtr * sizeof(int), st) != hipSuccess)
    return BGS_ERR_LAUNCH;
  int G = (A + 4095) / 4096;
  G = G < 1 ? 1 : (G > 128 ? 128 : G);
  hipLaunchKernelGGL(sample_scan_kernel, dim3(G, N), dim3(256), 0, st, assigned, A, t0_neg, seed,
                     draw_counter, ws);
  hipLaunchKernelGGL(sample_select_kernel, dim3(N), dim3(kThreadsS), 0, st, assigned, A, num,
                     n_exp_pos, neg_pos_ub, t0_neg, seed, draw_counter, ws);
  hipLaunchKernelGGL(sample_mark_kernel, dim3(G, N), dim3(256), 0, st, assigned, A, seed,
                     draw_counter, ws, pos_mask, neg_mask);
  BGS_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------
// RandomSampler for the RoI head (two_stage.py:192-210 with add_gt_as_proposals): per image a
// FIXED-SIZE index list into the candidate array (GT boxes first, then the proposals): the sampled
// positives (at most int(num * pos_fraction), a uniform subset when there are more) first, then
// uniformly sampled negatives, then padding (index of slot 0, valid = 0) when fewer than `num`
// candidates exist.  One workgroup per image sorts (class, bijective key) composites of the
// <= 4096 candidates in LDS (bitonic) — replaces a key tensor, two top-k's and ~15 element-wise
// launches per image of the tensor-op form (assign.sample_fixed).
namespace {

constexpr int kMaxCand = 4096;
constexpr int kMaxImgsS = 16;

struct CandTable {
  const int* assigned[kMaxImgsS];   // [count] int32: -1 / 0 / gt index + 1
  int count[kMaxImgsS];
  int n_gt[kMaxImgsS];              // leading candidates that are GT boxes (add_gt_as_proposals)
  int* gt_ind;                      // null, or [N, num]: assigned[inds] - 1 (pos_assigned_gt_inds; -1 = none)
  uint8_t* is_gt;                   // null, or [N, num]: inds < n_gt (SamplingResult.pos_is_gt)
};

__global__ __launch_bounds__(kThreadsS) void sample_rois_kernel(CandTable T, int num, int n_exp_pos,
                                                                uint64_t seed,
                                                                const long long* __restrict__ draw,
                                                                long long* __restrict__ inds,
                                                                uint8_t* __restrict__ is_pos,
                                                                uint8_t* __restrict__ valid) {
  __shared__ unsigned long long buf[kMaxCand];
  __shared__ int s_cnt[2];
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int A = T.count[n];
  const int* assigned = T.assigned[n];
  const uint64_t d = draw ? (uint64_t)draw[0] : 0ull;
  uint64_t h = seed + 0x9E3779B97F4A7C15ull * (d + 1ull) + 0xD1B54A32D192ED03ull * ((uint64_t)n + 1ull);
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27;
  const uint32_t offset = (uint32_t)(h >> 16);
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  // composite = class (0 pos, 1 neg, 2 other / padding) : 2 | key : 32 | index : 16 .. unique
  // The network is sized to the next power of two >= the candidate count (cfg[1]: 2020 -> 2048, 66 stages of
  // one compare-exchange per thread instead of 78 stages of two): the composites are unique and the padding
  // is the largest value, so the first `A` sorted entries do not depend on the network size.
  int P = 64;
  while (P < A) P <<= 1;
  int c_pos = 0, c_neg = 0;
  for (int i = tid; i < P; i += kThreadsS) {
    unsigned long long comp = ~0ull;
    if (i < A) {
      const int a = assigned[i];
      const unsigned long long cls = a > 0 ? 0ull : (a == 0 ? 1ull : 2ull);
      comp = (cls << 60) | ((unsigned long long)mix32((uint32_t)i + offset) << 16) |
             (unsigned long long)i;
      c_pos += a > 0;
      c_neg += a == 0;
    }
    buf[i] = comp;
  }
  c_pos = bgs::wave_sum_i(c_pos);
  c_neg = bgs::wave_sum_i(c_neg);
  if ((tid & 63) == 0) {
    atomicAdd(&s_cnt[0], c_pos);
    atomicAdd(&s_cnt[1], c_neg);
  }
  __syncthreads();
  // bitonic sort, ascending
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int j = tid; j < P / 2; j += kThreadsS) {
        const int lo = 2 * j - (j & (stride - 1));
        const int hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const unsigned long long a = buf[lo], b = buf[hi];
        if ((a > b) == asc) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
      bgs::bitonic_stage_sync(size, stride);               // wave-scope for stride <= 64 (bgs_common.h)
    }
  }
  __syncthreads();
  const int n_pos = s_cnt[0], n_neg = s_cnt[1];
  const int k_pos = min(n_exp_pos, n_pos);
  const long long first = (long long)(buf[0] & 0xffffull);
  for (int j = tid; j < num; j += kThreadsS) {
    long long idx = first;
    uint8_t p = 0, v = 0;
    if (j < k_pos) {
      idx = (long long)(buf[j] & 0xffffull);
      p = 1;
      v = 1;
    } else if (j - k_pos < n_neg) {
      idx = (long long)(buf[n_pos + (j - k_pos)] & 0xffffull);
      v = 1;
    }
    inds[(size_t)n * num + j] = idx;
    is_pos[(size_t)n * num + j] = p;
    valid[(size_t)n * num + j] = v;
    if (T.gt_ind) T.gt_ind[(size_t)n * num + j] = T.assigned[n][idx] - 1;
    if (T.is_gt) T.is_gt[(size_t)n * num + j] = idx < (long long)T.n_gt[n] ? 1 : 0;
  }
}

}  // namespace

extern "C" int bgs_sample_rois_ex(const int* const* host_assigned, const int* host_counts,
                                  const int* host_gt_counts, int N, int num, float pos_fraction,
                                  uint64_t seed, const long long* draw_counter, long long* inds,
                                  uint8_t* is_pos, uint8_t* valid, int* gt_ind, uint8_t* is_gt,
                                  bgs_stream_t stream);

extern "C" int bgs_sample_rois(const int* const* host_assigned, const int* host_counts, int N,
                               int num, float pos_fraction, uint64_t seed,
                               const long long* draw_counter, long long* inds, uint8_t* is_pos,
                               uint8_t* valid, bgs_stream_t stream) {
  return bgs_sample_rois_ex(host_assigned, host_counts, nullptr, N, num, pos_fraction, seed, draw_counter,
                            inds, is_pos, valid, nullptr, nullptr, stream);
}

// + gt_ind [N, num] int32 = assigned[inds] - 1 (`pos_assigned_gt_inds` of the sampled rows, sampling_result.py:
// 7-24; -1 for negatives) and is_gt [N, num] uint8 = inds < host_gt_counts[n] (`pos_is_gt`: the GT boxes that
// `add_gt_as_proposals` put in front of the candidates); either may be NULL.
extern "C" int bgs_sample_rois_ex(const int* const* host_assigned, const int* host_counts,
                                  const int* host_gt_counts, int N, int num, float pos_fraction,
                                  uint64_t seed, const long long* draw_counter, long long* inds,
                                  uint8_t* is_pos, uint8_t* valid, int* gt_ind, uint8_t* is_gt,
                                  bgs_stream_t stream) {
  if (N < 0 || N > kMaxImgsS || num <= 0 || !(pos_fraction >= 0.f && pos_fraction <= 1.f))
    return BGS_ERR_INVALID_ARG;
  if (N == 0) return BGS_OK;
  if (!host_assigned || !host_counts || !inds || !is_pos || !valid) return BGS_ERR_INVALID_ARG;
  CandTable T;
  for (int i = 0; i < kMaxImgsS; ++i) {
    T.assigned[i] = nullptr;
    T.count[i] = 0;
    T.n_gt[i] = 0;
  }
  T.gt_ind = gt_ind;
  T.is_gt = is_gt;
  for (int i = 0; i < N; ++i) {
    if (host_counts[i] <= 0 || host_counts[i] > kMaxCand || !host_assigned[i]) return BGS_ERR_UNSUPPORTED;
    T.assigned[i] = host_assigned[i];
    T.count[i] = host_counts[i];
    T.n_gt[i] = host_gt_counts ? host_gt_counts[i] : 0;
  }
  const int n_exp_pos = (int)((double)num * (double)pos_fraction);
  hipLaunchKernelGGL(sample_rois_kernel, dim3(N), dim3(kThreadsS), 0, (hipStream_t)stream, T, num,
                     n_exp_pos, seed, draw_counter, inds, is_pos, valid);
  BGS_RETURN_LAUNCH_STATUS();
}
