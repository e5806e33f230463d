// Batched sorted top-k of fp32 rows for gfx950 (MI355X): the RPN's proposal pre-selection.
//
// Replaces the `scores.topk(cfg.nms_pre)` per FPN level and the final `scores.topk(num)` of
// RPNHead.get_bboxes_single (mmdet/models/anchor_heads/rpn_head.py:79-83,99-103).  A training
// iteration needs the 2000 largest of 201,600 / 50,400 / 12,600 / 3,150 / 819 objectness logits
// per image and level, then the 2000 best of the 10,000 NMS survivors per image.  The torch
// library answers each of the six calls with its own multi-kernel radix select + sort (15-40
// launches, 0.5 ms per iteration in total, profiles/r2v_detector_prof_summary.md); here ALL rows of
// a call set (different lengths, different k) go through the same five launches:
//
//   3 x hist     (G x P workgroups)  LDS histogram of the next 11 / 11 / 10 key bits of the
//                                    elements that match the prefix found so far -> global bins; the
//                                    digit of the PREVIOUS pass (bins walked from the top until k is
//                                    covered) is found by every workgroup for itself first
//   collect (G x P workgroups)       keys above the exact 32-bit threshold (and exactly as many
//                                    equal to it as are still missing) -> a k-entry list
//   sort    (P workgroups)           bitonic sort of the <= 4096 (key, index) composites in LDS,
//                                    descending; values / indices out, zero fill beyond k.
//
// Keys are the usual order-preserving uint32 image of the floats; composites (key << 32 | ~index)
// sort larger values first and, among equal values, smaller indices first.  Which of several
// elements EQUAL to the threshold value are taken is unspecified (as in torch.topk).
// HBM-bound in principle (4 reads of the rows: 6.4 MB for the largest level of two images), in
// practice latency-bound: five short launches.
#include "bgs_common.h"

namespace {

constexpr int kMaxProblems = 64;
constexpr int kMaxK = 4096;
constexpr int kBins = 2048;
constexpr int kChunk = 4096;     // elements per workgroup of the streaming kernels
constexpr int kState = 8;        // per problem: prefix, pmask, k_rem, n_gt, n_eq

struct TopkProblems {
  const float* row[kMaxProblems];
  int len[kMaxProblems];
  int k[kMaxProblems];
  // element i of a row lives at row[(i / inner) * pitch + (i % inner)]; inner == 0: row[i].
  // (The RPN's objectness logits are the first A of 5A channels of the fused head output.)
  int inner[kMaxProblems];
  int pitch[kMaxProblems];
};

__device__ __forceinline__ float topk_elem(const TopkProblems& pr, int p, const float* row, int i) {
  const int inner = pr.inner[p];
  if (inner == 0) return row[i];
  const int q = i / inner;
  return row[(size_t)q * pr.pitch[p] + (i - q * inner)];
}

struct TopkWs {
  int* hist;                    // [P, 3, kBins]: one histogram per pass
  unsigned* state;              // [P, kState]
  unsigned long long* list;     // [P, kmax]
};

__device__ __forceinline__ unsigned key_of(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float value_of(unsigned key) {
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

__device__ __forceinline__ int shift_of(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __forceinline__ int bins_of(int pass) { return pass == 2 ? 1024 : 2048; }

// The digit of one pass, by all 256 threads of a workgroup: the bins of `g` (the finished global histogram of
// that pass) are walked from the TOP (largest keys) until k_rem is covered.  Returns the digit and the number
// still to take inside it (>= 1).  Every workgroup of the NEXT launch does this for itself (2048 bins from the
// L2, one scan) — the separate one-wave pick launches (3 x 6 us + boundaries) are gone.
__device__ __forceinline__ void topk_pick(const int* __restrict__ g, int nb, int k_rem, int* s_wave, int* s_res,
                                          int& digit, int& k_in) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = nb / 256;                                // bins per thread, thread 0 = top bins
  int cnt[8];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cnt[j] = j < per ? g[nb - 1 - (tid * per + j)] : 0;
    mine += cnt[j];
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  incl += base;
  const int excl = incl - mine;
  if (excl < k_rem && k_rem <= incl) {                     // exactly one thread (k_rem >= 1)
    int c = excl, d = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < per) {
        if (k_rem <= c + cnt[j]) {
          d = nb - 1 - (tid * per + j);
          break;
        }
        c += cnt[j];
      }
    }
    s_res[0] = d;
    s_res[1] = k_rem - c;
  }
  __syncthreads();
  digit = s_res[0];
  k_in = s_res[1];
  __syncthreads();                                         // (s_wave / s_res may be reused)
}

// state words per problem: [0..2] prefix, mask, k_rem after the first digit (written by workgroup 0 of pass 1,
// read by pass 2), [5..7] the same after the second digit (written by pass 2, read by collect); [3], [4] the
// list counters of collect.  Workgroups of one launch never read what a sibling writes.
__global__ __launch_bounds__(256) void topk_hist_kernel(TopkProblems pr, TopkWs ws, int pass) {
  __shared__ int h[kBins];
  __shared__ int s_wave[4], s_res[2];
  const int p = blockIdx.y, tid = threadIdx.x;
  const int len = pr.len[p];
  const int lo = blockIdx.x * kChunk;
  if (lo >= len) return;                                  // workgroup-uniform
  const int hi = min(len, lo + kChunk);
  const int k = min(pr.k[p], len);
  if (k <= 0) return;                                     // nothing is selected from this row
  const int nb = bins_of(pass), shift = shift_of(pass);
  for (int b = tid; b < nb; b += 256) h[b] = 0;
  unsigned* st = ws.state + p * kState;
  unsigned prefix = 0, pmask = 0;
  if (pass > 0) {
    int k_rem = k;
    if (pass == 2) {
      prefix = st[0];
      pmask = st[1];
      k_rem = (int)st[2];
    }
    int digit, k_in;
    topk_pick(ws.hist + ((size_t)p * 3 + (pass - 1)) * kBins, bins_of(pass - 1), k_rem, s_wave, s_res, digit, k_in);
    prefix |= (unsigned)digit << shift_of(pass - 1);
    pmask |= (unsigned)(bins_of(pass - 1) - 1) << shift_of(pass - 1);
    if (blockIdx.x == 0 && tid == 0) {
      unsigned* o = st + (pass == 1 ? 0 : 5);
      o[0] = prefix;
      o[1] = pmask;
      o[2] = (unsigned)k_in;
    }
  }
  __syncthreads();
  const float* row = pr.row[p];
  for (int i = lo + tid; i < hi; i += 256) {
    const unsigned key = key_of(topk_elem(pr, p, row, i));
    if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & (nb - 1)], 1);
  }
  __syncthreads();
  int* g = ws.hist + ((size_t)p * 3 + pass) * kBins;
  for (int b = tid; b < nb; b += 256)
    if (h[b]) atomicAdd(&g[b], h[b]);
}

__global__ __launch_bounds__(256) void topk_collect_kernel(TopkProblems pr, TopkWs ws, int kmax) {
  const int p = blockIdx.y, tid = threadIdx.x;
  const int len = pr.len[p];
  const int lo = blockIdx.x * kChunk;
  if (lo >= len) return;
  const int hi = min(len, lo + kChunk);
  const int k = min(pr.k[p], len);
  unsigned* st = ws.state + p * kState;
  unsigned long long* list = ws.list + (size_t)p * kmax;
  const float* row = pr.row[p];
  if (k <= 0) return;
  __shared__ int s_wave[4], s_res[2];
  int digit, need_eq;
  topk_pick(ws.hist + ((size_t)p * 3 + 2) * kBins, bins_of(2), (int)st[7], s_wave, s_res, digit, need_eq);
  const unsigned T = st[5] | ((unsigned)digit << shift_of(2));   // the exact 32-bit threshold key
  const int lane = tid & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  // The selected elements of this chunk are collected in LDS (one LDS atomic per wave and list) and appended with
  // ONE global atomic per workgroup and list; reserving global slots per wave and iteration put a returning
  // device atomic (~1.5 us) into about half of the iterations (32 us for the P2 row).  Which of several elements
  // EQUAL to the threshold make the cut depends on arrival order, as before.
  constexpr int kLg = 512, kLe = 256;
  __shared__ unsigned long long s_g[kLg], s_e[kLe];
  __shared__ int s_n[2], s_base[2];
  if (tid < 2) s_n[tid] = 0;
  __syncthreads();
  // (wave-uniform trip count)
  for (int i0 = lo; i0 < hi; i0 += 256) {
    const int i = i0 + tid;
    const bool in = i < hi;
    const unsigned key = in ? key_of(topk_elem(pr, p, row, i)) : 0u;
    const unsigned long long comp = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)i);
    const bool gt = in && key > T, eq = in && key == T;
    const unsigned long long mg = __ballot(gt), me = __ballot(eq);
    if (mg) {
      const int leader = __ffsll((long long)mg) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&s_n[0], __popcll(mg));
      base = __shfl(base, leader, 64);
      const int slot = base + __popcll(mg & below);
      if (gt) {
        if (slot < kLg) s_g[slot] = comp;
        else list[atomicAdd(&st[3], 1u)] = comp;           // LDS list full: straight to the list ([0, k - need_eq))
      }
    }
    if (me) {
      const int leader = __ffsll((long long)me) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&s_n[1], __popcll(me));
      base = __shfl(base, leader, 64);
      const int slot = base + __popcll(me & below);
      if (eq) {
        if (slot < kLe) s_e[slot] = comp;
        else {
          const int g = (int)atomicAdd(&st[4], 1u);
          if (g < need_eq) list[k - 1 - g] = comp;         // ties fill from the back
        }
      }
    }
  }
  __syncthreads();
  const int ng = min(s_n[0], kLg), ne = min(s_n[1], kLe);
  if (tid == 0) {
    s_base[0] = ng ? (int)atomicAdd(&st[3], (unsigned)ng) : 0;
    s_base[1] = ne ? (int)atomicAdd(&st[4], (unsigned)ne) : 0;
  }
  __syncthreads();
  for (int i = tid; i < ng; i += 256) list[s_base[0] + i] = s_g[i];
  for (int i = tid; i < ne; i += 256) {
    const int g = s_base[1] + i;
    if (g < need_eq) list[k - 1 - g] = s_e[i];
  }
}

__global__ __launch_bounds__(1024) void topk_sort_kernel(TopkProblems pr, TopkWs ws, int kmax,
                                                         float* __restrict__ out_val,
                                                         long long* __restrict__ out_idx) {
  __shared__ unsigned long long buf[kMaxK];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int k = min(pr.k[p], pr.len[p]);
  int m = 1;
  while (m < k) m <<= 1;
  const unsigned long long* list = ws.list + (size_t)p * kmax;
  for (int i = tid; i < m; i += 1024) buf[i] = i < k ? list[i] : 0ull;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int j = tid; j < (m >> 1); j += 1024) {
        const int lo = (j / stride) * (stride << 1) + (j % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);              // descending blocks first: final = desc
        const unsigned long long a = buf[lo], b = buf[hi];
        if ((a < b) == desc) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
      bgs::bitonic_stage_sync(size, stride);               // wave-scope for stride <= 64 (bgs_common.h)
    }
  }
  __syncthreads();                                         // (sorts of < 128 elements end on a wave-scope stage)
  float* ov = out_val + (size_t)p * kmax;
  long long* oi = out_idx + (size_t)p * kmax;
  for (int i = tid; i < kmax; i += 1024) {
    if (i < k) {
      const unsigned long long c = buf[i];
      ov[i] = value_of((unsigned)(c >> 32));
      oi[i] = (long long)(0xffffffffu - (unsigned)(c & 0xffffffffu));
    } else {
      ov[i] = 0.f;
      oi[i] = 0;
    }
  }
}

}  // namespace

extern "C" size_t bgs_topk_workspace_bytes(int P, int kmax) {
  if (P <= 0 || kmax <= 0) return 0;
  return (size_t)P * (3 * kBins * sizeof(int) + kState * sizeof(unsigned) +
                      (size_t)kmax * sizeof(unsigned long long));
}

extern "C" int bgs_topk_sorted_f32(const float* const* host_rows, const int* host_len,
                                   const int* host_k, const int* host_inner,
                                   const int* host_pitch, int P, int kmax, float* out_val,
                                   long long* out_idx, void* workspace, bgs_stream_t stream) {
  if (P < 0 || kmax <= 0) return BGS_ERR_INVALID_ARG;
  if (P == 0) return BGS_OK;
  if (!host_rows || !host_len || !host_k || !out_val || !out_idx || !workspace)
    return BGS_ERR_INVALID_ARG;
  if (P > kMaxProblems || kmax > kMaxK) return BGS_ERR_UNSUPPORTED;
  TopkProblems pr;
  int maxlen = 0;
  for (int p = 0; p < kMaxProblems; ++p) {
    pr.row[p] = nullptr;
    pr.len[p] = pr.k[p] = pr.inner[p] = pr.pitch[p] = 0;
  }
  if ((host_inner == nullptr) != (host_pitch == nullptr)) return BGS_ERR_INVALID_ARG;
  for (int p = 0; p < P; ++p) {
    if (host_len[p] < 0 || host_k[p] < 0 || host_k[p] > kmax) return BGS_ERR_INVALID_ARG;
    if (host_len[p] > 0 && !host_rows[p]) return BGS_ERR_INVALID_ARG;
    pr.row[p] = host_rows[p];
    pr.len[p] = host_len[p];
    pr.k[p] = host_k[p];
    if (host_inner) {
      if (host_inner[p] < 0 || (host_inner[p] > 0 && host_pitch[p] < host_inner[p]))
        return BGS_ERR_INVALID_ARG;
      pr.inner[p] = host_inner[p];
      pr.pitch[p] = host_pitch[p];
    }
    if (host_len[p] > maxlen) maxlen = host_len[p];
  }
  TopkWs ws;
  ws.hist = reinterpret_cast<int*>(workspace);
  ws.state = reinterpret_cast<unsigned*>(ws.hist + (size_t)P * 3 * kBins);
  ws.list = reinterpret_cast<unsigned long long*>(ws.state + (size_t)P * kState);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, (size_t)P * (3 * kBins * sizeof(int) + kState * sizeof(unsigned)),
                     st) != hipSuccess)
    return BGS_ERR_LAUNCH;
  const int G = maxlen > 0 ? (maxlen + kChunk - 1) / kChunk : 1;
  for (int pass = 0; pass < 3; ++pass)
    hipLaunchKernelGGL(topk_hist_kernel, dim3(G, P), dim3(256), 0, st, pr, ws, pass);
  hipLaunchKernelGGL(topk_collect_kernel, dim3(G, P), dim3(256), 0, st, pr, ws, kmax);
  hipLaunchKernelGGL(topk_sort_kernel, dim3(P), dim3(1024), 0, st, pr, ws, kmax, out_val, out_idx);
  BGS_RETURN_LAUNCH_STATUS();
}
