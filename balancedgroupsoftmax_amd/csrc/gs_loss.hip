// Fused Balanced-Group-Softmax loss forward + backward for gfx950 (MI355X).
//
// Replaces, in one pass over the [N, W] logits, the reference's per-bin Python loop
//   GSBBoxHeadWith0.loss        mmdet/models/bbox_heads/gs_bbox_head_with0.py:160-171
//   -> _slice_preds             :134-145   (narrow(1, start, len) views)
//   -> CrossEntropyLoss.forward mmdet/models/losses/cross_entropy_loss.py:86-103
//   -> cross_entropy            :9-19      (F.cross_entropy(reduction='none'))
//   -> weight_reduce_loss       mmdet/models/losses/utils.py:26-52  (sum()/avg_factor)
// and the autograd backward of the above (B x {log_softmax, nll, mul, sum, div} kernels
// forward + the same backward in the reference).
//
// Mapping (HBM-bound streaming op, 9.9 KB algorithmic traffic per RoI): see gs_rowwave.h —
//   one 4-wave workgroup per RoI row (grid-stride over rows); the row is read ONCE (16-byte
//   coalesced loads -> LDS) and its gradient row written ONCE (LDS -> 16-byte coalesced
//   stores; no atomics: every column has one owner bin); each bin is swept by one wave with
//   bin-aligned lane indexing so that max / sum / target / weight are wave-uniform scalars and
//   the reductions are DPP wave all-reduces; the bins of a row run concurrently on the 4 waves.
//   * bin geometry arrives by value in the kernarg segment and the bin labels were gathered
//     by the prepare kernel: the only memory round trip before the math is the row load;
//   * (row, bin) pairs with zero sample weight skip the softmax entirely (wave-uniform);
//   * the loss term of a (row, bin) needs no extra global load (target logit read from LDS).
// Per-workgroup partial losses are reduced in a fixed order by a second tiny kernel
// (bitwise reproducible, no atomics).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "bgs_common.h"
#include "gs_rowwave.h"

namespace {

constexpr int kBlock = 256;   // generic fallback + helper kernels
constexpr int kWaves = kBlock / BGS_WAVE;
constexpr int kMaxGrid = 2048;
// workspace = [BGS_MAX_BINS][<= kMaxGrid] partial sums, then one int: the grid that wrote them (the main kernels
// record it, bgs_gs_loss_reduce reads it — the row-per-wave kernel's grid is not a function of N alone)
constexpr int kGridSlot = kMaxGrid * BGS_MAX_BINS;
int g_rowwave_pf = 5;        // bgs_gs_loss_tuning: 5 (default) = row-per-wave kernel for 4096 < N < 12288 rows, mode 3 elsewhere | 6 / 7 = row-per-wave
                             // for every N >= the row threshold with plain / non-temporal row loads | 0 = no next-row prefetch | 1 = prefetch | 2 / 3 / 4 = prefetch + non-temporal
                             // loads / stores / both (A/B; 3 = default: profiles/r8e_gs_rowwave_prefetch_ab.txt)

// PF (rows of at most 2 * 256 * VEC floats — W <= 2048 for the 16-byte path: every LVIS table): the NEXT row of the
// workgroup is fetched into registers while the current one is processed, so that a workgroup's rows no longer cost
// a global-memory round trip each in front of their first barrier.  NT bit 1: the gradient leaves with the
// non-temporal hint (written once, read by a later kernel from HBM anyway at these sizes); bit 0: the row loads
// carry it too (measured slower).  At N = 65,536 on one box: 4.53 TB/s (round-3 kernel) -> 5.06 (PF) -> 5.34
// (PF + nt stores) = 0.67 of the 8 TB/s HBM peak against a 6.29 TB/s copy ceiling
// (profiles/r8e_gs_rowwave_prefetch_ab.txt).  Same arithmetic, same order: bit-identical losses and gradients.
template <int VEC, bool WRITE_GRAD, bool PF = false, int NT = 0>
__global__ __launch_bounds__(kBlock) void gs_loss_rowwave_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ bin_labels,
    const float* __restrict__ weights, const float* __restrict__ avg, bgs::BinGeom geom, int N,
    int B, int W, int wpad, float* __restrict__ partial, float* __restrict__ dlogits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][wpad] double-buffered row
                                                                // (+ read slack, see launcher)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);

  // lane b < B of every wave keeps the per-bin constants / per-row scalars of bin b
  float my_inv_avg = 0.f;
  if (lane < B) my_inv_avg = 1.f / (avg ? avg[lane] : fmaxf((float)N, 1.f));
  float lacc = 0.f;  // wave (b % kWaves), lane b accumulates the loss of bin b

  // PF: this thread's (at most two) VEC-wide pieces of a row, and the row's bin label / weight of lane b
  const int c0 = tid * VEC, c1 = (tid + kBlock) * VEC;
  float pf0[VEC], pf1[VEC];
  int pf_bl = 0;
  float pf_w = 1.f;
  auto prefetch = [&](int r) {
    const float* g = logits + (size_t)r * W;
    if (NT & 1) {                        // streamed once: non-temporal
      if (c0 < W) bgs::load_vec_nt<VEC>(g + c0, pf0);
      if (c1 < W) bgs::load_vec_nt<VEC>(g + c1, pf1);
    } else {
      if (c0 < W) bgs::load_vec<VEC>(g + c0, pf0);
      if (c1 < W) bgs::load_vec<VEC>(g + c1, pf1);
    }
    if (lane < B) {
      pf_bl = bin_labels[(size_t)lane * N + r];
      pf_w = weights ? weights[(size_t)lane * N + r] : 1.f;
    }
  };
  if (PF && (int)blockIdx.x < N) prefetch(blockIdx.x);

  int par = 0;
  for (int r = blockIdx.x; r < N; r += gridDim.x, par ^= 1) {
    float* row = smem + (size_t)par * wpad;
    int my_bl = 0;
    float my_coef = 0.f;
    if (PF) {
      if (c0 < W) bgs::store_vec<VEC>(row + c0, pf0);
      if (c1 < W) bgs::store_vec<VEC>(row + c1, pf1);
      my_bl = pf_bl;
      my_coef = lane < B ? pf_w * my_inv_avg : 0.f;
      if (r + (int)gridDim.x < N) prefetch(r + gridDim.x);      // in flight under this row's sweeps
    } else {
      bgs::stage_row<VEC>(logits + (size_t)r * W, row, W, tid, kBlock);
      if (lane < B) {
        my_bl = bin_labels[(size_t)lane * N + r];
        const float w = weights ? weights[(size_t)lane * N + r] : 1.f;
        my_coef = w * my_inv_avg;
      }
    }
    __syncthreads();
    for (int b = wave; b < B; b += kWaves) {  // bins are independent: one wave each
      const int s = geom.start[b], n = geom.len[b];
      const float coef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_coef), b));
      const int tgt = min(max(__builtin_amdgcn_readlane(my_bl, b), 0), n - 1);
      float* seg = row + s;
      if (coef == 0.f) {  // wave-uniform: this (row, bin) carries no weight -> zero gradient
        if (WRITE_GRAD)
          for (int j = lane; j < n; j += BGS_WAVE) seg[j] = 0.f;
        continue;
      }
      float term;
      if (n <= BGS_WAVE * bgs::kSweep) {  // wave-uniform; true for every shipped table
        term = bgs::bin_loss_registers<WRITE_GRAD>(seg, n, lane, coef, tgt);
      } else {
        const float zt = seg[tgt];  // target logit, read before the in-place exp
        float m, S;
        bgs::bin_softmax_inplace(seg, n, lane, m, S);
        term = coef * ((m + logf(S)) - zt);
        if (WRITE_GRAD) bgs::bin_grad_inplace(seg, n, lane, coef / S, coef, tgt);
      }
      if (lane == b) lacc += term;
    }
    if (WRITE_GRAD) {
      __syncthreads();
      if (NT & 2) {
        float* g = dlogits + (size_t)r * W;
        for (int c = tid * VEC; c < W; c += kBlock * VEC) {
          float t[VEC];
          bgs::load_vec<VEC>(row + c, t);
          bgs::store_vec_nt<VEC>(g + c, t);
        }
      } else {
        bgs::unstage_row<VEC>(row, dlogits + (size_t)r * W, W, tid, kBlock);
      }
    }
  }
  // bin b lives in wave b % kWaves, lane b: no cross-wave reduction needed
  if (lane < B && (lane % kWaves) == wave)
    partial[(size_t)lane * gridDim.x + blockIdx.x] = lacc;
  if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(partial)[kGridSlot] = gridDim.x;
}

// Round 6: a row per WAVE, the LDS row private to it (gs_rowwave.h, wave_phase) — no workgroup barrier anywhere in
// the row loop.  The 4-wave-per-row kernel above parks a whole workgroup twice per row (row staged -> sweeps ->
// gradient out) and its 8 workgroups per CU keep 8 rows in flight; here 32 independent waves per CU each hold a
// row in LDS and the next one in registers, every wave streams at its own pace, and the sweeps of one wave hide
// under the loads and stores of the 31 others.  KV = 16-byte pieces of a row per lane (W <= 256 * KV).  Per-bin
// arithmetic is bin_loss_registers, unchanged: the gradient is bit-identical to the kernel above; a bin's loss is
// the same terms summed in a different order (wave w of workgroup g takes rows 4 g + w, + 4 * grid, ...; the four
// waves' sums are added in wave order at the end).
template <int KV, bool WRITE_GRAD, bool NTL = false>
__global__ __launch_bounds__(kBlock) void gs_loss_wavepriv_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ bin_labels,
    const float* __restrict__ weights, const float* __restrict__ avg, bgs::BinGeom geom, int N,
    int B, int W, int wpad, float* __restrict__ partial, float* __restrict__ dlogits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [kWaves][wpad] + read slack (launcher)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  float* row = smem + (size_t)wave * wpad;
  const int nw = gridDim.x * kWaves;
  const int nq = W >> 2;                                   // 16-byte pieces of a row (W % 4 == 0)

  float my_inv_avg = 0.f;
  if (lane < B) my_inv_avg = 1.f / (avg ? avg[lane] : fmaxf((float)N, 1.f));
  float lacc = 0.f;                                        // lane b: the loss of bin b over this wave's rows

  float pf[KV][4];
  int pf_bl = 0;
  float pf_w = 1.f;
  auto prefetch = [&](int r) {
    const float* g = logits + (size_t)r * W;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int q = lane + BGS_WAVE * k;
      if (q < nq) {
        if (NTL) bgs::load_vec_nt<4>(g + 4 * q, pf[k]);
        else bgs::load_vec<4>(g + 4 * q, pf[k]);
      }
    }
    if (lane < B) {
      pf_bl = bin_labels[(size_t)lane * N + r];
      pf_w = weights ? weights[(size_t)lane * N + r] : 1.f;
    }
  };
  int r = blockIdx.x * kWaves + wave;
  if (r < N) prefetch(r);
  for (; r < N; r += nw) {
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int q = lane + BGS_WAVE * k;
      if (q < nq) bgs::store_vec<4>(row + 4 * q, pf[k]);
    }
    const int my_bl = pf_bl;
    const float my_coef = lane < B ? pf_w * my_inv_avg : 0.f;
    if (r + nw < N) prefetch(r + nw);                      // in flight under this row's sweeps
    bgs::wave_phase();
    for (int b = 0; b < B; ++b) {
      const int s = geom.start[b], n = geom.len[b];
      const float coef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_coef), b));
      const int tgt = min(max(__builtin_amdgcn_readlane(my_bl, b), 0), n - 1);
      float* seg = row + s;
      if (coef == 0.f) {                                   // wave-uniform: no weight -> zero gradient
        if (WRITE_GRAD)
          for (int j = lane; j < n; j += BGS_WAVE) seg[j] = 0.f;
        continue;
      }
      const float term = bgs::bin_loss_registers<WRITE_GRAD>(seg, n, lane, coef, tgt);
      if (lane == b) lacc += term;
    }
    if (WRITE_GRAD) {
      bgs::wave_phase();
      float* g = dlogits + (size_t)r * W;
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const int q = lane + BGS_WAVE * k;
        if (q < nq) {
          float t[4];
          bgs::load_vec<4>(row + 4 * q, t);
          bgs::store_vec_nt<4>(g + 4 * q, t);
        }
      }
      bgs::wave_phase();
    }
  }
  // the four waves' sums of a bin, added in wave order (fixed: bitwise reproducible)
  __syncthreads();
  if (lane < B) smem[wave * BGS_MAX_BINS + lane] = lacc;
  __syncthreads();
  if (tid < B) {
    float s = smem[tid];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += smem[w * BGS_MAX_BINS + tid];
    partial[(size_t)tid * gridDim.x + blockIdx.x] = s;
  }
  if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(partial)[kGridSlot] = gridDim.x;
}

__device__ __forceinline__ float block_max(float v, float* sm) {
  v = bgs::wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < kWaves; ++w) r = fmaxf(r, sm[w]);
  return r;
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = bgs::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < kWaves; ++w) r += sm[w];
  return r;
}

// Any width / alignment / bin count <= BGS_MAX_BINS: one 256-thread block per row, the row is
// re-read from L1/L2 for the max, sum and gradient sweeps.  Also serves as an independent
// cross-check of the register-resident kernel in the GPU tests.
template <bool WRITE_GRAD>
__global__ __launch_bounds__(kBlock) void gs_loss_generic_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ bin_labels,
    const float* __restrict__ weights, const float* __restrict__ avg, bgs::BinGeom geom, int N,
    int B, int W, float* __restrict__ partial, float* __restrict__ dlogits) {
  __shared__ float sm[kWaves];
  __shared__ float acc[BGS_MAX_BINS];
  const int tid = threadIdx.x;
  if (tid < BGS_MAX_BINS) acc[tid] = 0.f;
  __syncthreads();
  for (int r = blockIdx.x; r < N; r += gridDim.x) {
    const float* zr = logits + (size_t)r * W;
    float* gr = WRITE_GRAD ? dlogits + (size_t)r * W : nullptr;
    if (WRITE_GRAD) {
      for (int j = tid; j < W; j += kBlock) gr[j] = 0.f;
      __syncthreads();
    }
    for (int b = 0; b < B; ++b) {
      const int s = geom.start[b], n = geom.len[b];  // validated on the host
      if (n == 0) continue;
      const float a = avg ? avg[b] : fmaxf((float)N, 1.f);
      const float w = weights ? weights[(size_t)b * N + r] : 1.f;
      const float coef = w * (1.f / a);
      if (coef == 0.f) continue;  // block-uniform
      const int bl = min(max(bin_labels[(size_t)b * N + r], 0), n - 1);
      float pm = -INFINITY;
      for (int j = tid; j < n; j += kBlock) pm = fmaxf(pm, zr[s + j]);
      const float m = block_max(pm, sm);
      float ps = 0.f;
      for (int j = tid; j < n; j += kBlock) ps += __expf(zr[s + j] - m);
      const float S = block_sum(ps, sm);
      if (tid == 0) acc[b] += coef * ((m + logf(S)) - zr[s + bl]);
      if (WRITE_GRAD) {
        const float invS = 1.f / S;
        for (int j = tid; j < n; j += kBlock)
          gr[s + j] = coef * (__expf(zr[s + j] - m) * invS - (j == bl ? 1.f : 0.f));
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid < B) partial[(size_t)tid * gridDim.x + blockIdx.x] = acc[tid];
  if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(partial)[kGridSlot] = gridDim.x;
}

// loss[b] = sum_g partial[b, g] in a fixed order.  1024 threads: all loads of a bin are issued
// at once (one memory round trip per bin, the bins' loads overlap), then wave + LDS reduction.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial,
                                                               int G, int B,
                                                               float* __restrict__ out,
                                                               float scale) {
  __shared__ float sm[BGS_MAX_BINS][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (G < 0) G = reinterpret_cast<const int*>(partial)[kGridSlot];     // recorded by the kernel that wrote the partials
  float acc[BGS_MAX_BINS];
#pragma unroll
  for (int b = 0; b < BGS_MAX_BINS; ++b) {
    acc[b] = 0.f;
    if (b < B) {
      for (int g = tid; g < G; g += 1024) acc[b] += partial[(size_t)b * G + g];
    }
  }
#pragma unroll
  for (int b = 0; b < BGS_MAX_BINS; ++b) {
    if (b < B) {
      const float s = bgs::wave_sum(acc[b]);
      if (lane == 0) sm[b][wave] = s;
    }
  }
  __syncthreads();
  if (tid < B) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sm[tid][w];
    out[tid] = s * scale;
  }
}

// dlogits[:, bin b] *= g[b]; early-out when every g[b] == 1 (the plain Faster R-CNN case).
__global__ __launch_bounds__(kBlock) void gs_scale_grad_kernel(float* __restrict__ dlogits,
                                                               bgs::BinGeom geom,
                                                               const float* __restrict__ g, int N,
                                                               int B, int W) {
  bool all_one = true;
  for (int b = 0; b < B; ++b) all_one = all_one && (g[b] == 1.f);
  if (all_one) return;
  extern __shared__ __attribute__((aligned(16))) float scale[];
  for (int c = threadIdx.x; c < W; c += kBlock) {
    float sc = 0.f;
    for (int b = 0; b < B; ++b) {
      if (c >= geom.start[b] && c < geom.start[b] + geom.len[b]) sc = g[b];
    }
    scale[c] = sc;
  }
  __syncthreads();
  const size_t total = (size_t)N * W;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % (size_t)W);
    dlogits[i] *= scale[c];
  }
}

// ---------------------------------------------------------------------------------------------
// Fused head kernel (second version): GSBBoxHeadWith0.loss() in ONE streaming launch —
//   _remap_labels + _sample_others (gs_bbox_head_with0.py:63-112), the per-bin losses and their
//   gradient (:160-171), and the box branch (:173-185: SmoothL1 on the positive rows' own class,
//   optionally its dense [N, 4R] gradient) — for N <= 4096 rows and no per-class reweighting.
// What the first version got wrong (profiles/r4h: 20 us per 1024-row launch, 4x the plain loss
// kernel): every one of the N workgroups hashed N 64-bit keys per sampled bin with all four waves
// and exchanged the rank through LDS + a barrier per bin.  Now:
//   * prologue, once per workgroup: the 16-bit flag word {real, foreground in bin b} of EVERY row
//     (8 B of label + B table gathers per row, all loads of a thread's rows in flight together)
//     and the per-bin foreground counts by wave ballots — no LDS atomics;
//   * the "others" decision of the workgroup's own row in bin b is taken by THE WAVE THAT OWNS
//     BIN b (bins are independent, one wave each): the row's position among the bin's candidates
//     (a count over the flag words below the row, one DPP wave sum) goes through the bin's keyed
//     pseudo-random permutation of [0, n_bg) (bgs::gs_perm, wave-uniform: scalar ALU) — drawn iff
//     the image is < k_b: exact-k sampling without a key per row, no cross-wave exchange, so a row
//     costs the same two barriers as the plain loss kernel;
//   * closed forms as before: k_b = int(n_fg * ratio), avg_b = n_real | n_fg + k_b | 1;
//   * the per-bin loss weights ride in the kernel arguments (coef = w / avg * loss_weight);
//   * the box branch of the row is four lanes of the last wave.
// Results are bitwise those of bgs_gs_prepare + bgs_gs_loss_fwd_bwd (+ bgs_bbox_smooth_l1_fwd_bwd)
// (tests/test_gpu_gs.py).
constexpr int kFusedMaxN = 4096;
constexpr int kFusedRowsPerPass = 4;           // rows of the prologue a thread keeps in flight
__device__ const float g_one = 1.0f;           // "no row_weights": every row reads this 1

struct GsHeadArgs {
  const float* logits;
  const int64_t* labels;
  const int64_t* l2b;
  const uint16_t* class_bits;    // [C] bit b: class is foreground in bin b (l2b[b][c] > 0), or null
  const float* row_weights;
  bgs::BinGeom geom;
  float lw[BGS_MAX_BINS];        // per-bin loss weight (CrossEntropyLoss.loss_weight)
  int N, C, B, W, wpad;
  double ratio;
  uint64_t seed;
  const uint64_t* seed_offset;
  float* partial;                // [B + 1][gridDim.x]: per-bin loss partials, then the box partials
  float* dlogits;
  float* avg_out;
  int32_t* bl_out;
  float* w_out;
  // box branch (bbox_pred == nullptr: none)
  const float* bbox_pred;
  const float* bbox_targets;
  const float* bbox_weights;
  float* dbbox;                  // dense [N, 4R] gradient or nullptr
  int R;
  float beta, box_w;
  unsigned long long* tstamps;   // debug (bgs_gs_head_debug_timestamps): [gridDim.x][8] s_memtime marks, or null
  // round 6: the reduction of the partials INSIDE the main launch, by the workgroup that arrives last (null: the separate
  // gs_head_reduce_kernel launch).  ticket: a zeroed device word, left zero by the last workgroup.
  unsigned* ticket;
  float* fold_out;               // [B + 1] loss terms
  float* fold_total;             // [1] or null
  uint64_t* fold_counter;        // the draw counter to advance, or null
};

// v_writelane_b32 with a run-time lane (hipcc has no builtin for it).  gfx9 VALU instructions read ONE SGPR
// over the constant bus; the lane select of v_writelane is exempt when it is M0.  (M0 is a reserved register
// to the compiler, hence the diagnostic; nothing else in this file's kernels uses it — no LDS-DMA, no s_movrel,
// no s_sendmsg — which the disassembly confirms.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ int gs_writelane(int val, int lane, int old) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0"
               : "+v"(old)
               : "s"(val), "s"(lane)
               : "m0");
  return old;
}
#pragma clang diagnostic pop

__device__ __forceinline__ float gs_sl1(float d, float beta, float& grad) {
  const float ad = fabsf(d);
  if (ad < beta) {
    grad = d / beta;
    return 0.5f * ad * ad / beta;
  }
  grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return ad - 0.5f * beta;
}

// LDS layout (dynamic): [2][wpad] rows + read slack | flags u16 [N rounded to 64] (VAR 1: 16 bit planes of
// N / 64 64-bit words, the same bytes) | class bits u16
// [C rounded to 8] | counts int [kWaves][MAX_BINS + 1] | box gradient float [4]
__host__ __device__ inline size_t gs_head_lds_bytes(int N, int C, int wpad) {
  return sizeof(float) * (2 * (size_t)wpad + BGS_WAVE * bgs::kSweep) + 2 * (size_t)((N + 63) & ~63) +
         2 * (size_t)((C + 7) & ~7) + sizeof(int) * kWaves * (BGS_MAX_BINS + 1) + sizeof(float) * 4;
}

// VAR 1 (round 3, `bgs_gs_head_variant`): BIT PLANES instead of per-row flag words.  The prologue turns the
// flag words of a wave's 64 rows into one 64-bit ballot per plane (plane b = "real and foreground in bin b",
// plane 15 = "real") and lane p of the wave stores plane p's word: sh_pl[p][row / 64].  Then the number of
// real rows is one popcount pass over plane 15 (lane w = word w), and for a bin b both its foreground count
// and the row's position among the bin's candidates come out of ONE pass over plane b + ONE wave sum
// (popc(F) << 16 | popc((R ^ F) & below-the-row mask)) in the wave that owns the bin — instead of four packed
// counter registers with a wave sum each, an LDS exchange of the per-wave counts, and a scan over the flag words
// below the row.  Same decisions, same arithmetic behind them: bitwise the results of VAR 0.
template <int VEC, bool WRITE_GRAD, bool BOX, int VAR>
__global__ __launch_bounds__(kBlock) void gs_head_fused_kernel(GsHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, C = a.C, B = a.B, W = a.W, wpad = a.wpad;
  unsigned short* sh_flags = reinterpret_cast<unsigned short*>(smem + 2 * (size_t)wpad + BGS_WAVE * bgs::kSweep);
  unsigned long long* sh_pl = reinterpret_cast<unsigned long long*>(sh_flags);   // VAR 1: [16][NW]
  const int NW = (N + 63) >> 6;
  unsigned short* sh_cbits = sh_flags + ((N + 63) & ~63);
  int* sh_cntw = reinterpret_cast<int*>(sh_cbits + ((C + 7) & ~7));
  float* sh_box = reinterpret_cast<float*>(sh_cntw + kWaves * (BGS_MAX_BINS + 1));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  uint64_t seed = a.seed;
  if (a.seed_offset) seed += 0x2545F4914F6CDD1Dull * a.seed_offset[0];   // device-side draw counter
#define GS_MARK(i_)                                                                     \
  do {                                                                                  \
    if (a.tstamps && tid == 0) a.tstamps[(size_t)blockIdx.x * 8 + (i_)] = clock64();    \
  } while (0)
  GS_MARK(0);

  // ---- prologue.  The flag word of a row is a function of its label alone: {foreground in bin b}
  //      = class_bits[label] — a C-entry 16-bit table (2.4 KB for LVIS) that every workgroup copies
  //      into LDS with a handful of coalesced loads, instead of B scattered 8-byte gathers into the
  //      [B, C] int64 table per row (N x B cache lines per workgroup: what the first version of this
  //      kernel spent its time on).  Without a prebuilt table (class_bits == null) the workgroup
  //      derives it from label2binlabel by a coalesced sweep.  All loads are unconditional
  //      (addresses are selected, not loads: a branch around a load makes hipcc wait for it where
  //      it is issued); the labels, the table and the workgroup's first logits row are in flight
  //      together.  Lane b <= B of every wave counts bin b over the wave's share of the rows by
  //      ballots (b == B: real rows) — no LDS atomics.
  int mycnt = 0;
  // the workgroup's first row: its label is the OLDEST load of the kernel; the dependent gather of its bin
  // labels is issued below, behind every independent load (the label is wave-uniform, so its first use is a
  // v_readfirstlane behind an s_waitcnt: taken here, that wait was a full memory round trip in front of all
  // the other loads — round 3, from the ISA)
  const int64_t yraw_first = a.labels[blockIdx.x];
  const float* rw_base = a.row_weights ? a.row_weights : &g_one;
  const int rw_step = a.row_weights ? 1 : 0;
  constexpr int RP = kFusedRowsPerPass;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int cpad = (C + 7) & ~7;
  // (no table: the address of the labels stands in — the value is not used)
  const uint16_t* cb_src = a.class_bits ? a.class_bits + (tid * 8 < cpad ? tid * 8 : 0)
                                        : reinterpret_cast<const uint16_t*>(a.labels);
  const u32x4_t cb0 = *reinterpret_cast<const u32x4_t*>(cb_src);
  int64_t y0[RP];
  float rwv0[RP];
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const int r = tid + kBlock * i;
    const int rc = r < N ? r : 0;
    y0[i] = a.labels[rc];
    rwv0[i] = rw_base[(size_t)rc * rw_step];
  }
  float t0[VEC], t1[VEC];
  const int c0 = tid * VEC, c1 = c0 + kBlock * VEC;
  {   // (W >= 2 * VEC * kBlock is not required: out-of-range lanes re-read column 0)
    const float* g = a.logits + (size_t)blockIdx.x * W;
    bgs::load_vec<VEC>(g + (c0 < W ? c0 : 0), t0);
    bgs::load_vec<VEC>(g + (c1 < W ? c1 : 0), t1);
  }
  int my_bl_first = 0;
  if (lane < B) {     // counted wait: only the label load has to have landed
    const int64_t yc = yraw_first < 0 ? 0 : (yraw_first >= C ? (int64_t)C - 1 : yraw_first);
    my_bl_first = (int)a.l2b[(size_t)lane * C + yc];
  }
  if (a.class_bits) {
    // 8 table entries (16 B) per thread and pass; the table is padded to a multiple of 8 entries and
    // 16-byte aligned (bgs_gs_class_bin_mask): the first pass was loaded at the top of the kernel
    *reinterpret_cast<u32x4_t*>(sh_cbits + (tid * 8 < cpad ? tid * 8 : 0)) = cb0;
    for (int c = (tid + kBlock) * 8; c < cpad; c += kBlock * 8)
      *reinterpret_cast<u32x4_t*>(sh_cbits + c) = *reinterpret_cast<const u32x4_t*>(a.class_bits + c);
  } else {
    for (int c = tid; c < C; c += kBlock) {
      unsigned bits = 0u;
      for (int b = 0; b < B; ++b)
        if (a.l2b[(size_t)b * C + c] > 0) bits |= 1u << b;
      sh_cbits[c] = (unsigned short)bits;
    }
  }
  if (c0 < W) bgs::store_vec<VEC>(smem + c0, t0);
  if (c1 < W) bgs::store_vec<VEC>(smem + c1, t1);
  for (int c = c1 + kBlock * VEC; c < W; c += kBlock * VEC) {   // rows wider than 2 x 256 x VEC
    float t[VEC];
    bgs::load_vec<VEC>(a.logits + (size_t)blockIdx.x * W + c, t);
    bgs::store_vec<VEC>(smem + c, t);
  }
  GS_MARK(1);                             // loads landed, LDS written
  __syncthreads();                        // class bits (and the first row) are in LDS
  GS_MARK(2);
  unsigned pk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  // one pass = RP rows per thread (labels + row weights in registers); the first pass uses the loads of the top
  // of the kernel and is straight-line code, batches of N > kBlock * RP rows loop
  auto pass = [&](const int base, const int64_t (&yv)[RP], const float (&rwv)[RP]) {
    unsigned bits[RP];
    // the RP table lookups are issued together, their consumers sit behind ONE wait (read -> wait -> write per
    // row was RP serial LDS round trips)
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int64_t y = yv[i] < 0 ? 0 : (yv[i] >= C ? (int64_t)C - 1 : yv[i]);
      bits[i] = sh_cbits[(int)y];
    }
    if constexpr (VAR == 1) {
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        const int r = base + tid + kBlock * i;
        const unsigned wbits = (r < N && rwv[i] > 0.f) ? (bits[i] | 0x8000u) : 0u;
        int lo = 0, hi = 0;                      // lane p: plane p's word of this wave's 64 rows
        for (int p = 1; p < B; ++p) {
          const unsigned long long m = __ballot((wbits >> p) & 1u);
          lo = gs_writelane((int)(unsigned)m, p, lo);
          hi = gs_writelane((int)(unsigned)(m >> 32), p, hi);
        }
        const unsigned long long mr = __ballot(wbits >> 15);
        lo = gs_writelane((int)(unsigned)mr, 15, lo);
        hi = gs_writelane((int)(unsigned)(mr >> 32), 15, hi);
        const int w = ((base + kBlock * i) >> 6) + wave;
        if (lane < 16 && w < NW)
          sh_pl[lane * NW + w] = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
      }
    } else {
      bool real[RP];
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        const int r = base + tid + kBlock * i;
        real[i] = r < N && rwv[i] > 0.f;
        if (r < N) sh_flags[r] = (unsigned short)(real[i] ? (bits[i] | 0x8000u) : 0u);
      }
      // per-thread counters, two 16-bit fields per register: bins 2 f and 2 f + 1 (a wave's share of
      // the rows is <= 64 * 16: no carry), field 15 = real rows
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        const unsigned wbits = real[i] ? (bits[i] | 0x8000u) : 0u;
#pragma unroll
        for (int f = 0; f < 8; ++f)
          if (2 * f < B || f == 7) pk[f] += ((wbits >> (2 * f)) & 1u) | (((wbits >> (2 * f + 1)) & 1u) << 16);
      }
    }
  };
  pass(0, y0, rwv0);
  for (int base = kBlock * RP; base < N; base += kBlock * RP) {
    int64_t yv[RP];
    float rwv[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int r = base + tid + kBlock * i;
      const int rc = r < N ? r : 0;
      yv[i] = a.labels[rc];
      rwv[i] = rw_base[(size_t)rc * rw_step];
    }
    pass(base, yv, rwv);
  }
  // one DPP sum per used register; lane b <= B then picks its field (b == B: the real rows, field 15)
  if constexpr (VAR == 0) {
    const int fsel = lane == B ? 15 : lane;
    unsigned mine = 0u;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      if (2 * f < B || f == 7) {
        const unsigned t = (unsigned)bgs::wave_sum_i_fast((int)pk[f]);
        if ((fsel >> 1) == f) mine = t;
      }
    }
    mycnt = (int)((mine >> (16 * (fsel & 1))) & 0xffffu);
  }
  if (VAR == 0 && lane <= B) sh_cntw[wave * (BGS_MAX_BINS + 1) + lane] = mycnt;
  GS_MARK(3);                             // flags + ballots done
  __syncthreads();                        // flags, counts and the first row are in LDS
  GS_MARK(4);

  // lane b < B of every wave: constants of bin b (gs_prepare_kernel's mode / k / avg)
  int my_mode = 0, my_k = 0, my_nbg = 0;  // 0 = all zero, 1 = all one, 2 = sampled
  float my_scale = 0.f;                   // loss_weight / avg
  int n_real = 0;
  unsigned long long Rw = 0ull;             // VAR 1: lane w holds word w of the "real" plane
  if constexpr (VAR == 1) {
    Rw = lane < NW ? sh_pl[15 * NW + lane] : 0ull;
    n_real = bgs::wave_sum_i_fast(__popcll(Rw));
  } else {
#pragma unroll
    for (int v = 0; v < kWaves; ++v) n_real += sh_cntw[v * (BGS_MAX_BINS + 1) + B];
  }
  if (VAR == 0 && lane < B) {
    int n_fg = 0;
#pragma unroll
    for (int v = 0; v < kWaves; ++v) n_fg += sh_cntw[v * (BGS_MAX_BINS + 1) + lane];
    const int n_bg = n_real - n_fg;
    my_nbg = n_bg;
    float total;
    if (lane == 0) {
      my_mode = 1;
      total = (float)n_real;
    } else if (n_fg == 0) {
      my_mode = 0;
      total = 0.f;
    } else {
      my_k = (int)((double)n_fg * a.ratio);
      my_mode = (my_k >= n_bg) ? 1 : 2;
      total = my_mode == 1 ? (float)n_real : (float)(n_fg + my_k);
    }
    const float av = fmaxf(total, 1.f);
    my_scale = (1.f / av) * a.lw[lane];
    if (blockIdx.x == 0 && wave == 0 && a.avg_out) a.avg_out[lane] = av;
  }
  const float box_scale = a.box_w / fmaxf((float)n_real, 1.f);
  float lacc = 0.f, box_acc = 0.f;

  int par = 0;
  for (int r = blockIdx.x; r < N; r += gridDim.x, par ^= 1) {
    float* row = smem + (size_t)par * wpad;
    const bool first = r == (int)blockIdx.x;
    if (!first) bgs::stage_row<VEC>(a.logits + (size_t)r * W, row, W, tid, kBlock);
    bool real_r;
    if constexpr (VAR == 1) {
      const int wr = bgs::uniform(r >> 6);
      const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)Rw, wr);
      const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(Rw >> 32), wr);
      real_r = (((r & 32) ? rhi : rlo) >> (r & 31)) & 1u;
    } else {
      real_r = (sh_flags[r] & 0x8000u) != 0u;
    }
    int64_t yraw = yraw_first;
    int my_bl = my_bl_first;
    if (!first) {
      yraw = a.labels[r];
      const int64_t yr = yraw < 0 ? 0 : (yraw >= C ? (int64_t)C - 1 : yraw);
      my_bl = 0;
      if (lane < B) my_bl = (int)a.l2b[(size_t)lane * C + yr];
      __syncthreads();                    // row staged (the previous row left through the other buffer)
    }
    for (int b = wave; b < B; b += kWaves) {  // bins are independent: one wave each
      const int s = a.geom.start[b], n = a.geom.len[b];
      int mode_b;
      const int bl_b = __builtin_amdgcn_readlane(my_bl, b);
      float w = 0.f, scale_b;
      if constexpr (VAR == 1) {
        // one pass over the bin's plane: foreground count and the row's candidate position together
        int k_b = 0, nbg_b = 0;
        unsigned pos = 0u;
        float total;
        if (b == 0) {
          mode_b = 1;
          total = (float)n_real;
        } else {
          const unsigned long long Fw = lane < NW ? sh_pl[b * NW + lane] : 0ull;
          const int wr = r >> 6;
          const unsigned long long below = (1ull << (r & 63)) - 1ull;
          const unsigned long long msk = lane < wr ? ~0ull : (lane == wr ? below : 0ull);
          const unsigned tot =
              (unsigned)bgs::wave_sum_i_fast((__popcll(Fw) << 16) | __popcll((Rw ^ Fw) & msk));
          const int n_fg = (int)(tot >> 16);
          pos = tot & 0xffffu;
          nbg_b = n_real - n_fg;
          if (n_fg == 0) {
            mode_b = 0;
            total = 0.f;
          } else {
            k_b = (int)((double)n_fg * a.ratio);
            mode_b = (k_b >= nbg_b) ? 1 : 2;
            total = mode_b == 1 ? (float)n_real : (float)(n_fg + k_b);
          }
        }
        const float av = fmaxf(total, 1.f);
        scale_b = (1.f / av) * a.lw[b];
        if (first && blockIdx.x == 0 && lane == 0 && a.avg_out) a.avg_out[b] = av;
        if (real_r) {
          if (mode_b == 1 || (mode_b == 2 && bl_b > 0)) {
            w = 1.f;
          } else if (mode_b == 2) {
            const unsigned salt = bgs::gs_bin_salt(seed, (uint32_t)b);
            w = (k_b > 0 && bgs::gs_perm(salt, pos, (unsigned)nbg_b) < (unsigned)k_b) ? 1.f : 0.f;
          }
        }
      } else {
        mode_b = __builtin_amdgcn_readlane(my_mode, b);
        scale_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_scale), b));
        if (real_r) {
          if (mode_b == 1 || (mode_b == 2 && bl_b > 0)) {
            w = 1.f;
          } else if (mode_b == 2) {
            // position of the row among the bin's candidates (real, non-foreground rows) in row
            // order, then the bin's keyed permutation of [0, n_bg): drawn iff the image is < k_b
            const int k_b = __builtin_amdgcn_readlane(my_k, b);
            const int nbg_b = __builtin_amdgcn_readlane(my_nbg, b);
            int cnt = 0;
            for (int q0 = lane; q0 < r; q0 += BGS_WAVE * 4) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int q = q0 + BGS_WAVE * u;
                const unsigned f = sh_flags[q < r ? q : 0];
                cnt += (q < r && (f & 0x8000u) && !((f >> b) & 1u)) ? 1 : 0;
              }
            }
            const unsigned pos = (unsigned)bgs::wave_sum_i_fast(cnt);
            const unsigned salt = bgs::gs_bin_salt(seed, (uint32_t)b);
            w = (k_b > 0 && bgs::gs_perm(salt, pos, (unsigned)nbg_b) < (unsigned)k_b) ? 1.f : 0.f;
          }
        }
      }
      const float coef = w * scale_b;
      if (lane == 0) {
        if (a.bl_out) a.bl_out[(size_t)b * N + r] = bl_b;
        if (a.w_out) a.w_out[(size_t)b * N + r] = w;
      }
      const int tgt = min(max(bl_b, 0), n - 1);
      float* seg = row + s;
      if (coef == 0.f) {
        if (WRITE_GRAD)
          for (int j = lane; j < n; j += BGS_WAVE) seg[j] = 0.f;
        continue;
      }
      float term;
      if (n <= BGS_WAVE * bgs::kSweep) {
        term = bgs::bin_loss_registers<WRITE_GRAD>(seg, n, lane, coef, tgt);
      } else {
        const float zt = seg[tgt];
        float m, S;
        bgs::bin_softmax_inplace(seg, n, lane, m, S);
        term = coef * ((m + logf(S)) - zt);
        if (WRITE_GRAD) bgs::bin_grad_inplace(seg, n, lane, coef / S, coef, tgt);
      }
      if (lane == b) lacc += term;
    }
    // ---- box branch of this row (bbox_head.py:117-129 / gs_bbox_head_with0.py:173-185): the
    //      positive row's own class slot, four lanes of the last wave
    const int64_t slot = (a.R == 1) ? 0 : yraw;
    const bool pos = BOX && yraw > 0 && slot < a.R;
    if (BOX && wave == kWaves - 1) {
      float val = 0.f, g = 0.f;
      if (pos && lane < 4) {
        const float p = a.bbox_pred[((size_t)r * a.R + (size_t)slot) * 4 + lane];
        const float t = a.bbox_targets[(size_t)r * 4 + lane];
        const float bw = a.bbox_weights[(size_t)r * 4 + lane];
        val = gs_sl1(p - t, a.beta, g) * bw;
        g = g * bw * box_scale;
      }
      val = bgs::wave_sum(val);
      if (lane == 0) box_acc += val;
      if (a.dbbox && lane < 4) sh_box[lane] = g;
    }
    if (first) GS_MARK(5);                // wave 0: rank scan + softmax of its bins done
    __syncthreads();                      // gradient row (and the box gradient) complete
    if (first) GS_MARK(6);
    if (WRITE_GRAD) bgs::unstage_row<VEC>(row, a.dlogits + (size_t)r * W, W, tid, kBlock);
    if (BOX && a.dbbox) {                 // dense [N, 4R] gradient: zeros but for the positive slot
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 gp = {sh_box[0], sh_box[1], sh_box[2], sh_box[3]};
      f32x4* drow = reinterpret_cast<f32x4*>(a.dbbox + (size_t)r * a.R * 4);
      for (int c = tid; c < a.R; c += kBlock)
        drow[c] = (pos && c == (int)slot) ? gp : f32x4{0.f, 0.f, 0.f, 0.f};
      if (r + (int)gridDim.x < N) __syncthreads();   // sh_box is rewritten by the next row
    }
  }
  if (lane < B && (lane % kWaves) == wave)
    a.partial[(size_t)lane * gridDim.x + blockIdx.x] = lacc;
  if (BOX && wave == kWaves - 1 && lane == 0)
    a.partial[(size_t)B * gridDim.x + blockIdx.x] = box_acc;
  GS_MARK(7);
#undef GS_MARK
}

// ---- variants 2 / 3 (round 3): RPAR rows per workgroup IN PARALLEL, one 4-wave group each ----------------------------
// The prologue of the kernel above (every workgroup recounts the N labels) is paid once per ROW; more rows per
// workgroup one after the other were slower (the kernel is bound by a workgroup's serial path).  Here a workgroup is
// RPAR x 256 threads: all of them share ONE prologue (N labels over RPAR x 256 threads: 4 / RPAR label rows per
// thread, bit planes as in variant 1), then each 4-wave group runs its own row exactly as before — the serial path of
// a row is unchanged, the redundant counting drops RPAR x (1024 rows: 256 workgroups of 16 waves, one per CU, each
// label is looked at by 256 workgroups instead of 1024).  One row per group, no grid stride: N <= kMaxGrid, and the
// partials are written per ROW (column r of [B + 1][N]) — the same values in the same places as the one-row-per-
// workgroup kernels at grid = N, so everything downstream is bitwise unchanged.  Without row weights every row is
// real: the "real" plane is known in closed form (no ballots, no popcount pass).
__host__ __device__ inline size_t gs_head_multi_lds_bytes(int N, int C, int wpad, int rpar) {
  return sizeof(float) * ((size_t)rpar * wpad + BGS_WAVE * bgs::kSweep) + 2 * (size_t)((N + 63) & ~63) +
         2 * (size_t)((C + 7) & ~7) + sizeof(float) * 4 * rpar + sizeof(float) * (BGS_MAX_BINS + 2);
}

// ---- the reduction folded into the main launch (round 6) --------------------------------------------------------
// Every workgroup publishes its partials WRITE-THROUGH (agent-scope relaxed atomic stores: `global_store ... sc1`),
// drains them (`s_waitcnt vmcnt(0)` in every storing wave), and one lane takes a ticket (a returning agent-scope
// atomic add).  The workgroup that draws the last ticket reads ALL partials back with agent-scope relaxed atomic
// loads (`sc1`: served past its own L1; the producers' stores went through their L2s, so neither a fence nor an L2
// write-back is needed — cdna_hip_programming.md, Guideline 16, form R1) and finishes exactly as
// gs_head_reduce_kernel does: one wave per row of partials, the same lanes adding the same values in the same order,
// one DPP sum — bitwise the two-launch result whichever workgroup happens to be last.  It then advances the draw
// counter (every workgroup read it at its start: all of them have arrived) and zeroes the ticket for the next launch.
__device__ __forceinline__ void gs_store_partial(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gs_load_partial(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT));
}

// all threads of the workgroup call it after their partial stores; `scratch`: BGS_MAX_BINS + 2 floats of LDS nobody
// else touches; nwaves = waves of the workgroup; draw0 = the counter value this launch started from
__device__ __forceinline__ void gs_head_fold_tail(const GsHeadArgs& a, int G, bool has_box, float n_real_f,
                                                  uint64_t draw0, float* scratch, int nwaves) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int gw = bgs::uniform(tid >> 6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have left
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scratch[BGS_MAX_BINS + 1] = old == gridDim.x - 1 ? 1.f : 0.f;
  }
  __syncthreads();
  if (scratch[BGS_MAX_BINS + 1] == 0.f) return;             // workgroup-uniform
  const int B = a.B;
  const int rows = B + (has_box ? 1 : 0);
  for (int row = gw; row <= B; row += nwaves) {              // (gs_head_reduce_kernel: wave `row` of B + 1 waves)
    float acc = 0.f;
    if (row < rows) {
      // (no index clamp: one base address + immediate offsets.  Reads past the row's end stay inside the workspace —
      //  (B + 1) rows of G <= 2048 floats in kMaxGrid * (BGS_MAX_BINS + 1) — and are masked out of the sum.)
      const float* src = a.partial + (size_t)row * G + lane;
      for (int g0 = 0; g0 < G; g0 += BGS_WAVE * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = gs_load_partial(src + g0 + BGS_WAVE * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (g0 + lane + BGS_WAVE * u < G) ? v[u] : 0.f;
      }
    }
    float sum = bgs::wave_sum(acc);
    if (row == B && has_box) sum *= a.box_w / n_real_f;      // avg[0] = max(#real rows, 1)
    if (lane == 0) {
      scratch[row] = sum;
      a.fold_out[row] = sum;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int b = 0; b <= B; ++b) t += scratch[b];
    if (a.fold_total) a.fold_total[0] = t;
    if (a.fold_counter) a.fold_counter[0] = draw0 + 1ull;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// sum over the wave of values that are zero outside lanes 0..15: four DPP steps inside row 0 (every lane of the row
// ends up with the row's sum)
__device__ __forceinline__ int gs_row0_sum_i(int v) {
#ifndef BGS_NO_DPP
  v += __builtin_amdgcn_update_dpp(v, v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(v, v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);  // row_ror:8
  return __builtin_amdgcn_readlane(v, 0);
#else
  return bgs::wave_sum_i(v);
#endif
}

// bgs::bin_loss_registers with the gradient of the bin going from the registers straight to its columns of the
// global gradient row (64 consecutive floats per store instruction) instead of back into the LDS row: the wave that
// owns a bin finishes it alone, the row needs no third barrier and no LDS round trip on the way out (variants 4 / 5).
// The same arithmetic in the same order: bitwise the same gradient.
__device__ __forceinline__ float gs_bin_loss_direct(const float* __restrict__ seg, float* __restrict__ gseg, int n,
                                                    int lane, float coef, int tgt) {
  float x[bgs::kSweep];
  bool ok[bgs::kSweep];
#pragma unroll
  for (int u = 0; u < bgs::kSweep; ++u) {
    const int j = lane + BGS_WAVE * u;
    ok[u] = j < n;
    x[u] = seg[j];
  }
  const float zt = seg[tgt];
  float pm = -INFINITY;
#pragma unroll
  for (int u = 0; u < bgs::kSweep; ++u) pm = fmaxf(pm, ok[u] ? x[u] : -INFINITY);
  const float m = bgs::wave_max(pm);
  float ps = 0.f;
#pragma unroll
  for (int u = 0; u < bgs::kSweep; ++u) {
    x[u] = ok[u] ? __builtin_amdgcn_exp2f((x[u] - m) * bgs::kLog2e) : 0.f;
    ps += x[u];
  }
  const float S = bgs::wave_sum(ps);
  const float k = coef / S;
#pragma unroll
  for (int u = 0; u < bgs::kSweep; ++u) {
    const int j = lane + BGS_WAVE * u;
    if (ok[u]) gseg[j] = x[u] * k - (j == tgt ? coef : 0.f);
  }
  return coef * ((m + logf(S)) - zt);
}

template <int VEC, bool WRITE_GRAD, bool BOX, int RPAR, bool DIRECT>
__global__ __launch_bounds__(kBlock * RPAR, 6) void gs_head_multi_kernel(GsHeadArgs a) {   // (6 waves per SIMD: the 80 registers it had before the folded tail)
  constexpr int T = kBlock * RPAR;
  constexpr int RP = kFusedRowsPerPass / RPAR;      // label rows per thread and pass (a pass = 1024 rows)
  static_assert(RP >= 1, "RPAR <= 4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, C = a.C, B = a.B, W = a.W, wpad = a.wpad;
  const int NW = (N + 63) >> 6;
  const int cpad = (C + 7) & ~7;
  unsigned long long* sh_pl =
      reinterpret_cast<unsigned long long*>(smem + (size_t)RPAR * wpad + BGS_WAVE * bgs::kSweep);   // [16][NW]
  unsigned short* sh_cbits = reinterpret_cast<unsigned short*>(sh_pl) + ((N + 63) & ~63);
  float* sh_box = reinterpret_cast<float*>(sh_cbits + cpad);                                        // [RPAR][4]
  float* sh_fold = sh_box + 4 * RPAR;               // [BGS_MAX_BINS + 2]: the folded reduction's scratch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int gw = bgs::uniform(tid >> 6);            // wave of the workgroup
  const int grp = gw >> 2;                          // 4-wave group = row slot
  const int wave = gw & 3;                          // wave of the group
  const int htid = tid & (kBlock - 1);
  const int r = (int)blockIdx.x * RPAR + grp;       // the group's row
  const bool has_row = r < N;
  const int rc0 = has_row ? r : 0;
  uint64_t seed = a.seed;
  const uint64_t draw0 = a.seed_offset ? a.seed_offset[0] : 0ull;
  if (a.seed_offset) seed += 0x2545F4914F6CDD1Dull * draw0;
  const bool all_real = a.row_weights == nullptr;
  // (a load behind a branch is waited for where it is issued: without row weights the same load reads the labels'
  //  bytes and its value is ignored)
  const float* rw_src = all_real ? reinterpret_cast<const float*>(a.labels) : a.row_weights;

  // ---- loads: the group's label first (oldest), then everything independent, then the dependent gather
  const int64_t yraw = a.labels[rc0];
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const uint16_t* cb_src = a.class_bits ? a.class_bits + (tid * 8 < cpad ? tid * 8 : 0)
                                        : reinterpret_cast<const uint16_t*>(a.labels);
  const u32x4_t cb0 = *reinterpret_cast<const u32x4_t*>(cb_src);
  int64_t y0[RP];
  float rwv0[RP];
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const int q = tid + T * i;
    const int qc = q < N ? q : 0;
    y0[i] = a.labels[qc];
    rwv0[i] = rw_src[qc];
  }
  float* row = smem + (size_t)grp * wpad;
  float t0[VEC], t1[VEC];
  const int c0 = htid * VEC, c1 = c0 + kBlock * VEC;
  {
    const float* g = a.logits + (size_t)rc0 * W;
    bgs::load_vec<VEC>(g + (c0 < W ? c0 : 0), t0);
    bgs::load_vec<VEC>(g + (c1 < W ? c1 : 0), t1);
  }
  int my_bl = 0;
  if (lane < B) {
    const int64_t yc = yraw < 0 ? 0 : (yraw >= C ? (int64_t)C - 1 : yraw);
    my_bl = (int)a.l2b[(size_t)lane * C + yc];
  }
  // the salt of the wave's first sampled bin (bin 0 is never sampled): scalar work in the shadow of the loads
  const int first_sampled = wave ? wave : kWaves;
  const unsigned salt_first = DIRECT ? bgs::gs_bin_salt(seed, (uint32_t)first_sampled) : 0u;
  if (a.class_bits) {
    if (tid * 8 < cpad) *reinterpret_cast<u32x4_t*>(sh_cbits + tid * 8) = cb0;
    for (int c = (tid + T) * 8; c < cpad; c += T * 8)
      *reinterpret_cast<u32x4_t*>(sh_cbits + c) = *reinterpret_cast<const u32x4_t*>(a.class_bits + c);
  } else {
    for (int c = tid; c < C; c += T) {
      unsigned bits = 0u;
      for (int b = 0; b < B; ++b)
        if (a.l2b[(size_t)b * C + c] > 0) bits |= 1u << b;
      sh_cbits[c] = (unsigned short)bits;
    }
  }
  if (c0 < W) bgs::store_vec<VEC>(row + c0, t0);
  if (c1 < W) bgs::store_vec<VEC>(row + c1, t1);
  for (int c = c1 + kBlock * VEC; c < W; c += kBlock * VEC) {
    float t[VEC];
    bgs::load_vec<VEC>(a.logits + (size_t)rc0 * W + c, t);
    bgs::store_vec<VEC>(row + c, t);
  }
  __syncthreads();                          // class bits and the rows are in LDS

  // ---- bit planes: lane p of a wave stores plane p's ballot word of the wave's 64 rows
  auto pass = [&](const int base, const int64_t (&yv)[RP], const float (&rwv)[RP]) {
    unsigned bits[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int64_t y = yv[i] < 0 ? 0 : (yv[i] >= C ? (int64_t)C - 1 : yv[i]);
      bits[i] = sh_cbits[(int)y];
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int q = base + tid + T * i;
      const unsigned wbits = (q < N && (all_real || rwv[i] > 0.f)) ? (bits[i] | 0x8000u) : 0u;
      int lo = 0, hi = 0;
      for (int p = 1; p < B; ++p) {
        const unsigned long long m = __ballot((wbits >> p) & 1u);
        lo = gs_writelane((int)(unsigned)m, p, lo);
        hi = gs_writelane((int)(unsigned)(m >> 32), p, hi);
      }
      if (!all_real) {
        const unsigned long long mr = __ballot(wbits >> 15);
        lo = gs_writelane((int)(unsigned)mr, 15, lo);
        hi = gs_writelane((int)(unsigned)(mr >> 32), 15, hi);
      }
      const int w = ((base + T * i) >> 6) + gw;
      if (lane < 16 && w < NW)
        sh_pl[lane * NW + w] = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
    }
  };
  pass(0, y0, rwv0);
  for (int base = T * RP; base < N; base += T * RP) {
    int64_t yv[RP];
    float rwv[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int q = base + tid + T * i;
      const int qc = q < N ? q : 0;
      yv[i] = a.labels[qc];
      rwv[i] = rw_src[qc];
    }
    pass(base, yv, rwv);
  }
  __syncthreads();                          // planes are in LDS

  const bool narrow = NW <= 16;             // the plane words sit in lanes 0..15: 4-step sums
  unsigned long long Rw;                    // lane w: word w of the "real" plane
  int n_real;
  if (all_real) {
    const int wn = N >> 6;
    Rw = lane < wn ? ~0ull : (lane == wn ? ((1ull << (N & 63)) - 1ull) : 0ull);
    n_real = N;
  } else {
    Rw = lane < NW ? sh_pl[15 * NW + lane] : 0ull;
    n_real = narrow ? gs_row0_sum_i(__popcll(Rw)) : bgs::wave_sum_i_fast(__popcll(Rw));
  }
  const float box_scale = a.box_w / fmaxf((float)n_real, 1.f);
  float lacc = 0.f, box_acc = 0.f;
  const int64_t slot = (a.R == 1) ? 0 : yraw;
  const bool pos_row = BOX && has_row && yraw > 0 && slot < a.R;

  if (has_row) {                            // wave-uniform
    const int wr = r >> 6;
    bool real_r;
    {
      const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)Rw, wr);
      const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(Rw >> 32), wr);
      real_r = (((r & 32) ? rhi : rlo) >> (r & 31)) & 1u;
    }
    for (int b = wave; b < B; b += kWaves) {  // bins are independent: one wave each
      const int s = a.geom.start[b], n = a.geom.len[b];
      const int bl_b = __builtin_amdgcn_readlane(my_bl, b);
      int mode_b, k_b = 0, nbg_b = 0;
      unsigned pos = 0u;
      float total, w = 0.f;
      if (b == 0) {
        mode_b = 1;
        total = (float)n_real;
      } else {
        // one pass over the bin's plane: foreground count and the row's candidate position together
        const unsigned long long Fw = lane < NW ? sh_pl[b * NW + lane] : 0ull;
        const unsigned long long below = (1ull << (r & 63)) - 1ull;
        const unsigned long long msk = lane < wr ? ~0ull : (lane == wr ? below : 0ull);
        const int packed = (__popcll(Fw) << 16) | __popcll((Rw ^ Fw) & msk);
        const unsigned tot = (unsigned)(narrow ? gs_row0_sum_i(packed) : bgs::wave_sum_i_fast(packed));
        const int n_fg = (int)(tot >> 16);
        pos = tot & 0xffffu;
        nbg_b = n_real - n_fg;
        if (n_fg == 0) {
          mode_b = 0;
          total = 0.f;
        } else {
          k_b = (int)((double)n_fg * a.ratio);
          mode_b = (k_b >= nbg_b) ? 1 : 2;
          total = mode_b == 1 ? (float)n_real : (float)(n_fg + k_b);
        }
      }
      const float av = fmaxf(total, 1.f);
      const float scale_b = (1.f / av) * a.lw[b];
      if (r == 0 && lane == 0 && a.avg_out) a.avg_out[b] = av;
      if (real_r) {
        if (mode_b == 1 || (mode_b == 2 && bl_b > 0)) {
          w = 1.f;
        } else if (mode_b == 2) {
          const unsigned salt = (DIRECT && b == first_sampled) ? salt_first : bgs::gs_bin_salt(seed, (uint32_t)b);
          w = (k_b > 0 && bgs::gs_perm(salt, pos, (unsigned)nbg_b) < (unsigned)k_b) ? 1.f : 0.f;
        }
      }
      const float coef = w * scale_b;
      if (lane == 0) {
        if (a.bl_out) a.bl_out[(size_t)b * N + r] = bl_b;
        if (a.w_out) a.w_out[(size_t)b * N + r] = w;
      }
      const int tgt = min(max(bl_b, 0), n - 1);
      float* seg = row + s;
      float* gseg = (WRITE_GRAD && DIRECT) ? a.dlogits + (size_t)r * W + s : nullptr;   // the bin's columns of the gradient row
      if (coef == 0.f) {
        if (WRITE_GRAD)
          for (int j = lane; j < n; j += BGS_WAVE) (DIRECT ? gseg : seg)[j] = 0.f;
        continue;
      }
      float term;
      if (n <= BGS_WAVE * bgs::kSweep) {
        term = (WRITE_GRAD && DIRECT) ? gs_bin_loss_direct(seg, gseg, n, lane, coef, tgt)
                                      : bgs::bin_loss_registers<WRITE_GRAD>(seg, n, lane, coef, tgt);
      } else {
        const float zt = seg[tgt];
        float m, S;
        bgs::bin_softmax_inplace(seg, n, lane, m, S);
        term = coef * ((m + logf(S)) - zt);
        if (WRITE_GRAD) {
          bgs::bin_grad_inplace(seg, n, lane, coef / S, coef, tgt);
          if (DIRECT)        // the wave's own LDS writes, read back in order: no barrier
            for (int j = lane; j < n; j += BGS_WAVE) gseg[j] = seg[j];
        }
      }
      if (lane == b) lacc += term;
    }
    // box branch of the row: four lanes of the group's last wave
    if (BOX && wave == kWaves - 1) {
      float val = 0.f, g = 0.f;
      if (pos_row && lane < 4) {
        const float p = a.bbox_pred[((size_t)r * a.R + (size_t)slot) * 4 + lane];
        const float t = a.bbox_targets[(size_t)r * 4 + lane];
        const float bw = a.bbox_weights[(size_t)r * 4 + lane];
        val = gs_sl1(p - t, a.beta, g) * bw;
        g = g * bw * box_scale;
      }
      val = bgs::wave_sum(val);
      if (lane == 0) box_acc += val;
      if (a.dbbox && lane < 4) sh_box[grp * 4 + lane] = g;
    }
  }
  // gradient rows (and the box gradients) complete; with direct stores only the dense box gradient needs the barrier
  if (!(WRITE_GRAD && DIRECT) || (BOX && a.dbbox)) __syncthreads();
  if (has_row) {
    if (WRITE_GRAD && !DIRECT) bgs::unstage_row<VEC>(row, a.dlogits + (size_t)r * W, W, htid, kBlock);
    if (BOX && a.dbbox) {                   // dense [N, 4R] gradient: zeros but for the positive slot
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 gp = {sh_box[grp * 4 + 0], sh_box[grp * 4 + 1], sh_box[grp * 4 + 2], sh_box[grp * 4 + 3]};
      f32x4* drow = reinterpret_cast<f32x4*>(a.dbbox + (size_t)r * a.R * 4);
      for (int c = htid; c < a.R; c += kBlock)
        drow[c] = (pos_row && c == (int)slot) ? gp : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // partials per ROW: [B + 1][N]
    if (a.ticket) {                           // (kernel-argument uniform) write-through: read back by the last workgroup
      if (lane < B && (lane % kWaves) == wave) gs_store_partial(a.partial + (size_t)lane * N + r, lacc);
      if (BOX && wave == kWaves - 1 && lane == 0) gs_store_partial(a.partial + (size_t)B * N + r, box_acc);
    } else {
      if (lane < B && (lane % kWaves) == wave) a.partial[(size_t)lane * N + r] = lacc;
      if (BOX && wave == kWaves - 1 && lane == 0) a.partial[(size_t)B * N + r] = box_acc;
    }
  }
  if (a.ticket) gs_head_fold_tail(a, N, BOX, fmaxf((float)n_real, 1.f), draw0, sh_fold, T / BGS_WAVE);
}

// out[b] = sum_g partial[b][g] for b < B (loss weights are already inside), out[B] = box loss
// (scaled by loss_weight / avg[0]; 0 without a box branch), total[0] = their sum — the scalar the
// reference forms in parse_losses; fixed summation order.  One wave per row of partials (all loads
// of a row in flight at once, one DPP sum), one barrier.  `counter` (the device draw counter read
// by the NEXT call's main kernel) is advanced here, behind every reader of this call.
__global__ __launch_bounds__(1024) void gs_head_reduce_kernel(const float* __restrict__ partial, int G,
                                                              int B, int has_box, float box_w,
                                                              const float* __restrict__ avg,
                                                              float* __restrict__ out,
                                                              float* __restrict__ total,
                                                              uint64_t* __restrict__ counter) {
  __shared__ float fin[BGS_MAX_BINS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;     // 16 waves >= B + 1 rows
  const int rows = B + (has_box ? 1 : 0);
  // the two values the tail depends on start with the partials (behind the sum / the barrier each was one more
  // memory round trip on the kernel's only path)
  const uint64_t draw = (tid == 0 && counter) ? counter[0] : 0ull;
  const float avg0 = (has_box && wave == B) ? avg[0] : 1.f;
  if (wave <= B) {
    float acc = 0.f;
    if (wave < rows) {
      const float* row = partial + (size_t)wave * G;
      for (int g0 = 0; g0 < G; g0 += BGS_WAVE * 16) {       // 16 loads in flight per lane (G <= 2048)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int g = g0 + lane + BGS_WAVE * u;
          v[u] = row[g < G ? g : 0];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (g0 + lane + BGS_WAVE * u < G) ? v[u] : 0.f;
      }
    }
    float s = bgs::wave_sum(acc);
    if (wave == B && has_box) s *= box_w / avg0;
    if (lane == 0) {
      fin[wave] = s;
      out[wave] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int b = 0; b <= B; ++b) t += fin[b];
    if (total) total[0] = t;
    if (counter) counter[0] = draw + 1ull;
  }
}

// dlogits[:, bin b] *= gt[b] + gT;  dbbox *= gt[B] + gT  (gt [B + 1] = upstream gradient of the loss
// terms or null = 0, gT = upstream gradient of the total or null = 0); early-out when every factor
// is 1 (the usual case: total.backward()).
__global__ __launch_bounds__(kBlock) void gs_head_scale_grad_kernel(float* __restrict__ dlogits,
                                                                    float* __restrict__ dbbox,
                                                                    bgs::BinGeom geom,
                                                                    const float* __restrict__ gterms,
                                                                    const float* __restrict__ gtotal,
                                                                    int N, int B, int W, int R4) {
  const float gt = gtotal ? gtotal[0] : 0.f;
  bool all_one = true;
  for (int b = 0; b < B; ++b) all_one = all_one && ((gterms ? gterms[b] : 0.f) + gt == 1.f);
  const float gbox = (gterms ? gterms[B] : 0.f) + gt;
  extern __shared__ __attribute__((aligned(16))) float scale[];
  if (dlogits && !all_one) {
    for (int c = threadIdx.x; c < W; c += kBlock) {
      float sc = 0.f;
      for (int b = 0; b < B; ++b)
        if (c >= geom.start[b] && c < geom.start[b] + geom.len[b]) sc = (gterms ? gterms[b] : 0.f) + gt;
      scale[c] = sc;
    }
    __syncthreads();
    const size_t total = (size_t)N * W;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock)
      dlogits[i] *= scale[(int)(i % (size_t)W)];
  }
  if (dbbox && gbox != 1.f) {
    const size_t total = (size_t)N * R4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock)
      dbbox[i] *= gbox;
  }
}

template <int VEC>
void launch_rowwave(bool grad, int grid, hipStream_t st, const float* logits, const int32_t* bl,
                    const float* w, const float* avg, const bgs::BinGeom& geom, int N, int B,
                    int W, float* partial, float* dlogits) {
  const int wpad = (W + 3) & ~3;
  // + slack: the register sweep reads up to 64*kSweep floats from a bin's start
  const size_t lds = sizeof(float) * (2 * (size_t)wpad + BGS_WAVE * bgs::kSweep);
  const bool pf = g_rowwave_pf && W <= 2 * kBlock * VEC && N > grid;    // a second row per workgroup to fetch ahead
  if (grad && pf && g_rowwave_pf >= 2) {
#define GS_NT(NT_) case NT_: hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, true, true, NT_>), dim3(grid), dim3(kBlock), lds, st, \
                                                logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits); break;
    switch (g_rowwave_pf >= 5 ? 2 : g_rowwave_pf - 1) { GS_NT(1) GS_NT(2) default: GS_NT(3) }   // modes 5 - 7 below their row threshold: mode 3
#undef GS_NT
  } else if (grad && pf)
    hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, true, true>), dim3(grid), dim3(kBlock), lds, st,
                       logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);
  else if (grad)
    hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, true>), dim3(grid), dim3(kBlock), lds, st,
                       logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);
  else
    hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, false>), dim3(grid), dim3(kBlock), lds, st,
                       logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);
}

// one workgroup per row; at most kMaxGrid workgroups (grid-stride beyond)
inline int loss_grid(int N) { return N <= 0 ? 1 : (N < kMaxGrid ? N : kMaxGrid); }

// Row-per-wave kernel (modes 5 - 7): eligible for 16-byte rows of at most 2048 floats whose bins fit the register sweep.
// Where it pays (profiles/r10a / r10c_gs_stream_ab.txt, two boxes): 4096 rows 8.6 vs 8.9 - 9.2 us, 8192 rows 14.2 vs 16.4 -
// 17.1 us (-15 %); from 16,384 rows the 4-wave-per-row kernel (mode 3) is the faster one again (28.0 vs 30.0 us; at
// 65,536 rows 122 vs 126 us — both AT the rate of a hipMemcpyAsync D2D of the same 324 + 324 MB on the same box,
// 5.3 - 5.4 TB/s).  Default (mode 5): rows in [kWavePrivMinRows, kWavePrivMaxRows); modes 6 / 7 ignore the upper bound.
// Grid: every workgroup resident at once (LDS: 4 rows + read slack each), rows dealt round-robin.
constexpr int kWavePrivMinRows = 4097;      // (up to the fused head's 4096 rows the two paths stay bitwise equal: the tests pin it)
constexpr int kWavePrivMaxRows = 12288;
int g_wavepriv_min_rows = kWavePrivMinRows;

inline bool wavepriv_eligible(const bgs::BinGeom& geom, int B, int W, int vec) {
  if (vec != 4 || W > 2048) return false;
  for (int b = 0; b < B; ++b)
    if (geom.len[b] > BGS_WAVE * bgs::kSweep) return false;
  return true;
}

inline int wavepriv_grid(int N, size_t lds) {
  int per_cu = (int)((160u * 1024u) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const int cap = 256 * per_cu;
  const int want = (N + kWaves - 1) / kWaves;
  return want < cap ? want : cap;
}

// returns the grid it launched (the number of per-bin partial sums in the workspace)
int launch_wavepriv(bool grad, hipStream_t st, const float* logits, const int32_t* bl, const float* w,
                    const float* avg, const bgs::BinGeom& geom, int N, int B, int W, float* partial,
                    float* dlogits) {
  const int wpad = W;                                      // W % 4 == 0
  const size_t lds = sizeof(float) * ((size_t)kWaves * wpad + bgs::row_read_slack(geom, B, W));
  const int grid = wavepriv_grid(N, lds);
  const int kv = (W / 4 + BGS_WAVE - 1) / BGS_WAVE;
  // Non-temporal row loads (mode 7, A/B only): a plain copy of this working set gains 10 - 15 % from the hint on BOTH
  // sides once read + written bytes exceed the 256 MB Infinity Cache (324 + 324 MB: 5.4 -> 6.2 - 6.5 TB/s,
  // tools/hbm_copy_bench.hip, profiles/r10b_hbm_copy_bench.txt) — inside this kernel and the 4-wave one it LOSES 2 - 20 %
  // at every size (profiles/r10c_gs_stream_ab.txt), so no default mode uses it.
  const bool ntl = g_rowwave_pf == 7;
#define GS_WP(KV_)                                                                                          \
  do {                                                                                                      \
    if (grad && ntl)                                                                                        \
      hipLaunchKernelGGL((gs_loss_wavepriv_kernel<KV_, true, true>), dim3(grid), dim3(kBlock), lds, st,     \
                         logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);                        \
    else if (grad)                                                                                          \
      hipLaunchKernelGGL((gs_loss_wavepriv_kernel<KV_, true>), dim3(grid), dim3(kBlock), lds, st, logits,   \
                         bl, w, avg, geom, N, B, W, wpad, partial, dlogits);                                \
    else                                                                                                    \
      hipLaunchKernelGGL((gs_loss_wavepriv_kernel<KV_, false>), dim3(grid), dim3(kBlock), lds, st, logits,  \
                         bl, w, avg, geom, N, B, W, wpad, partial, dlogits);                                \
  } while (0)
  if (kv <= 2) GS_WP(2);
  else if (kv <= 5) GS_WP(5);
  else GS_WP(8);
#undef GS_WP
  return grid;
}

}  // namespace

// tuning / test hook: 0 = the round-3 kernel | 1 = next row fetched ahead | 2 / 3 / 4 = .. + non-temporal row loads /
// gradient stores / both (3 = default)
extern "C" void bgs_gs_loss_tuning(int prefetch) { g_rowwave_pf = prefetch < 0 ? 0 : (prefetch > 7 ? 5 : prefetch); }
// rows from which mode 5 applies (< 0: back to the default); a test / A/B hook like the one above
extern "C" void bgs_gs_loss_wavepriv_min_rows(int rows) { g_wavepriv_min_rows = rows < 0 ? kWavePrivMinRows : rows; }

extern "C" size_t bgs_gs_loss_workspace_bytes(int N, int B) {
  (void)N;
  (void)B;
  return (size_t)kMaxGrid * (BGS_MAX_BINS + 1) * sizeof(float);
}

// Test hook: BGS_GS_FORCE_GENERIC=1 (read at every call) routes through the fallback kernel.
static bool force_generic() {
  const char* e = getenv("BGS_GS_FORCE_GENERIC");
  return e && e[0] == '1';
}

extern "C" int bgs_gs_loss_fwd_bwd(const float* logits, const int32_t* bin_labels,
                                   const int64_t* host_pred_slice, const float* weights,
                                   const float* avg, int N, int B, int W, float* loss_out,
                                   float* dlogits, void* workspace, bgs_stream_t stream) {
  if (N < 0 || B <= 0 || W <= 0) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS) return BGS_ERR_UNSUPPORTED;
  if (!workspace || !host_pred_slice) return BGS_ERR_INVALID_ARG;
  if (N > 0 && (!logits || !bin_labels)) return BGS_ERR_INVALID_ARG;
  bgs::BinGeom geom;
  int tiles = 0;
  const int rc = bgs::make_bin_geom(host_pred_slice, B, W, &geom, &tiles);
  if (rc != BGS_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const bool grad = dlogits != nullptr;
  int grid = loss_grid(N);
  if (N == 0) {
    (void)hipMemsetAsync(partial, 0, sizeof(float) * B, st);
  } else {
    const uintptr_t al = (uintptr_t)logits | (uintptr_t)(dlogits ? dlogits : logits);
    int vec = 1;
    if (W % 4 == 0 && al % 16 == 0) vec = 4;
    else if (W % 2 == 0 && al % 8 == 0) vec = 2;
    // wave kernel: bins must tile [0, W) (true for tables built by tools/lvis_analyse.py) and
    // two staged rows must fit the 64 KB default LDS window
    if (!force_generic() && tiles && g_rowwave_pf >= 5 && N >= g_wavepriv_min_rows &&
        (g_rowwave_pf > 5 || N < kWavePrivMaxRows) && wavepriv_eligible(geom, B, W, vec)) {
      grid = launch_wavepriv(grad, st, logits, bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
    } else if (!force_generic() && tiles && W <= 7936) {
      if (vec == 4) launch_rowwave<4>(grad, grid, st, logits, bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
      else if (vec == 2) launch_rowwave<2>(grad, grid, st, logits, bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
      else launch_rowwave<1>(grad, grid, st, logits, bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
    } else {
      if (grad)
        hipLaunchKernelGGL((gs_loss_generic_kernel<true>), dim3(grid), dim3(kBlock), 0, st, logits,
                           bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
      else
        hipLaunchKernelGGL((gs_loss_generic_kernel<false>), dim3(grid), dim3(kBlock), 0, st,
                           logits, bin_labels, weights, avg, geom, N, B, W, partial, dlogits);
    }
  }
  // loss_out == NULL: leave the per-workgroup partials in the workspace (bgs_gs_loss_reduce
  // finishes the job) — lets a profiler time the streaming kernel on its own.
  if (loss_out)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partial, grid, B,
                       loss_out, 1.0f);
  BGS_RETURN_LAUNCH_STATUS();
}

namespace {

// shared launcher of the fused head kernel; returns the grid in *grid_out
unsigned long long* g_gs_tstamps = nullptr;

// Tickets of the folded reduction (gs_head_fold_tail): a small pool of zeroed device words per device, handed out
// round-robin — two head launches in flight at once (other streams, other graphs) take different words; the last
// workgroup of a launch leaves its word zero.  Allocated on first use; a first use under stream capture (hipMalloc is
// not capturable) falls back to the separate reduce launch for that call.
constexpr int kTicketPool = 64, kTicketDevices = 16;
unsigned* g_ticket_pool[kTicketDevices] = {};
unsigned g_ticket_next = 0;
// bgs_gs_head_fold / BGS_GS_HEAD_FOLD: 0 (default) = the separate reduce launch | 1 = the reduction inside the main launch.
// MEASURED (profiles/r10h_gs_head_fold_ab.txt, hipGraph replay of the whole head step, interleaved): N = 1024 16.6 us folded
// vs 14.7 us as two launches, N = 512 15.3 vs 14.5, N = 2048 26.0 vs 24.5 — the fold LOSES 1 - 2 us: 256 - 512 arrivals on
// one ticket word cost 3 - 4 us (the guide's `fanin` row), every workgroup drains its gradient stores before it may
// arrive, and the last workgroup's read-back of the 24 KB of partials is a second memory round trip on the launch's only
// path — more than the ~1.5 us launch boundary + the 4.9 us reduce kernel it replaces, whose own loads start the
// moment the main kernel retires.  Kept as a tested, bit-identical A/B arm ("or prove by A/B why not").
int g_head_fold = -1;

bool head_fold_enabled() {
  if (g_head_fold < 0) {
    const char* e = getenv("BGS_GS_HEAD_FOLD");
    g_head_fold = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_head_fold != 0;
}

unsigned* take_ticket(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTicketDevices) return nullptr;
  if (!g_ticket_pool[dev]) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
    unsigned* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned) * kTicketPool) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (hipMemset(p, 0, sizeof(unsigned) * kTicketPool) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p);
      return nullptr;
    }
    g_ticket_pool[dev] = p;
  }
  return g_ticket_pool[dev] + (g_ticket_next++ % kTicketPool);
}

// Rows per workgroup of the fused head kernel.  Every workgroup recounts the N labels (flags, per-bin
// counts) before its first row; with R rows per workgroup that prologue is paid N / R times instead of
// N times, against fewer workgroups to hide latency with.  0 = default; BGS_GS_HEAD_ROWS / the tuning
// entry override (sweep: profiles/r5k_gs_head_rows_sweep.txt).
int g_head_rows = -1;
int head_rows_per_wg() {
  if (g_head_rows < 0) {
    const char* e = getenv("BGS_GS_HEAD_ROWS");
    g_head_rows = e ? atoi(e) : 0;
    if (g_head_rows < 0 || g_head_rows > 64) g_head_rows = 0;
  }
  return g_head_rows;
}

// variant of the fused head kernel: explicit (bgs_gs_head_variant / BGS_GS_HEAD_VARIANT = 0..5) or automatic — rows
// in parallel behind one prologue wherever the per-row partials fit the workspace (N <= kMaxGrid): four per
// workgroup once that still gives every CU a workgroup (N >= 1024), else two; beyond, one row per workgroup with bit
// planes (profiles/r6w_gs_head_ab.txt: 10.3 / 9.7 / 7.7 / 7.7 us at N = 1024, 8.9 / 8.6 / 7.2 / 8.4 at 512,
// 21.3 / 20.8 / 14.9 / 14.0 at 2048 for variants 0 / 1 / 2 / 3).  Direct gradient stores (variants 4 / 5) win while
// the launch is latency-bound and lose once the store instructions count (profiles/r6z_gs_head_ab.txt: 7.43 vs 7.68 us
// at N = 1024, 7.07 vs 7.19 at 512, 15.3 vs 13.9 at 2048): taken up to N = 1024.
int g_head_variant = -2;                    // -2: not read yet, -1: automatic
int head_variant_for(int N) {
  if (g_head_variant == -2) {
    const char* e = getenv("BGS_GS_HEAD_VARIANT");
    g_head_variant = e ? atoi(e) : -1;
    if (g_head_variant < -1 || g_head_variant > 5) g_head_variant = -1;
  }
  int v = g_head_variant;
  if (v < 0) v = N < 1024 ? 4 : (N == 1024 ? 5 : 3);
  if (v >= 2 && N > kMaxGrid) v = 1;
  return v;
}

int launch_gs_head(GsHeadArgs& a, const int64_t* host_pred_slice, const float* host_bin_loss_weight,
                   hipStream_t st, int* grid_out) {
  a.tstamps = g_gs_tstamps;
  if (a.N <= 0 || a.C <= 0 || a.B <= 0 || a.W <= 0) return BGS_ERR_INVALID_ARG;
  if (a.B > BGS_MAX_BINS - 1 || a.N > kFusedMaxN) return BGS_ERR_UNSUPPORTED;
  int tiles = 0;
  const int rc = bgs::make_bin_geom(host_pred_slice, a.B, a.W, &a.geom, &tiles);
  if (rc != BGS_OK) return rc;
  if (!tiles) return BGS_ERR_UNSUPPORTED;
  for (int b = 0; b < BGS_MAX_BINS; ++b)
    a.lw[b] = (host_bin_loss_weight && b < a.B) ? host_bin_loss_weight[b] : 1.f;
  a.wpad = (a.W + 3) & ~3;
  const size_t lds = gs_head_lds_bytes(a.N, a.C, a.wpad);
  if (lds > 64 * 1024) return BGS_ERR_UNSUPPORTED;      // the default LDS window (callers fall back
                                                         // to bgs_gs_prepare + bgs_gs_loss_fwd_bwd)
  int grid = loss_grid(a.N);
  const int rows_per_wg = head_rows_per_wg();
  if (rows_per_wg > 1) grid = (a.N + rows_per_wg - 1) / rows_per_wg;
  *grid_out = grid;
  bgs_internal_census_bump(BGS_CENSUS_GS_HEAD_FUSED);
  const uintptr_t al = (uintptr_t)a.logits | (uintptr_t)(a.dlogits ? a.dlogits : a.logits);
  const bool grad = a.dlogits != nullptr;
  const bool box = a.bbox_pred != nullptr;
  // variants 2 / 3: RPAR rows per workgroup in parallel (partials per row: N columns)
  const int variant = head_variant_for(a.N);
  const int rpar = (variant == 3 || variant == 5) ? 4 : ((variant == 2 || variant == 4) ? 2 : 1);
  const bool direct = variant >= 4;
  if (rpar > 1 && a.N <= kMaxGrid && rows_per_wg <= 1 &&
      gs_head_multi_lds_bytes(a.N, a.C, a.wpad, rpar) <= 64 * 1024) {
    const size_t mlds = gs_head_multi_lds_bytes(a.N, a.C, a.wpad, rpar);
    const int mgrid = (a.N + rpar - 1) / rpar;
    *grid_out = a.N;
#define BGS_MULTI_LAUNCH2(VEC_, GRAD_, BOX_, DIR_)                                                                  \
  do {                                                                                                              \
    if (rpar == 4)                                                                                                  \
      hipLaunchKernelGGL((gs_head_multi_kernel<VEC_, GRAD_, BOX_, 4, DIR_>), dim3(mgrid), dim3(kBlock * 4), mlds, st, a); \
    else                                                                                                            \
      hipLaunchKernelGGL((gs_head_multi_kernel<VEC_, GRAD_, BOX_, 2, DIR_>), dim3(mgrid), dim3(kBlock * 2), mlds, st, a); \
  } while (0)
#define BGS_MULTI_LAUNCH(VEC_, GRAD_, BOX_)                                                                         \
  do {                                                                                                              \
    if (GRAD_ && direct) BGS_MULTI_LAUNCH2(VEC_, GRAD_, BOX_, GRAD_);                                               \
    else BGS_MULTI_LAUNCH2(VEC_, GRAD_, BOX_, false);                                                               \
  } while (0)
#define BGS_MULTI_VEC(VEC_)                                                                                   \
  do {                                                                                                        \
    if (grad) { if (box) BGS_MULTI_LAUNCH(VEC_, true, true); else BGS_MULTI_LAUNCH(VEC_, true, false); }      \
    else { if (box) BGS_MULTI_LAUNCH(VEC_, false, true); else BGS_MULTI_LAUNCH(VEC_, false, false); }         \
  } while (0)
    if (a.W % 4 == 0 && al % 16 == 0) BGS_MULTI_VEC(4);
    else if (a.W % 2 == 0 && al % 8 == 0) BGS_MULTI_VEC(2);
    else BGS_MULTI_VEC(1);
#undef BGS_MULTI_VEC
#undef BGS_MULTI_LAUNCH
#undef BGS_MULTI_LAUNCH2
    return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
  }
  a.ticket = nullptr;                       // (one row per workgroup, N > 2048: the separate reduce launch)
  const bool planes = variant >= 1;
#define BGS_HEAD_LAUNCH(VEC_, GRAD_, BOX_)                                                              \
  do {                                                                                                  \
    if (planes)                                                                                         \
      hipLaunchKernelGGL((gs_head_fused_kernel<VEC_, GRAD_, BOX_, 1>), dim3(grid), dim3(kBlock), lds, st, a); \
    else                                                                                                \
      hipLaunchKernelGGL((gs_head_fused_kernel<VEC_, GRAD_, BOX_, 0>), dim3(grid), dim3(kBlock), lds, st, a); \
  } while (0)
#define BGS_HEAD_VEC(VEC_)                                                                        \
  do {                                                                                            \
    if (grad) { if (box) BGS_HEAD_LAUNCH(VEC_, true, true); else BGS_HEAD_LAUNCH(VEC_, true, false); }   \
    else { if (box) BGS_HEAD_LAUNCH(VEC_, false, true); else BGS_HEAD_LAUNCH(VEC_, false, false); }      \
  } while (0)
  if (a.W % 4 == 0 && al % 16 == 0) BGS_HEAD_VEC(4);
  else if (a.W % 2 == 0 && al % 8 == 0) BGS_HEAD_VEC(2);
  else BGS_HEAD_VEC(1);
#undef BGS_HEAD_VEC
#undef BGS_HEAD_LAUNCH
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}

}  // namespace

// _remap_labels + _sample_others + loss forward + backward in ONE launch (+ the partial reduce):
// see gs_head_fused_kernel.  N <= 4096, bins must tile [0, W), no per-class reweighting (those
// cases use bgs_gs_prepare + bgs_gs_loss_fwd_bwd; BGS_ERR_UNSUPPORTED also when the rows do not fit
// the 64 KB LDS window).  avg_out [B] is always written (the box loss normaliser reads bin 0's);
// bin_labels_out / weights_out [B, N] are optional (tests).
extern "C" int bgs_gs_head_loss_fused(const float* logits, const int64_t* labels,
                                      const int64_t* label2binlabel, const float* row_weights,
                                      const int64_t* host_pred_slice, int N, int C, int B, int W,
                                      double others_sample_ratio, uint64_t seed,
                                      const uint64_t* seed_offset, float* loss_out, float* dlogits,
                                      float* avg_out, int32_t* bin_labels_out, float* weights_out,
                                      void* workspace, bgs_stream_t stream) {
  if (!logits || !labels || !label2binlabel || !host_pred_slice || !loss_out || !avg_out || !workspace)
    return BGS_ERR_INVALID_ARG;
  GsHeadArgs a = {};
  a.logits = logits; a.labels = labels; a.l2b = label2binlabel; a.row_weights = row_weights;
  a.N = N; a.C = C; a.B = B; a.W = W; a.ratio = others_sample_ratio; a.seed = seed;
  a.seed_offset = seed_offset; a.partial = (float*)workspace; a.dlogits = dlogits;
  a.avg_out = avg_out; a.bl_out = bin_labels_out; a.w_out = weights_out;
  int grid = 0;
  const int rc = launch_gs_head(a, host_pred_slice, nullptr, (hipStream_t)stream, &grid);
  if (rc != BGS_OK) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a.partial,
                     grid, B, loss_out, 1.0f);
  BGS_RETURN_LAUNCH_STATUS();
}

// The whole GSBBoxHeadWith0.loss() (gs_bbox_head_with0.py:147-186) as two launches:
//   main kernel  = label remap + "others" sampling + per-bin loss fwd + bwd + the box branch
//   reduce       = loss_out[0..B-1] per-bin losses (x bin_loss_weight), loss_out[B] = loss_bbox
//                  (x box_loss_weight / #real rows), total_out[0] = their sum; advances
//                  *draw_counter (device uint64, read by the main kernel as the draw index).
// bbox_pred == NULL: no box branch (loss_out[B] = 0).  dbbox_pred (dense [N, 4R] gradient) only
// when the box branch trains.  loss_out == NULL: main kernel only (profiling hook; the counter is
// not advanced).  Same limits as bgs_gs_head_loss_fused.
extern "C" int bgs_gs_head_step(const float* logits, const int64_t* labels,
                                const int64_t* label2binlabel, const uint16_t* class_bin_mask,
                                const float* row_weights, const int64_t* host_pred_slice,
                                const float* host_bin_loss_weight, int N, int C, int B, int W,
                                double others_sample_ratio, uint64_t seed, uint64_t* draw_counter,
                                const float* bbox_pred, const float* bbox_targets,
                                const float* bbox_weights, int num_reg_classes, float beta,
                                float box_loss_weight, float* loss_out, float* total_out,
                                float* dlogits, float* dbbox_pred, float* avg_out,
                                int32_t* bin_labels_out, float* weights_out, void* workspace,
                                bgs_stream_t stream) {
  if (!logits || !labels || !label2binlabel || !host_pred_slice || !avg_out || !workspace)
    return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)class_bin_mask & 15) return BGS_ERR_INVALID_ARG;
  if (bbox_pred) {
    if (!bbox_targets || !bbox_weights || num_reg_classes <= 0 || !(beta > 0.f)) return BGS_ERR_INVALID_ARG;
    if (((uintptr_t)bbox_pred | (uintptr_t)bbox_targets | (uintptr_t)bbox_weights |
         (uintptr_t)dbbox_pred) % 16 != 0)
      return BGS_ERR_INVALID_ARG;
  } else if (dbbox_pred) {
    return BGS_ERR_INVALID_ARG;
  }
  GsHeadArgs a = {};
  a.logits = logits; a.labels = labels; a.l2b = label2binlabel; a.class_bits = class_bin_mask;
  a.row_weights = row_weights;
  a.N = N; a.C = C; a.B = B; a.W = W; a.ratio = others_sample_ratio; a.seed = seed;
  a.seed_offset = draw_counter; a.partial = (float*)workspace; a.dlogits = dlogits;
  a.avg_out = avg_out; a.bl_out = bin_labels_out; a.w_out = weights_out;
  a.bbox_pred = bbox_pred; a.bbox_targets = bbox_targets; a.bbox_weights = bbox_weights;
  a.dbbox = dbbox_pred; a.R = num_reg_classes; a.beta = beta; a.box_w = box_loss_weight;
  int grid = 0;
  if (loss_out && head_fold_enabled()) {    // the reduction inside the main launch (N <= 2048: launch_gs_head keeps it)
    a.ticket = take_ticket((hipStream_t)stream);
    a.fold_out = loss_out;
    a.fold_total = total_out;
    a.fold_counter = draw_counter;
  }
  const int rc = launch_gs_head(a, host_pred_slice, host_bin_loss_weight, (hipStream_t)stream, &grid);
  if (rc != BGS_OK || !loss_out || a.ticket) return rc;
  hipLaunchKernelGGL(gs_head_reduce_kernel, dim3(1), dim3((unsigned)(BGS_WAVE * (a.B + 1))), 0, (hipStream_t)stream, a.partial,
                     grid, B, bbox_pred ? 1 : 0, box_loss_weight, avg_out, loss_out, total_out,
                     draw_counter);
  BGS_RETURN_LAUNCH_STATUS();
}

// class_bin_mask[c] = sum_b (label2binlabel[b][c] > 0) << b — the per-class foreground-bin table
// bgs_gs_head_step copies into LDS (built once per table; B <= 15).
namespace {
__global__ __launch_bounds__(256) void gs_class_bits_kernel(const int64_t* __restrict__ l2b, int C, int B,
                                                            uint16_t* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ((C + 7) & ~7)) return;
  if (c >= C) {            // padding entries (the table is read in 16-byte pieces)
    out[c] = 0;
    return;
  }
  unsigned bits = 0u;
  for (int b = 0; b < B; ++b)
    if (l2b[(size_t)b * C + c] > 0) bits |= 1u << b;
  out[c] = (uint16_t)bits;
}
}  // namespace

// Debug / profiling hook: when buf != NULL every later fused-head launch records 8 s_memtime marks per
// workgroup into buf[grid][8] (kernel start, loads landed, barrier 1, flags done, barrier 2, bins done,
// barrier 3, end); NULL switches it off (the default).  tools/gs_phase_times.py.
extern "C" void bgs_gs_head_debug_timestamps(unsigned long long* buf) { g_gs_tstamps = buf; }

extern "C" void bgs_gs_head_variant(int variant) {      // < 0: back to the default (BGS_GS_HEAD_VARIANT or automatic)
  g_head_variant = variant < 0 ? -2 : ((variant <= 5) ? variant : 0);
}

extern "C" int bgs_gs_head_variant_used(int N) { return head_variant_for(N); }

// 0 (default; < 0 restores it) = the separate gs_head_reduce_kernel launch | 1 = bgs_gs_head_step reduces its partials
// inside the main launch for N <= 2048 (the workgroup that arrives last; bitwise the two-launch result; 1 - 2 us slower)
extern "C" void bgs_gs_head_fold(int on) { g_head_fold = on < 0 ? 0 : (on ? 1 : 0); }

extern "C" void bgs_gs_head_tuning(int rows_per_workgroup) {
  g_head_rows = (rows_per_workgroup >= 0 && rows_per_workgroup <= 64) ? rows_per_workgroup : 0;
}

extern "C" int bgs_gs_class_bin_mask(const int64_t* label2binlabel, int C, int B, uint16_t* out,
                                     bgs_stream_t stream) {
  if (!label2binlabel || !out || C <= 0 || B <= 0 || ((uintptr_t)out & 15)) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS - 1) return BGS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gs_class_bits_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, label2binlabel, C, B, out);
  BGS_RETURN_LAUNCH_STATUS();
}

// Backward of bgs_gs_head_step: grad_terms [B + 1] (upstream gradient of {bins, box}; NULL = 0) and
// grad_total [1] (of the total; NULL = 0); scales dlogits per bin by grad_terms[b] + grad_total and
// dbbox_pred by grad_terms[B] + grad_total in place (one launch, early-out when every factor is 1).
extern "C" int bgs_gs_head_step_scale_grad(float* dlogits, float* dbbox_pred,
                                           const int64_t* host_pred_slice, const float* grad_terms,
                                           const float* grad_total, int N, int B, int W,
                                           int num_reg_classes, bgs_stream_t stream) {
  if (N < 0 || B <= 0 || W <= 0 || B > BGS_MAX_BINS - 1) return BGS_ERR_INVALID_ARG;
  if (N == 0 || (!dlogits && !dbbox_pred)) return BGS_OK;
  if (!host_pred_slice || (!grad_terms && !grad_total)) return BGS_ERR_INVALID_ARG;
  bgs::BinGeom geom;
  const int rc = bgs::make_bin_geom(host_pred_slice, B, W, &geom, nullptr);
  if (rc != BGS_OK) return rc;
  const size_t total = (size_t)N * (dbbox_pred ? (size_t)num_reg_classes * 4 : (size_t)W);
  size_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 512) blocks = 512;       // two workgroups per CU: the usual call early-outs
  bgs_internal_census_bump(BGS_CENSUS_GS_SCALE_GRAD);
  hipLaunchKernelGGL(gs_head_scale_grad_kernel, dim3((unsigned)blocks), dim3(kBlock),
                     sizeof(float) * (size_t)W, (hipStream_t)stream, dlogits, dbbox_pred, geom,
                     grad_terms, grad_total, N, B, W, num_reg_classes * 4);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_gs_loss_reduce(const void* workspace, int N, int B, float* loss_out,
                                  bgs_stream_t stream) {
  if (N < 0 || B <= 0 || B > BGS_MAX_BINS || !workspace || !loss_out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, N == 0 ? 1 : -1, B, loss_out, 1.0f);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_gs_scale_grad(float* dlogits, const int64_t* host_pred_slice, const float* g,
                                 int N, int B, int W, bgs_stream_t stream) {
  if (N < 0 || B <= 0 || W <= 0 || B > BGS_MAX_BINS) return BGS_ERR_INVALID_ARG;
  if (N == 0) return BGS_OK;
  if (!dlogits || !host_pred_slice || !g) return BGS_ERR_INVALID_ARG;
  bgs::BinGeom geom;
  const int rc = bgs::make_bin_geom(host_pred_slice, B, W, &geom, nullptr);
  if (rc != BGS_OK) return rc;
  const size_t total = (size_t)N * W;
  size_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gs_scale_grad_kernel, dim3((unsigned)blocks), dim3(kBlock),
                     sizeof(float) * (size_t)W, (hipStream_t)stream, dlogits, geom, g, N, B, W);
  BGS_RETURN_LAUNCH_STATUS();
}
