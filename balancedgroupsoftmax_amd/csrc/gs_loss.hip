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

#include "bgs_common.h"
#include "gs_rowwave.h"

namespace {

constexpr int kBlock = 256;   // generic fallback + helper kernels
constexpr int kWaves = kBlock / BGS_WAVE;
constexpr int kMaxGrid = 2048;

template <int VEC, bool WRITE_GRAD>
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

  int par = 0;
  for (int r = blockIdx.x; r < N; r += gridDim.x, par ^= 1) {
    float* row = smem + (size_t)par * wpad;
    bgs::stage_row<VEC>(logits + (size_t)r * W, row, W, tid, kBlock);
    int my_bl = 0;
    float my_coef = 0.f;
    if (lane < B) {
      my_bl = bin_labels[(size_t)lane * N + r];
      const float w = weights ? weights[(size_t)lane * N + r] : 1.f;
      my_coef = w * my_inv_avg;
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
      bgs::unstage_row<VEC>(row, dlogits + (size_t)r * W, W, tid, kBlock);
    }
  }
  // bin b lives in wave b % kWaves, lane b: no cross-wave reduction needed
  if (lane < B && (lane % kWaves) == wave)
    partial[(size_t)lane * gridDim.x + blockIdx.x] = lacc;
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
}

// loss[b] = sum_g partial[b, g] in a fixed order.  1024 threads: all loads of a bin are issued
// at once (one memory round trip per bin, the bins' loads overlap), then wave + LDS reduction.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial,
                                                               int G, int B,
                                                               float* __restrict__ out,
                                                               float scale) {
  __shared__ float sm[BGS_MAX_BINS][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
// Fused head kernel: _remap_labels + _sample_others (gs_prepare.hip) INSIDE the loss kernel, for
// N <= 4096 rows and no per-class reweighting — one launch instead of prepare -> boundary -> loss.
// Every workgroup derives what it needs itself:
//   prologue (all rows, 8 B per row from L2): per row a 16-bit flag word {real, foreground in bin
//   b} in LDS and the per-bin foreground counts -> k_b = int(n_fg * ratio), mode_b and, in closed
//   form, avg_b = max(sum_r w_b[r], 1) = n_real (all-ones modes) | n_fg + k_b (sampled) | 1;
//   per own row: w_b[r] = foreground, or "rank of (key(b, r), r) among the bin's background rows
//   < k_b" — the same exact-k, ties-by-row-index selection over the same counter-based keys as
//   gs_prepare_kernel's radix select, evaluated for ONE row by counting (N / 256 hashes per lane).
// Results are bitwise those of bgs_gs_prepare + bgs_gs_loss_fwd_bwd (tests/test_gpu_gs.py).
constexpr int kFusedMaxN = 4096;

template <int VEC, bool WRITE_GRAD>
__global__ __launch_bounds__(kBlock) void gs_head_fused_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const int64_t* __restrict__ l2b, const float* __restrict__ row_weights, bgs::BinGeom geom,
    int N, int C, int B, int W, int wpad, double ratio, uint64_t seed,
    const uint64_t* __restrict__ seed_offset, float* __restrict__ partial,
    float* __restrict__ dlogits, float* __restrict__ avg_out, int32_t* __restrict__ bl_out,
    float* __restrict__ w_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][wpad] rows (+ slack)
  __shared__ unsigned short sh_flags[kFusedMaxN];               // bit 15: real row, bit b: fg in bin b
  __shared__ int sh_cnt[BGS_MAX_BINS + 1];                      // n_fg per bin, [B] = n_real
  __shared__ int sh_rank[kWaves][BGS_MAX_BINS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = bgs::uniform(tid >> 6);
  if (seed_offset) seed += 0x2545F4914F6CDD1Dull * seed_offset[0];   // device-side draw counter

  // ---- prologue: flags of every row + per-bin foreground counts
  if (tid <= BGS_MAX_BINS) sh_cnt[tid] = 0;
  __syncthreads();
  for (int r = tid; r < N; r += kBlock) {
    int64_t y = labels[r];
    y = y < 0 ? 0 : (y >= C ? (int64_t)C - 1 : y);
    const bool real = !row_weights || row_weights[r] > 0.f;
    unsigned bits = 0u;
    for (int b = 0; b < B; ++b)
      if (l2b[(size_t)b * C + y] > 0) bits |= 1u << b;
    if (real) {
      atomicAdd(&sh_cnt[B], 1);
      for (unsigned m = bits; m; m &= m - 1) atomicAdd(&sh_cnt[__builtin_ctz(m)], 1);
    }
    sh_flags[r] = (unsigned short)(real ? (bits | 0x8000u) : 0u);
  }
  __syncthreads();
  // lane b < B of every wave: constants of bin b (gs_prepare_kernel's mode / k / avg)
  int my_mode = 0, my_k = 0;          // 0 = all zero, 1 = all one, 2 = sampled
  float my_inv_avg = 0.f;
  if (lane < B) {
    const int n_real = sh_cnt[B], n_fg = sh_cnt[lane], n_bg = n_real - n_fg;
    float total;
    if (lane == 0) {
      my_mode = 1;
      total = (float)n_real;
    } else if (n_fg == 0) {
      my_mode = 0;
      total = 0.f;
    } else {
      my_k = (int)((double)n_fg * ratio);
      my_mode = (my_k >= n_bg) ? 1 : 2;
      total = my_mode == 1 ? (float)n_real : (float)(n_fg + my_k);
    }
    const float a = fmaxf(total, 1.f);
    my_inv_avg = 1.f / a;
    if (blockIdx.x == 0 && wave == 0 && avg_out) avg_out[lane] = a;
  }
  float lacc = 0.f;

  int par = 0;
  for (int r = blockIdx.x; r < N; r += gridDim.x, par ^= 1) {
    float* row = smem + (size_t)par * wpad;
    bgs::stage_row<VEC>(logits + (size_t)r * W, row, W, tid, kBlock);
    const unsigned fr = sh_flags[r];
    // ---- sampling decision of this row in the sampled bins: rank among the bin's background rows
    for (int b = 1; b < B; ++b) {
      const int mode_b = __builtin_amdgcn_readlane(my_mode, b);
      const bool need = mode_b == 2 && (fr & 0x8000u) && !((fr >> b) & 1u);   // block-uniform
      int cnt = 0;
      if (need) {
        const unsigned mine = bgs::hash_u32(seed, (uint32_t)b, (uint32_t)r);
        for (int q = tid; q < N; q += kBlock) {
          const unsigned f = sh_flags[q];
          if ((f & 0x8000u) && !((f >> b) & 1u)) {
            const unsigned key = bgs::hash_u32(seed, (uint32_t)b, (uint32_t)q);
            cnt += (key < mine || (key == mine && q < r)) ? 1 : 0;
          }
        }
        cnt = bgs::wave_sum_i(cnt);
      }
      if (lane == 0) sh_rank[wave][b] = cnt;
    }
    int64_t yr = labels[r];
    yr = yr < 0 ? 0 : (yr >= C ? (int64_t)C - 1 : yr);
    int my_bl = 0;
    if (lane < B) my_bl = (int)l2b[(size_t)lane * C + yr];
    __syncthreads();                      // row staged + ranks published
    float my_coef = 0.f;
    if (lane < B) {
      float w = 0.f;
      if (fr & 0x8000u) {
        if (my_mode == 1) {
          w = 1.f;
        } else if (my_mode == 2) {
          if (my_bl > 0) {
            w = 1.f;
          } else {
            int rank = 0;
#pragma unroll
            for (int v = 0; v < kWaves; ++v) rank += sh_rank[v][lane];
            w = rank < my_k ? 1.f : 0.f;
          }
        }
      }
      my_coef = w * my_inv_avg;
      if (wave == 0) {
        if (bl_out) bl_out[(size_t)lane * N + r] = my_bl;
        if (w_out) w_out[(size_t)lane * N + r] = w;
      }
    }
    for (int b = wave; b < B; b += kWaves) {  // bins are independent: one wave each
      const int s = geom.start[b], n = geom.len[b];
      const float coef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_coef), b));
      const int tgt = min(max(__builtin_amdgcn_readlane(my_bl, b), 0), n - 1);
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
    __syncthreads();                      // gradient row complete; sh_rank free for the next row
    if (WRITE_GRAD) bgs::unstage_row<VEC>(row, dlogits + (size_t)r * W, W, tid, kBlock);
  }
  if (lane < B && (lane % kWaves) == wave)
    partial[(size_t)lane * gridDim.x + blockIdx.x] = lacc;
}

template <int VEC>
void launch_rowwave(bool grad, int grid, hipStream_t st, const float* logits, const int32_t* bl,
                    const float* w, const float* avg, const bgs::BinGeom& geom, int N, int B,
                    int W, float* partial, float* dlogits) {
  const int wpad = (W + 3) & ~3;
  // + slack: the register sweep reads up to 64*kSweep floats from a bin's start
  const size_t lds = sizeof(float) * (2 * (size_t)wpad + BGS_WAVE * bgs::kSweep);
  if (grad)
    hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, true>), dim3(grid), dim3(kBlock), lds, st,
                       logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);
  else
    hipLaunchKernelGGL((gs_loss_rowwave_kernel<VEC, false>), dim3(grid), dim3(kBlock), lds, st,
                       logits, bl, w, avg, geom, N, B, W, wpad, partial, dlogits);
}

// one workgroup per row; at most kMaxGrid workgroups (grid-stride beyond)
inline int loss_grid(int N) { return N <= 0 ? 1 : (N < kMaxGrid ? N : kMaxGrid); }

}  // namespace

extern "C" size_t bgs_gs_loss_workspace_bytes(int N, int B) {
  (void)N;
  (void)B;
  return (size_t)kMaxGrid * BGS_MAX_BINS * sizeof(float);
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
  const int grid = loss_grid(N);
  if (N == 0) {
    (void)hipMemsetAsync(partial, 0, sizeof(float) * B, st);
  } else {
    const uintptr_t al = (uintptr_t)logits | (uintptr_t)(dlogits ? dlogits : logits);
    int vec = 1;
    if (W % 4 == 0 && al % 16 == 0) vec = 4;
    else if (W % 2 == 0 && al % 8 == 0) vec = 2;
    // wave kernel: bins must tile [0, W) (true for tables built by tools/lvis_analyse.py) and
    // two staged rows must fit the 64 KB default LDS window
    if (!force_generic() && tiles && W <= 7936) {
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

// _remap_labels + _sample_others + loss forward + backward in ONE launch (+ the partial reduce):
// see gs_head_fused_kernel.  N <= 4096, bins must tile [0, W), no per-class reweighting (those
// cases use bgs_gs_prepare + bgs_gs_loss_fwd_bwd).  avg_out [B] is always written (the box loss
// normaliser reads bin 0's); bin_labels_out / weights_out [B, N] are optional (tests).
extern "C" int bgs_gs_head_loss_fused(const float* logits, const int64_t* labels,
                                      const int64_t* label2binlabel, const float* row_weights,
                                      const int64_t* host_pred_slice, int N, int C, int B, int W,
                                      double others_sample_ratio, uint64_t seed,
                                      const uint64_t* seed_offset, float* loss_out, float* dlogits,
                                      float* avg_out, int32_t* bin_labels_out, float* weights_out,
                                      void* workspace, bgs_stream_t stream) {
  if (N <= 0 || C <= 0 || B <= 0 || W <= 0) return BGS_ERR_INVALID_ARG;
  if (B > BGS_MAX_BINS - 1 || N > kFusedMaxN) return BGS_ERR_UNSUPPORTED;
  if (!logits || !labels || !label2binlabel || !host_pred_slice || !loss_out || !avg_out || !workspace)
    return BGS_ERR_INVALID_ARG;
  bgs::BinGeom geom;
  int tiles = 0;
  const int rc = bgs::make_bin_geom(host_pred_slice, B, W, &geom, &tiles);
  if (rc != BGS_OK) return rc;
  if (!tiles || W > 7936) return BGS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const int grid = loss_grid(N);
  const uintptr_t al = (uintptr_t)logits | (uintptr_t)(dlogits ? dlogits : logits);
  const int wpad = (W + 3) & ~3;
  const size_t lds = sizeof(float) * (2 * (size_t)wpad + BGS_WAVE * bgs::kSweep);
#define BGS_FUSED_LAUNCH(VEC_, GRAD_)                                                             \
  hipLaunchKernelGGL((gs_head_fused_kernel<VEC_, GRAD_>), dim3(grid), dim3(kBlock), lds, st, logits, \
                     labels, label2binlabel, row_weights, geom, N, C, B, W, wpad,                  \
                     others_sample_ratio, seed, seed_offset, partial, dlogits, avg_out,            \
                     bin_labels_out, weights_out)
  const bool grad = dlogits != nullptr;
  if (W % 4 == 0 && al % 16 == 0) { if (grad) BGS_FUSED_LAUNCH(4, true); else BGS_FUSED_LAUNCH(4, false); }
  else if (W % 2 == 0 && al % 8 == 0) { if (grad) BGS_FUSED_LAUNCH(2, true); else BGS_FUSED_LAUNCH(2, false); }
  else { if (grad) BGS_FUSED_LAUNCH(1, true); else BGS_FUSED_LAUNCH(1, false); }
#undef BGS_FUSED_LAUNCH
  if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partial, grid, B, loss_out,
                     1.0f);
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_gs_loss_reduce(const void* workspace, int N, int B, float* loss_out,
                                  bgs_stream_t stream) {
  if (N < 0 || B <= 0 || B > BGS_MAX_BINS || !workspace || !loss_out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, loss_grid(N), B, loss_out, 1.0f);
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
