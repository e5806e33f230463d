// Gradient clipping + SGD-with-momentum update of ALL trainable tensors in two launch phases (gfx950).
//
// The reference's optimizer hook (mmdet/core/utils/dist_utils.py:51-58: DistOptimizerHook.after_train_iter)
// runs  clip_grads(max_norm = 35, norm_type = 2)  ->  optimizer.step()  (torch.optim.SGD, momentum 0.9,
// weight decay 1e-4, configs/bags/*.py `optimizer`); Fp16OptimizerHook (mmdet/core/fp16/hooks.py:58-83)
// divides the gradients by the loss scale first.  As torch foreach ops that is ~21 launches for the 160
// trainable tensors of a `selectp = 0` iteration (per-tensor norms, stack, norm, clamp, scale, weight decay,
// momentum, update: 0.65 ms) and ~10 for the two tensors of the shipped `selectp = 1`.  Here:
//   phase 1  sgd_norm_kernel   one workgroup per 16K-element chunk: sum of squares of (grad * grad_scale)
//                              -> partials[chunk]
//   phase 2  sgd_apply_kernel  every workgroup sums ALL partials in the same fixed order (a few thousand
//                              floats from L2: deterministic and identical in every workgroup), forms
//                              coef = min(1, max_norm / (||g|| + 1e-6))  (torch.nn.utils.clip_grad_norm_), then
//                              g <- g * grad_scale * coef  (written back: the hook clips IN PLACE),
//                              d = g + weight_decay * p,  buf = momentum * buf + d,  p = p - lr * buf
//                              with torch's operation order and no fused multiply-adds.
// Tensors are passed BY VALUE in the kernel arguments, 64 per launch (no device-side pointer table, nothing
// to upload: capturable in a hipGraph as is); a `selectp = 0` step is 3 + 3 launches.
#include <stdlib.h>

#include "bgs_common.h"

namespace {

constexpr int kSgdTensors = 64;           // tensors per launch
constexpr int kSgdChunk = 16384;          // elements per workgroup
constexpr int kSgdBlock = 256;

struct SgdBatch {
  float* p[kSgdTensors];
  float* g[kSgdTensors];
  float* m[kSgdTensors];
  long long numel[kSgdTensors];
  int first_chunk[kSgdTensors + 1];       // chunk index (within this launch) of each tensor's first chunk
  int n;
  int chunk_base;                         // global index of this launch's chunk 0 (partials)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int sgd_find_tensor(const SgdBatch& b, int chunk) {
  int lo = 0, hi = b.n - 1;               // largest t with first_chunk[t] <= chunk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.first_chunk[mid] <= chunk) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ float sgd_block_sum(float v, float* red) {
  v = bgs::wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                        // red may still be read from a previous call
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kSgdBlock / BGS_WAVE; ++w) s += red[w];
  return s;
}

__global__ __launch_bounds__(kSgdBlock) void sgd_norm_kernel(SgdBatch b, float grad_scale,
                                                             float* __restrict__ partials) {
  __shared__ float red[kSgdBlock / BGS_WAVE];
  const int chunk = blockIdx.x;
  const int t = sgd_find_tensor(b, chunk);
  const long long off = (long long)(chunk - b.first_chunk[t]) * kSgdChunk;
  const long long n = b.numel[t] - off < kSgdChunk ? b.numel[t] - off : kSgdChunk;
  const float* g = b.g[t] + off;
  float acc = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    const long long n4 = n >> 2;
    for (long long i = threadIdx.x; i < n4; i += kSgdBlock) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4) * grad_scale;
      acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kSgdBlock) {
      const float v = g[i] * grad_scale;
      acc += v * v;
    }
  } else {
    for (long long i = threadIdx.x; i < n; i += kSgdBlock) {
      const float v = g[i] * grad_scale;
      acc += v * v;
    }
  }
  const float s = sgd_block_sum(acc, red);
  if (threadIdx.x == 0) partials[b.chunk_base + chunk] = s;
}

__global__ __launch_bounds__(kSgdBlock) void sgd_apply_kernel(SgdBatch b, const float* __restrict__ partials,
                                                              int total_chunks, float max_norm,
                                                              float grad_scale, float lr, float momentum,
                                                              float weight_decay,
                                                              float* __restrict__ total_norm_out) {
  __shared__ float red[kSgdBlock / BGS_WAVE];
  float coef = grad_scale;
  if (max_norm > 0.f || total_norm_out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < total_chunks; i += kSgdBlock) acc += partials[i];
    const float total = sqrtf(sgd_block_sum(acc, red));
    if (total_norm_out && blockIdx.x == 0 && b.chunk_base == 0 && threadIdx.x == 0) total_norm_out[0] = total;
    if (max_norm > 0.f) {
      const float c = max_norm / (total + 1e-6f);
      coef = __fmul_rn(grad_scale, c < 1.f ? c : 1.f);
    }
  }
  const int chunk = blockIdx.x;
  const int t = sgd_find_tensor(b, chunk);
  const long long off = (long long)(chunk - b.first_chunk[t]) * kSgdChunk;
  const long long n = b.numel[t] - off < kSgdChunk ? b.numel[t] - off : kSgdChunk;
  float* p = b.p[t] + off;
  float* g = b.g[t] + off;
  float* m = b.m[t] + off;
  const bool scale_g = coef != 1.f;
  // torch.optim.SGD (_single_tensor_sgd): d = g + wd * p; buf = buf * momentum + d; p = p + (-lr) * buf —
  // separate roundings (no contraction into fused multiply-adds)
#define BGS_SGD_ONE(P_, G_, M_)                                      \
  do {                                                               \
    float gg = (G_);                                                 \
    if (scale_g) gg = __fmul_rn(gg, coef);                           \
    float d = gg;                                                    \
    if (weight_decay != 0.f) d = __fadd_rn(gg, __fmul_rn(weight_decay, (P_)));   \
    float bf = d;                                                    \
    if (momentum != 0.f) bf = __fadd_rn(__fmul_rn((M_), momentum), d);           \
    (G_) = gg;                                                       \
    (M_) = bf;                                                       \
    (P_) = __fadd_rn((P_), __fmul_rn(-lr, bf));                      \
  } while (0)
  if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m)) & 15) == 0) {
    const long long n4 = n >> 2;
    for (long long i = threadIdx.x; i < n4; i += kSgdBlock) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
      f32x4 gv = *reinterpret_cast<f32x4*>(g + i * 4);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) BGS_SGD_ONE(pv[u], gv[u], mv[u]);
      *reinterpret_cast<f32x4*>(p + i * 4) = pv;
      if (scale_g) *reinterpret_cast<f32x4*>(g + i * 4) = gv;
      *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    }
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kSgdBlock) BGS_SGD_ONE(p[i], g[i], m[i]);
  } else {
    for (long long i = threadIdx.x; i < n; i += kSgdBlock) BGS_SGD_ONE(p[i], g[i], m[i]);
  }
#undef BGS_SGD_ONE
}

inline long long sgd_chunks(long long numel) { return (numel + kSgdChunk - 1) / kSgdChunk; }

}  // namespace

extern "C" size_t bgs_sgd_clip_workspace_bytes(const long long* host_numel, int n_tensors) {
  if (!host_numel || n_tensors <= 0) return 0;
  long long c = 0;
  for (int i = 0; i < n_tensors; ++i) c += host_numel[i] > 0 ? sgd_chunks(host_numel[i]) : 0;
  return (size_t)c * sizeof(float) + 16;
}

// host_* : HOST arrays of n_tensors device pointers / element counts (fp32 tensors; momentum buffers of the
// same shapes, zero-initialised before the first step: momentum * 0 + d == torch's first-step clone of d).
// max_norm <= 0: no clipping.  grad_scale: multiplied into every gradient first (1 / loss_scale; 1 = none).
// total_norm_out: device float[1] (the L2 norm of the scaled, unclipped gradients) or NULL.
extern "C" int bgs_sgd_clip_step(const void* const* host_params, const void* const* host_grads,
                                 const void* const* host_momentum, const long long* host_numel,
                                 int n_tensors, float max_norm, float grad_scale, float lr, float momentum,
                                 float weight_decay, void* workspace, size_t workspace_bytes,
                                 float* total_norm_out, bgs_stream_t stream) {
  if (n_tensors < 0) return BGS_ERR_INVALID_ARG;
  if (n_tensors == 0) return BGS_OK;
  if (!host_params || !host_grads || !host_momentum || !host_numel || !workspace) return BGS_ERR_INVALID_ARG;
  if (workspace_bytes < bgs_sgd_clip_workspace_bytes(host_numel, n_tensors)) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)workspace % 4 != 0) return BGS_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* partials = reinterpret_cast<float*>(workspace);
  long long total_chunks = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (host_numel[i] < 0 || (host_numel[i] > 0 && (!host_params[i] || !host_grads[i] || !host_momentum[i])))
      return BGS_ERR_INVALID_ARG;
    total_chunks += sgd_chunks(host_numel[i]);
  }
  if (total_chunks > 0x3fffffffLL) return BGS_ERR_UNSUPPORTED;
  const bool need_norm = max_norm > 0.f || total_norm_out != nullptr;
  for (int phase = need_norm ? 0 : 1; phase < 2; ++phase) {
    int base = 0;
    for (int t0 = 0; t0 < n_tensors;) {
      SgdBatch b;
      b.n = 0;
      b.chunk_base = base;
      int chunks = 0;
      while (t0 < n_tensors && b.n < kSgdTensors) {
        if (host_numel[t0] > 0) {
          b.p[b.n] = reinterpret_cast<float*>(const_cast<void*>(host_params[t0]));
          b.g[b.n] = reinterpret_cast<float*>(const_cast<void*>(host_grads[t0]));
          b.m[b.n] = reinterpret_cast<float*>(const_cast<void*>(host_momentum[t0]));
          b.numel[b.n] = host_numel[t0];
          b.first_chunk[b.n] = chunks;
          chunks += (int)sgd_chunks(host_numel[t0]);
          ++b.n;
        }
        ++t0;
      }
      if (!b.n) continue;
      b.first_chunk[b.n] = chunks;
      for (int i = b.n; i < kSgdTensors; ++i) {
        b.p[i] = b.g[i] = b.m[i] = nullptr;
        b.numel[i] = 0;
        b.first_chunk[i + 1] = chunks;
      }
      if (phase == 0)
        hipLaunchKernelGGL(sgd_norm_kernel, dim3((unsigned)chunks), dim3(kSgdBlock), 0, st, b, grad_scale,
                           partials);
      else
        hipLaunchKernelGGL(sgd_apply_kernel, dim3((unsigned)chunks), dim3(kSgdBlock), 0, st, b, partials,
                           (int)total_chunks, need_norm ? max_norm : 0.f, grad_scale, lr, momentum,
                           weight_decay, total_norm_out);
      if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
      base += chunks;
    }
  }
  return BGS_OK;
}
