// Box-regression loss (SmoothL1 on the class-specific deltas of positive RoIs), forward +
// backward, for gfx950.
//
// Replaces the loss_bbox branch of GSBBoxHeadWith0.loss / BBoxHead.loss
// (mmdet/models/bbox_heads/gs_bbox_head_with0.py:173-185, bbox_head.py:117-129):
//     pos_inds = labels > 0                                  (boolean-mask indexing: host sync)
//     pos_bbox_pred = bbox_pred.view(N, -1, 4)[pos_inds, labels[pos_inds]]
//     loss = SmoothL1Loss(pos_bbox_pred, bbox_targets[pos_inds], bbox_weights[pos_inds],
//                         avg_factor=N)                       (losses/smooth_l1_loss.py:9-45)
// plus its autograd backward, which materialises a dense zero [N, 4R] gradient (20 MB for
// R = 1231) and scatters 4 values per positive row into it.
//
// Here: one float4 per (row, class) slot — the 4 deltas of a class are 16 contiguous,
// 16-byte-aligned bytes — so the dense gradient is produced by a single streaming pass of
// coalesced global_store_dwordx4 (zeros everywhere except slot (r, labels[r])), and the loss
// terms are computed by the thread that owns that slot.  Algorithmic bytes: 16*R per row
// written (+64 B read for positives).  No mask compaction, no host sync.
#include "bgs_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kMaxGrid = 4096;

__device__ __forceinline__ float sl1(float d, float beta, float& grad) {
  const float ad = fabsf(d);
  if (ad < beta) {
    grad = d / beta;
    return 0.5f * ad * ad / beta;
  }
  grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return ad - 0.5f * beta;
}

// WRITE_GRAD: threads sweep all N*R float4 slots.  Otherwise only N threads (one per row).
template <bool WRITE_GRAD>
__global__ __launch_bounds__(kBlock) void bbox_sl1_kernel(
    const float* __restrict__ bbox_pred, const int64_t* __restrict__ labels,
    const float* __restrict__ targets, const float* __restrict__ bweights, int N, int R,
    float beta, float scale, float* __restrict__ partial, float* __restrict__ dpred) {
  float acc = 0.f;
  const size_t total = WRITE_GRAD ? (size_t)N * R : (size_t)N;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const int r = WRITE_GRAD ? (int)(i / (size_t)R) : (int)i;
    const int c = WRITE_GRAD ? (int)(i % (size_t)R) : -1;
    const int64_t y = labels[r];
    // slot of this row's positive class (agnostic regression: slot 0)
    const int64_t slot = (R == 1) ? 0 : y;
    const bool pos = (y > 0) && (slot < R);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (pos && (!WRITE_GRAD || c == (int)slot)) {
      const f32x4 p =
          *reinterpret_cast<const f32x4*>(bbox_pred + ((size_t)r * R + (size_t)slot) * 4);
      const f32x4 t = *reinterpret_cast<const f32x4*>(targets + (size_t)r * 4);
      const f32x4 w = *reinterpret_cast<const f32x4*>(bweights + (size_t)r * 4);
      float gx, gy, gz, gw;
      acc += sl1(p.x - t.x, beta, gx) * w.x;
      acc += sl1(p.y - t.y, beta, gy) * w.y;
      acc += sl1(p.z - t.z, beta, gz) * w.z;
      acc += sl1(p.w - t.w, beta, gw) * w.w;
      g = f32x4{gx * w.x * scale, gy * w.y * scale, gz * w.z * scale, gw * w.w * scale};
    }
    if (WRITE_GRAD) *reinterpret_cast<f32x4*>(dpred + i * 4) = g;
  }
  // block reduce (fixed order) -> partial[blockIdx.x]
  __shared__ float sm[kBlock / BGS_WAVE];
  acc = bgs::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kBlock / BGS_WAVE; ++w) s += sm[w];
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void bbox_reduce_kernel(const float* __restrict__ partial,
                                                             int G, float scale,
                                                             float* __restrict__ out) {
  __shared__ float sm[kBlock / BGS_WAVE];
  float s = 0.f;
  for (int g = threadIdx.x; g < G; g += kBlock) s += partial[g];
  s = bgs::wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / BGS_WAVE; ++w) t += sm[w];
    out[0] = t * scale;
  }
}

}  // namespace

extern "C" size_t bgs_bbox_loss_workspace_bytes(int N) {
  (void)N;
  return (size_t)kMaxGrid * sizeof(float);
}

extern "C" int bgs_bbox_smooth_l1_fwd_bwd(const float* bbox_pred, const int64_t* labels,
                                          const float* bbox_targets, const float* bbox_weights,
                                          int N, int R, float beta, float avg_factor,
                                          float loss_weight, float* loss_out, float* dbbox_pred,
                                          void* workspace, bgs_stream_t stream) {
  if (N < 0 || R <= 0 || !(beta > 0.f) || !(avg_factor > 0.f)) return BGS_ERR_INVALID_ARG;
  if (!loss_out || !workspace) return BGS_ERR_INVALID_ARG;
  if (N > 0 && (!bbox_pred || !labels || !bbox_targets || !bbox_weights))
    return BGS_ERR_INVALID_ARG;
  if (((uintptr_t)bbox_pred | (uintptr_t)bbox_targets | (uintptr_t)bbox_weights |
       (uintptr_t)dbbox_pred) % 16 != 0)
    return BGS_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const float scale = loss_weight / avg_factor;
  const size_t total = dbbox_pred ? (size_t)N * R : (size_t)N;
  size_t grid = (total + kBlock - 1) / kBlock;
  if (grid < 1) grid = 1;
  if (grid > (size_t)kMaxGrid) grid = kMaxGrid;
  if (dbbox_pred)
    hipLaunchKernelGGL((bbox_sl1_kernel<true>), dim3((unsigned)grid), dim3(kBlock), 0, st,
                       bbox_pred, labels, bbox_targets, bbox_weights, N, R, beta, scale, partial,
                       dbbox_pred);
  else
    hipLaunchKernelGGL((bbox_sl1_kernel<false>), dim3((unsigned)grid), dim3(kBlock), 0, st,
                       bbox_pred, labels, bbox_targets, bbox_weights, N, R, beta, scale, partial,
                       dbbox_pred);
  hipLaunchKernelGGL(bbox_reduce_kernel, dim3(1), dim3(kBlock), 0, st, partial, (int)grid, scale,
                     loss_out);
  BGS_RETURN_LAUNCH_STATUS();
}
