/*
 * bgs.h — C ABI of libbgs.so: MI355X (gfx950) kernels for the Balanced Group Softmax
 * detection hot path.
 *
 * Conventions (mirroring the ownership rules of the reference's native ops,
 * mmdet/ops/roi_align/roi_align.py:23-26 — the CALLER allocates every output):
 *   - all pointers are DEVICE pointers unless named host_*; plain C types only;
 *   - no allocation, no synchronisation, no host<->device copies inside any call;
 *     every kernel is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream, which is what the reference's ops use, roi_align_kernel.cu:137);
 *   - return value: 0 = enqueued OK, otherwise a BGS_ERR_* code (never throws/aborts);
 *   - thread-safe for distinct streams and distinct workspaces.
 *
 * Each entry point cites the reference code it replaces (paths relative to the
 * reference repo root).
 */
#ifndef BGS_H_
#define BGS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGS_OK 0
#define BGS_ERR_INVALID_ARG 1   /* null pointer / negative size / misaligned buffer      */
#define BGS_ERR_UNSUPPORTED 2   /* shape outside what the kernels were built for          */
#define BGS_ERR_LAUNCH 3        /* hipGetLastError() != hipSuccess after the launch       */

#define BGS_MAX_BINS 16         /* the reference ships 3-, 5- and 9-bin tables            */

typedef void* bgs_stream_t;     /* hipStream_t */

/* Library version (major*10000 + minor*100 + patch) and error text. */
int bgs_version(void);
const char* bgs_error_string(int code);

/* Device self-test of the wave64 reduction primitive: in [64] float, out [4] float =
 * {max, sum} by the build's primitive (DPP) followed by {max, sum} by a ds_bpermute butterfly. */
int bgs_selftest_wave_reduce(const float* in, float* out, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Group-softmax label remap + "others" sampling.
 * Replaces GSBBoxHeadWith0._remap_labels / _sample_others
 *   (mmdet/models/bbox_heads/gs_bbox_head_with0.py:91-112, :63-89) and the Reweight
 *   variant (gs_bbox_head_with0_reweight.py:57-87).
 *
 *   labels          [N]    int64, 0 = background, < C
 *   label2binlabel  [B,C]  int64 (label2binlabel.pt)
 *   cls_weight      [B-1, cls_weight_stride] float or NULL: per-bin class weights indexed
 *                   by bin label (bins_cls_weight.pkl), row i-1 for bin i
 *   others_sample_ratio    keep all in-bin foreground rows + int(n_fg*ratio) others,
 *                   drawn uniformly WITHOUT replacement (counter-based RNG keyed by
 *                   (seed, bin, row); same (seed) => same draw; no host RNG, no sync)
 *   seed_offset     device uint64 [1] or NULL: a draw counter the caller bumps on the device
 *                   (e.g. inside a captured hipGraph, where `seed` itself is frozen)
 *   bin_labels_out  [B,N]  int32 or NULL: label2binlabel[b, labels[r]] (the integer gather,
 *                   bit-exact); consumed by bgs_gs_loss_fwd_bwd
 *   weights_out     [B,N]  float
 *   avg_out         [B]    float  = max(sum_r weights[b,r], 1)
 * ---------------------------------------------------------------------------------- */
int bgs_gs_prepare(const int64_t* labels, const int64_t* label2binlabel,
                   const float* cls_weight, int cls_weight_stride,
                   int N, int C, int B, double others_sample_ratio, uint64_t seed,
                   const uint64_t* seed_offset, int32_t* bin_labels_out, float* weights_out, float* avg_out,
                   bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused group-softmax loss forward + backward.
 * Replaces the B-iteration loop of GSBBoxHeadWith0.loss (gs_bbox_head_with0.py:160-171)
 *   = _slice_preds (:134-145) + CrossEntropyLoss.forward (losses/cross_entropy_loss.py:86-103)
 *   + cross_entropy (:9-19) + weight_reduce_loss (losses/utils.py:26-52) and the autograd
 *   backward of all of it, in ONE pass over the logits:
 *     loss[b]      = sum_r w[b,r] * (logsumexp(z[r, s_b:s_b+n_b]) - z[r, s_b + bl[b,r]]) / avg[b]
 *     dlogits[r,s_b+j] = (w[b,r]/avg[b]) * (softmax_j(z[r, s_b:s_b+n_b]) - [j == bl[b,r]])
 *
 *   logits      [N,W]  float (row stride W)
 *   bin_labels  [B,N]  int32 = label2binlabel[b, labels[r]] (output of bgs_gs_prepare)
 *   host_pred_slice [B,2] int64 (start, length) ON THE HOST (pred_slice_with0.pt); it is static
 *               metadata and travels by value in the kernel arguments.  Bins must not overlap.
 *   weights     [B,N]  float or NULL (= all ones);  avg [B] float or NULL (= max(N,1))
 *   loss_out    [B]    float, or NULL: stop after the streaming kernel and leave the per-
 *               workgroup partial sums in the workspace (bgs_gs_loss_reduce finishes the job)
 *   dlogits     [N,W]  float or NULL (forward only); columns outside every bin get 0
 *   workspace   bgs_gs_loss_workspace_bytes(N, B) bytes of scratch (per-block partial sums;
 *               reduced in a fixed order => bitwise reproducible, no atomics)
 * ---------------------------------------------------------------------------------- */
size_t bgs_gs_loss_workspace_bytes(int N, int B);
int bgs_gs_loss_fwd_bwd(const float* logits, const int32_t* bin_labels,
                        const int64_t* host_pred_slice,
                        const float* weights, const float* avg,
                        int N, int B, int W,
                        float* loss_out, float* dlogits, void* workspace,
                        bgs_stream_t stream);

/* Second phase of bgs_gs_loss_fwd_bwd(loss_out = NULL): loss_out[b] = sum of the partials. */
int bgs_gs_loss_reduce(const void* workspace, int N, int B, float* loss_out, bgs_stream_t stream);

/* In-place scaling of the stored gradient by the B upstream scalars
 * (d total / d loss_b; 1 for plain Faster R-CNN, stage_loss_weights for Cascade,
 * mmdet/models/detectors/cascade_rcnn.py:248-250):  dlogits[:, bin b] *= g[b].
 * Early-outs on the device when every g[b] == 1.  g [B] float (device). */
int bgs_gs_scale_grad(float* dlogits, const int64_t* host_pred_slice, const float* g,
                      int N, int B, int W, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Inference score merge.  Replaces GSBBoxHeadWith0._merge_score
 *   (gs_bbox_head_with0.py:239-273): softmax inside every bin, then
 *     scores[r,0] = p_0[r,0];  scores[r,c] = p_0[r,1] * p_b[r, k]  for the column
 *     cls2col[c] = s_b + k of class c (k >= 1);  cls2col[c] < 0  => scores[r,c] = 0.
 *   logits [N,W] float, host_pred_slice [B,2] int64 (host), cls2col [C] int32,
 *   scores_out [N,C] float.
 * ---------------------------------------------------------------------------------- */
int bgs_gs_merge_score(const float* logits, const int64_t* host_pred_slice,
                       const int32_t* cls2col,
                       int N, int C, int B, int W, float* scores_out, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Box-regression loss forward + backward.  Replaces the loss_bbox branch of
 *   GSBBoxHeadWith0.loss / BBoxHead.loss (gs_bbox_head_with0.py:173-185,
 *   bbox_head.py:117-129) with smooth_l1_loss (losses/smooth_l1_loss.py:9-15):
 *     pos = labels > 0;  d = bbox_pred.view(N,R,4)[pos, labels[pos]] - targets[pos]
 *     loss = loss_weight * sum(smooth_l1(d; beta) * bbox_weights[pos]) / avg_factor
 *   bbox_pred [N, 4*R] float; R = num_classes, or 1 when reg_class_agnostic
 *   loss_out [1] float; dbbox_pred [N,4*R] float or NULL (dense, zeros off the gathered
 *   slots, exactly like the autograd result);  workspace: bgs_bbox_loss_workspace_bytes(N).
 * ---------------------------------------------------------------------------------- */
size_t bgs_bbox_loss_workspace_bytes(int N);
int bgs_bbox_smooth_l1_fwd_bwd(const float* bbox_pred, const int64_t* labels,
                               const float* bbox_targets, const float* bbox_weights,
                               int N, int R, float beta, float avg_factor, float loss_weight,
                               float* loss_out, float* dbbox_pred, void* workspace,
                               bgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BGS_H_ */
