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
 *   row_weights     [N] float or NULL: the detector's label_weights; rows with a value <= 0 are
 *                   padding slots of a fixed-shape batch (the reference's sampler returns fewer
 *                   RoIs instead, two_stage.py:200-210) and are excluded from everything: not
 *                   counted, never drawn, weight 0 in every bin
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
                   const float* cls_weight, int cls_weight_stride, const float* row_weights,
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

/* The head's classification loss in ONE streaming launch (+ the partial reduce): _remap_labels
 * (gs_bbox_head_with0.py:91-112) and _sample_others (:63-89) are evaluated inside the loss kernel —
 * every workgroup derives the per-bin counts from the 8-byte labels itself and decides its own row's
 * sample weights by ranking the row's counter-based key among the bin's background rows (the same
 * exact-k, ties-by-row selection as bgs_gs_prepare; bitwise-equal results).  N <= 4096, B <= 15,
 * bins tiling [0, W), no per-class reweighting (otherwise BGS_ERR_UNSUPPORTED: use bgs_gs_prepare +
 * bgs_gs_loss_fwd_bwd).  labels [N] i64, label2binlabel [B,C] i64, row_weights [N] or NULL (<= 0:
 * padding slot), host_pred_slice [B,2] HOST; loss_out [B], dlogits [N,W] or NULL, avg_out [B]
 * (written: max(sum_r w_b[r], 1)), bin_labels_out / weights_out [B,N] or NULL; workspace >=
 * bgs_gs_loss_workspace_bytes(N, B). */
int bgs_gs_head_loss_fused(const float* logits, const int64_t* labels, const int64_t* label2binlabel,
                           const float* row_weights, const int64_t* host_pred_slice, int N, int C,
                           int B, int W, double others_sample_ratio, uint64_t seed,
                           const uint64_t* seed_offset, float* loss_out, float* dlogits,
                           float* avg_out, int32_t* bin_labels_out, float* weights_out,
                           void* workspace, bgs_stream_t stream);
/* The WHOLE GSBBoxHeadWith0.loss() (gs_bbox_head_with0.py:147-186: _remap_labels :91-112,
 * _sample_others :63-89, the per-bin CrossEntropyLoss terms :160-171 and the SmoothL1 box branch
 * :173-185, plus the sum parse_losses forms, mmdet/apis/train.py:24-47) as TWO launches:
 *   main kernel = bgs_gs_head_loss_fused's streaming kernel, with the per-bin loss weights
 *                 (host_bin_loss_weight [B] HOST or NULL = 1) folded into losses and gradient and
 *                 the box branch of every row evaluated by the row's workgroup: SmoothL1(beta) of
 *                 bbox_pred [N, 4*num_reg_classes] at the row's own class slot (class-agnostic:
 *                 num_reg_classes = 1) against bbox_targets / bbox_weights [N,4], normalised by
 *                 box_loss_weight / max(#real rows, 1); dbbox_pred = its dense [N, 4R] gradient
 *                 (NULL: not wanted — the shipped selectp=1 mode);
 *   reduce      = loss_out[0..B-1] per-bin losses, loss_out[B] = loss_bbox (0 when bbox_pred is
 *                 NULL), total_out[0] (or NULL) = their sum (fixed order: bitwise reproducible); advances
 *                 *draw_counter (device uint64 or NULL), which the main kernel read as the index of
 *                 this call's "others" draw — hipGraph replays draw fresh samples with no extra launch.
 * loss_out == NULL: main kernel only (profiling hook).  Limits of bgs_gs_head_loss_fused;
 * BGS_ERR_UNSUPPORTED also when two rows + the flag words exceed the 64 KB LDS window. */
int bgs_gs_head_step(const float* logits, const int64_t* labels, const int64_t* label2binlabel,
                     const uint16_t* class_bin_mask, const float* row_weights,
                     const int64_t* host_pred_slice, const float* host_bin_loss_weight, int N, int C,
                     int B, int W, double others_sample_ratio, uint64_t seed, uint64_t* draw_counter,
                     const float* bbox_pred, const float* bbox_targets, const float* bbox_weights,
                     int num_reg_classes, float beta, float box_loss_weight, float* loss_out,
                     float* total_out, float* dlogits, float* dbbox_pred, float* avg_out,
                     int32_t* bin_labels_out, float* weights_out, void* workspace,
                     bgs_stream_t stream);
/* class_bin_mask: uint16 (device), ((C + 7) & ~7) entries, 16-byte aligned: entry c < C has bit b set
 * iff class c is foreground in bin b (label2binlabel[b][c] > 0), the padding entries are 0.  Built
 * once per table; bgs_gs_head_step copies it into LDS instead of gathering the [B,C] int64 table per
 * row (NULL there: every workgroup derives it from label2binlabel, slower). */
int bgs_gs_class_bin_mask(const int64_t* label2binlabel, int C, int B, uint16_t* out,
                          bgs_stream_t stream);
/* Backward of bgs_gs_head_step: grad_terms [B+1] (device; upstream gradient of {bins, box}, NULL = 0)
 * and grad_total [1] (of total_out, NULL = 0): dlogits[:, bin b] *= grad_terms[b] + grad_total,
 * dbbox_pred *= grad_terms[B] + grad_total, in place, one launch, early-out on the device when
 * every factor is 1. */
int bgs_gs_head_step_scale_grad(float* dlogits, float* dbbox_pred, const int64_t* host_pred_slice,
                                const float* grad_terms, const float* grad_total, int N, int B, int W,
                                int num_reg_classes, bgs_stream_t stream);
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

/* ------------------------------------------------------------------------------------
 * Convolution / linear layer as an implicit GEMM on the fp32 matrix cores.
 * Replaces what the reference delegates to cuDNN/cuBLAS via nn.Conv2d (+ eval-mode
 * BatchNorm2d folded into w/bias, + ReLU, + residual add) and nn.Linear:
 *   mmdet/models/backbones/resnet.py:220-266, mmdet/models/necks/fpn.py:101-141,
 *   mmdet/models/anchor_heads/rpn_head.py:30-35, mmdet/models/bbox_heads/convfc_bbox_head.py:132-168.
 *     y[n,ho,wo,j] = act( sum_{r,s,c} x[n, ho*stride-pad+r, wo*stride-pad+s, c] * w[j,r,s,c]
 *                         + bias[j] + residual )
 *   x [N,H,W,Cin] NHWC float (Cin % 4 == 0), w [Cout,R,S,Cin] float, bias [Cout] or NULL,
 *   y [N,Ho,Wo,Cout] with Ho = (H + 2*pad - R)/stride + 1.
 *   residual_mode 0: none; 1: residual [N,Ho,Wo,Cout] (bottleneck identity);
 *                 2: residual [N,Ho/2,Wo/2,Cout] read with nearest-2x upsampling (FPN top-down,
 *                    fpn.py:117-121).   relu != 0 applies max(.,0) last.
 *   A linear layer is the H = W = R = S = 1 case: x [M,K], w [Cout,K].
 *   Arithmetic: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, exact fma chain).
 * ---------------------------------------------------------------------------------- */
int bgs_conv2d_nhwc_f32(const float* x, const float* w, const float* bias, const float* residual,
                        float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                        int stride, int pad, int relu, int residual_mode, bgs_stream_t stream);

/* Same two operations with a caller-provided scratch buffer, which enables the split-K path for
 * layers whose output is too small to fill the 256 CUs (M*Cout/4096 < ~600 workgroups: ResNet
 * layer3/4, the top FPN/RPN levels, the FC heads): gridDim.z slices of the reduction write raw
 * partial sums to the scratch, a second launch sums them in a fixed order and applies the
 * epilogue.  workspace may be NULL / 0 bytes (no split); bgs_conv2d_workspace_bytes(M, Cout) with
 * M = N*Ho*Wo (output pixels; for the data gradient N*H*W and Cin) is always sufficient. */
size_t bgs_conv2d_workspace_bytes(long long M, int Cout);
int bgs_conv2d_nhwc_f32_ws(const float* x, const float* w, const float* bias, const float* residual,
                           float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                           int stride, int pad, int relu, int residual_mode, void* workspace,
                           size_t workspace_bytes, bgs_stream_t stream);
int bgs_conv2d_dgrad_nhwc_f32_ws(const float* dy, const float* wt, const float* residual,
                                 const float* mask, float* dx, int N, int H, int W, int Cin,
                                 int Cout, int R, int S, int stride, int pad, int residual_mode,
                                 void* workspace, size_t workspace_bytes, bgs_stream_t stream);

/* Backward of bgs_conv2d_nhwc_f32 for the `selectp = 0` mode (train everything; the reference
 * gets these from cuDNN backward-data / backward-filter through autograd of the nn.Conv2d /
 * nn.Linear call sites listed above; tools/train.py:49-57 selects what trains).
 *
 * bgs_conv2d_dgrad_nhwc_f32: dx[n,h,w,ci] = sum_{r,s,co} dy[n,ho,wo,co] * W[co,r,s,ci] over the
 *   (ho,wo,r,s) with ho*stride-pad+r == h, wo*stride-pad+s == w.  stride 1 or 2, R == S.
 *   dy [N,Ho,Wo,Cout] (Cout % 4 == 0); wt = the filter re-laid-out by the caller as
 *   wt[ci][R-1-r][S-1-s][co] (so that the kernel streams it K-major); dx [N,H,W,Cin].
 *   Epilogue: dx += residual (mode 1: [N,H,W,Cin]; mode 3: a [N,2H,2W,Cin] map summed over 2x2
 *   blocks = backward of the FPN nearest-2x top-down add, fpn.py:117-121), then, when mask !=
 *   NULL, dx = mask[n,h,w,ci] > 0 ? dx : 0 (backward of the ReLU that produced the conv input).
 *
 * bgs_conv2d_wgrad_nhwc_f32: dw[co,r,s,ci] (+)= sum_m dy[m,co] * x[n, ho*stride-pad+r,
 *   wo*stride-pad+s, ci];  db[co] (+)= sum_m dy[m,co] when db != NULL.  Cin % 4 == 0,
 *   Cout % 4 == 0.  The reduction over m is split across workgroups; partials are summed in a
 *   fixed order (bitwise reproducible).  accumulate != 0 adds to the existing dw/db (weights
 *   shared by the five RPN levels).  workspace >= bgs_conv2d_wgrad_workspace_bytes(...).
 * ---------------------------------------------------------------------------------- */
int bgs_conv2d_dgrad_nhwc_f32(const float* dy, const float* wt, const float* residual,
                              const float* mask, float* dx, int N, int H, int W, int Cin,
                              int Cout, int R, int S, int stride, int pad, int residual_mode,
                              bgs_stream_t stream);
size_t bgs_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R, int S,
                                        int stride, int pad);
int bgs_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, float* db, int N, int H,
                              int W, int Cin, int Cout, int R, int S, int stride, int pad,
                              int accumulate, void* workspace, bgs_stream_t stream);
/* The same weight gradient on the bf16 matrix cores (csrc/conv_wgrad.hip, conv_wgrad_bfx_kernel):
 * planes = 3: fp32-faithful "bf16x6" (every product from the exact three-way bf16 split of BOTH
 * dy and x, fp32 accumulate; error vs fp64 not above the fp32-MFMA kernel's), planes = 1: operands
 * rounded to bf16 (the bf16 mode of cfg[4]).  Same layout, split-M reduction in a fixed order and
 * workspace contract (>= bgs_conv2d_wgrad_bfx_workspace_bytes); layers with Cout < 96 or K < 96 run
 * the fp32-MFMA kernel (exact).  (A/B switch: bgs_conv2d_wgrad_bfx_enable in include/bgs_tuning.h.) */
size_t bgs_conv2d_wgrad_bfx_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R, int S,
                                            int stride, int pad);
int bgs_conv2d_wgrad_nhwc_f32_bfx(const float* x, const float* dy, float* dw, float* db, int N, int H,
                                  int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  int accumulate, int planes, void* workspace, bgs_stream_t stream);

/* Eval-mode BatchNorm folded into the filter, and the backward of the fold (csrc/bn_fold.hip;
 * resnet.py:535-542 `norm_eval=True`): w [Cout,Cin,R,S] (the reference's parameter layout), optional
 * conv_bias [Cout], BN gamma / beta / running mean / var [Cout] (all NULL: no BN, scale 1) ->
 * wf [Cout,R,S,cin_padded] (KRSC, channels zero-padded) = w * s, bf [Cout] = beta - mean * s
 * (+ conv_bias * s), s = gamma / sqrt(var + eps).  Backward: dwf, dbf (NULL = 0) -> dw (NULL:
 * skipped), dconv_bias, dgamma, dbeta (each NULL: skipped).  R * S * cin_padded <= 8192. */
int bgs_fold_conv_bn_fwd(const float* w, const float* conv_bias, const float* gamma,
                         const float* beta, const float* mean, const float* var, float eps, int Cout,
                         int Cin, int R, int S, int cin_padded, float* wf, float* bf,
                         bgs_stream_t stream);
int bgs_fold_conv_bn_bwd(const float* dwf, const float* dbf, const float* w, const float* conv_bias,
                         const float* gamma, const float* mean, const float* var, float eps, int Cout,
                         int Cin, int R, int S, int cin_padded, float* dw, float* dconv_bias,
                         float* dgamma, float* dbeta, bgs_stream_t stream);


/* The 3x3 / stride 1 / pad 1 case of bgs_conv2d_nhwc_f32 (same call sites: fpn.py:131-134 output
 * convs, rpn_head.py:31 rpn_conv, resnet.py:244 conv2) with the workgroup's 8 x 16 output pixels
 * + halo staged in LDS once per 16-channel chunk and reused by the nine taps (DESIGN.md appendix
 * A): 9x less input traffic.  y = act(conv(x, w) + bias); Cin % 16 == 0; no residual.  First
 * version (fixed 128 x 128 tile, no split-K): the host mirror uses it where it is faster than the
 * general kernel — M >= 100000 output pixels and Cout % 128 == 0 (BGS_CONV_HALO=0|1 overrides). */
int bgs_conv3x3_halo_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int N,
                              int H, int W, int Cin, int Cout, int relu, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The same convolutions on the bf16 matrix cores with fp32-faithful results ("bf16x6",
 * csrc/conv_bfx.hip): every fp32 operand is split exactly into three bf16 terms
 * (hi + mid + lo, round-to-nearest-even) and the six products with i + j <= 2 are accumulated
 * in fp32 by v_mfma_f32_32x32x16_bf16 — 6/16 of the matrix-pipe time of the fp32 MFMA kernel at
 * the same error against an fp64 reference (dropped terms <= 2^-25 |a b|).  Same call sites and
 * epilogues as bgs_conv2d_nhwc_f32_ws / bgs_conv2d_dgrad_nhwc_f32_ws / bgs_conv3x3_halo_nhwc_f32;
 * activations stay fp32 NHWC.  The filter is split once by the caller:
 *   bgs_conv_bfx_split_weights(w [rows][K] fp32 -> out, bgs_conv_bfx_weight_bytes(rows, K) bytes,
 *   16-byte aligned), layout [3][2*ceil(K/32)][rows][16] bf16 (zero-padded K tail); rows = Cout
 *   (forward; K = R*S*Cin) or Cin (data gradient: the flipped / transposed filter, K = R*S*Cout).
 * planes = 3: the fp32-faithful mode above.  planes = 1: only the `hi` plane of both operands is
 *   used — operands rounded to bf16 (RNE), fp32 accumulate, fp32 results: the arithmetic of the
 *   reference's fp16/bf16 autocast (mmdet/core/fp16/decorators.py:9-160) with fp32 storage, 1/6 of
 *   the matrix-pipe work (BASELINE cfg[4] "bf16"); the split buffer is the same (plane 0 = bf16(w)).
 * workspace: bgs_conv_bfx_workspace_bytes(M, Cout, K) / bgs_conv3x3_halo_bfx_workspace_bytes(...)
 * bytes of split-K scratch (0 / NULL is always legal).
 * Tuning / A/B hooks and the *_last_launch queries of these kernels: include/bgs_tuning.h
 * (halo variant: 0 = default = 4: filter slices by LDS-DMA; 2: register-staged slices; 1: first
 * version; bgs_conv3x3_halo_bfx_last_launch reports the variant in bits 8.. of *nb).
 * (tile 0 = auto | 11 | 12 | 21 | 22 as MB*10+NB blocks of 64; splitk -1 = auto | 1..16);
 * *_last_launch report what the last launch used (tests assert the instantiation they meant
 * to cover).
 * ---------------------------------------------------------------------------------- */
size_t bgs_conv_bfx_weight_bytes(int rows, int K);
int bgs_conv_bfx_split_weights(const float* w, void* out, int rows, int K, bgs_stream_t stream);
/* The split planes of the DATA-GRADIENT filter straight from the forward filter w [Cout,R,S,Cin]:
 * = bgs_conv_bfx_split_weights of wt[ci][r'][s'][co] = w[co][R-1-r'][S-1-s'][ci] (rows = Cin, K = R*S*Cout;
 * `out`: bgs_conv_bfx_weight_bytes(Cin, R*S*Cout) bytes) — one launch instead of flip + permute + split per
 * trained conv and step (`selectp = 0`). */
int bgs_conv_bfx_split_weights_dgrad(const float* w, void* out, int Cout, int R, int S, int Cin,
                                     bgs_stream_t stream);
size_t bgs_conv_bfx_workspace_bytes(long long M, int Cout, int K);
int bgs_conv2d_nhwc_f32_bfx_ws(const float* x, const void* wsplit, const float* bias,
                               const float* residual, float* y, int N, int H, int W, int Cin,
                               int Cout, int R, int S, int stride, int pad, int relu,
                               int residual_mode, int planes, void* workspace,
                               size_t workspace_bytes, bgs_stream_t stream);
int bgs_conv2d_dgrad_nhwc_f32_bfx_ws(const float* dy, const void* wt_split, const float* residual,
                                     const float* mask, float* dx, int N, int H, int W, int Cin,
                                     int Cout, int R, int S, int stride, int pad,
                                     int residual_mode, int planes, void* workspace,
                                     size_t workspace_bytes, bgs_stream_t stream);
size_t bgs_conv3x3_halo_bfx_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int bgs_conv3x3_halo_nhwc_f32_bfx(const float* x, const void* wsplit, const float* bias, float* y,
                                  int N, int H, int W, int Cin, int Cout, int relu, int planes,
                                  void* workspace, size_t workspace_bytes, bgs_stream_t stream);
/* ... with `mask` [N,H,W,Cout] or NULL (y = mask > 0 ? y : 0): the data gradient of a 3x3 / stride 1
 * / pad 1 conv IS this conv of dy with the flipped, transposed filter, the mask being the ReLU
 * backward of the forward conv's input (functional.conv2d_dgrad_nhwc routes those layers here). */
int bgs_conv3x3_halo_nhwc_f32_bfx_ex(const float* x, const void* wsplit, const float* bias,
                                     const float* mask, float* y, int N, int H, int W, int Cin,
                                     int Cout, int relu, int planes, void* workspace,
                                     size_t workspace_bytes, bgs_stream_t stream);
/* Round 6.  The second half of a frozen ResNet bottleneck (mmdet/models/backbones/resnet.py:239-266) in ONE launch:
 * conv2 (3x3 / stride 1 / pad 1, Cmid -> Cmid, folded BN bias2, ReLU) -> conv3 (1x1, Cmid -> Cout3, folded BN bias3)
 * + residual + ReLU (relu3).  x [N,H,W,Cmid] fp32 NHWC; w2split / w3split = bgs_conv_bfx_split_weights of the folded
 * filters viewed as [Cmid][9 Cmid] / [Cout3][Cmid]; residual [N,H,W,Cout3] or NULL; y [N,H,W,Cout3].  The 64-channel
 * intermediate never reaches HBM.  Supported: Cmid = 64, Cout3 = 256 (ResNet-50 layer1), 16-byte aligned pointers;
 * BGS_ERR_UNSUPPORTED otherwise (run the two launches).  BIT-IDENTICAL to bgs_conv3x3_halo_nhwc_f32_bfx followed by
 * bgs_conv2d_nhwc_f32_bfx_ws with the residual.  The launch is bottleneck_tail_planes_kernel (8 x 8-pixel workgroups,
 * the whole 64-channel patch staged once; csrc/bottleneck_tail_planes.hip); BGS_FUSED_C3_PLANES=0 keeps the first form
 * (conv3x3_c3_fused_bfx_kernel, 8 x 16 pixels).  Same results, bit for bit. */
int bgs_conv3x3_c3_fused_nhwc_f32_bfx(const float* x, const void* w2split, const float* bias2,
                                      const void* w3split, const float* bias3, const float* residual,
                                      float* y, int N, int H, int W, int Cmid, int Cout3, int relu3,
                                      bgs_stream_t stream);


/* Grouped 3x3 convolution (pad 1, stride 1 or 2) + bias + ReLU, NHWC: conv2 of the ResNeXt
 * bottleneck (mmdet/models/backbones/resnext.py:47-57, cfg 5 = X101-64x4d).  x [N,H,W,C],
 * w [C,3,3,C/groups] (output channel, tap, input channel within the group), bias [C] or NULL,
 * y [N,Ho,Wo,C].  C/groups in {4, 8, 16, 32}. */
int bgs_grouped_conv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y,
                                 int N, int H, int W, int C, int groups, int stride, int relu,
                                 bgs_stream_t stream);

/* The same layer with both operands rounded to bf16 (RNE) for v_mfma_f32_16x16x16_bf16, fp32 tensors and
 * fp32 accumulate: the bf16 mode of BASELINE cfg[4].  Stride 1 and C % 64 == 0 only (BGS_ERR_UNSUPPORTED
 * otherwise: use the fp32 entry point). */
int bgs_grouped_conv3x3_nhwc_bf16ops(const float* x, const float* w, const float* bias, float* y,
                                     int N, int H, int W, int C, int groups, int stride, int relu,
                                     bgs_stream_t stream);

/* bf16 STORAGE mode of BASELINE cfg[4] (csrc/conv_bf16s.hip): the reference trains its X101 configurations
 * under Fp16OptimizerHook + wrap_fp16_model (mmdet/core/fp16/hooks.py:11-127, decorators.py:8-160): every
 * trunk activation is a HALF tensor in memory, products are half x half with fp32 accumulation.  These entry
 * points are the conv / linear, grouped-conv and max-pool call sites of `bgs_conv2d_nhwc_f32*`,
 * `bgs_grouped_conv3x3_nhwc_f32` and `bgs_maxpool3x3s2_nhwc_f32` with bf16 NHWC activations:
 *   bgs_conv2d_nhwc_bf16s: x bf16 [N,H,W,Cin]; w_hi = the first plane of bgs_conv_bfx_split_weights
 *     (bf16(w), [2 ceil(K/32)][Cout][16]); bias fp32 or NULL; residual_mode 0 | 1 (same shape) | 2 (nearest-2x
 *     upsampled, [N,Ho/2,Wo/2,Cout]); residual_bf16 / y_bf16: element type of residual / y (1 = bf16, 0 = fp32:
 *     consumers outside the trunk, e.g. the FPN laterals fpn.py:118-127).  y = act(bf16(x) * bf16(w) + bias +
 *     residual), fp32 accumulate, rounded once on store.  Cin % 8 == 0, Cout % 8 == 0, 16-byte aligned pointers.
 *   bgs_grouped_conv3x3_nhwc_bf16s: x, y bf16; w [C,3,3,C/groups] fp32 (rounded to bf16 in registers).
 *   bgs_maxpool3x3s2_nhwc_f32_to_bf16: the fp32 stem output pooled into the bf16 trunk. */
int bgs_conv2d_nhwc_bf16s(const void* x, const void* w_hi, const float* bias, const void* residual,
                          int residual_mode, int residual_bf16, void* y, int y_bf16, int N, int H, int W,
                          int Cin, int Cout, int R, int S, int stride, int pad, int relu,
                          bgs_stream_t stream);
int bgs_grouped_conv3x3_nhwc_bf16s(const void* x, const float* w, const float* bias, void* y, int N, int H,
                                   int W, int C, int groups, int stride, int relu, bgs_stream_t stream);
int bgs_maxpool3x3s2_nhwc_f32_to_bf16(const float* x, void* y, int N, int H, int W, int C,
                                      bgs_stream_t stream);

/* Backward of bgs_grouped_conv3x3_nhwc_f32 (`selectp = 0` on the ResNeXt configs; the reference
 * gets it from cuDNN through autograd of nn.Conv2d(groups=...), resnext.py:47-57).
 * dgrad: dy [N,Ho,Wo,C] -> dx [N,H,W,C] with the caller's re-laid-out filter
 *   wt[g*cg+cl][2-r][2-s][co_local] = w[g*cg+co_local][r][s][cl] (transposed inside each group,
 *   flipped); stride 1 or 2 (the MFMA forward kernel on dy, zero-upsampled for stride 2).
 * wgrad: dw [C,3,3,C/groups] (+)= sum_m dy[m,co] x[..]; db [C] (+)= column sums of dy (db may be
 *   NULL); partial sums per 1024-pixel chunk are added in a fixed order (bitwise reproducible);
 *   workspace >= bgs_grouped_conv3x3_wgrad_workspace_bytes(...). */
int bgs_grouped_conv3x3_dgrad_nhwc_f32(const float* dy, const float* wt, float* dx, int N, int H,
                                       int W, int C, int groups, int stride, bgs_stream_t stream);
size_t bgs_grouped_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int C, int groups, int stride);
int bgs_grouped_conv3x3_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, float* db, int N,
                                       int H, int W, int C, int groups, int stride, int accumulate,
                                       void* workspace, bgs_stream_t stream);

/* Image batch [N, C <= 4, H, W] fp32 (the reference's NCHW input, resnet.py:522) -> [N, H, W, 4] with the
 * channel axis zero-padded: the 16-byte-pixel input of the stem conv (one launch instead of pad + copy). */
int bgs_nchw_to_nhwc4_f32(const float* x, float* y, int N, int C, int H, int W, bgs_stream_t stream);

/* The ResNet stem in one launch (csrc/stem_fused.hip; mmdet/models/backbones/resnet.py:522-533: conv1 7x7 / stride 2 /
 * pad 3 (3 -> 64, eval-BN folded by the caller) -> ReLU -> MaxPool2d(3, 2, 1)), bf16x6 arithmetic (fp32-faithful),
 * reading the NCHW image directly: img [N, 3, H, W] float -> out [N, PH, PW, 64] NHWC float, PH = floor((CH - 1) / 2) + 1
 * with CH = floor((H - 1) / 2) + 1 (likewise W).  wsplit: bgs_stem_fused_split_weights of the folded filter
 * [64][7][7][cin_stride >= 3] (channels 0..2 are used), bgs_stem_fused_weight_bytes() bytes, 16-byte aligned; bias [64]
 * or NULL.  Replaces the chain bgs_nchw_to_nhwc4_f32 -> bgs_conv2d_nhwc_f32_bfx_ws -> bgs_maxpool3x3s2_nhwc_f32 for a
 * frozen stem (no backward): the [N, CH, CW, 64] conv map never reaches HBM. */
size_t bgs_stem_fused_weight_bytes(void);
int bgs_stem_fused_split_weights(const float* w, int cin_stride, void* out, bgs_stream_t stream);
int bgs_stem_conv7x7s2_relu_maxpool_nchw_f32(const float* img, const void* wsplit, const float* bias, float* out,
                                             int N, int H, int W, bgs_stream_t stream);

/* 3x3 / stride 2 / pad 1 max pooling, NHWC (ResNet stem, resnet.py:452). C % 4 == 0.
 * y [N, (H-1)/2+1, (W-1)/2+1, C]. */
int bgs_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C,
                              bgs_stream_t stream);
/* Its backward (`frozen_stages < 1`): x = the pool's input, dy [N,Ho,Wo,C] -> dx [N,H,W,C]
 * (overwritten).  The gradient of a window goes to its FIRST maximum in scan order, as in torch's
 * max_pool2d backward; gather formulation, no atomics. */
int bgs_maxpool3x3s2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W,
                                  int C, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Multi-level RoIAlign forward (NHWC).  Replaces SingleRoIExtractor.forward
 *   (mmdet/models/roi_extractors/single_level.py:54-73,89-107) + RoIAlignFunction.forward
 *   (mmdet/ops/roi_align/roi_align.py:12-29 -> src/roi_align_kernel.cu:16-124): level
 *   = clamp(floor(log2(sqrt(w*h)/finest_scale + 1e-6)), 0, L-1) is evaluated in the kernel,
 *   legacy box semantics (roi_end = (x2+1)*scale, samples outside [-1, H] x [-1, W] give 0).
 *   sample_num >= 0 as in the reference (roi_align_kernel.cu:95-99): n > 0 = an n x n sample grid per bin
 *   (2 in every shipped config: its own unrolled kernel); 0 = adaptive, ceil(roi_size / pooled_size)
 *   samples per axis and RoI (a degenerate RoI then has no samples: 0 / 0 = NaN, as in the reference).
 *   host_feats   [L] HOST array of device pointers to [num_images, H_l, W_l, C] float maps
 *   host_heights/host_widths/host_scales [L] HOST arrays (H_l, W_l, 1/stride_l)
 *   rois [K,5] float (batch_ind, x1, y1, x2, y2);  out [K, pooled_h, pooled_w, C] float
 *   (bin-major, channels contiguous; the reference's [K,C,ph,pw] is the transpose);
 *   levels_out [K] int32 or NULL (the level chosen per RoI, for tests).
 * ---------------------------------------------------------------------------------- */
int bgs_roi_align_nhwc_fwd(const float* const* host_feats, const int* host_heights,
                           const int* host_widths, const float* host_scales, int num_levels,
                           int num_images, float finest_scale, const float* rois, int K, int C,
                           int pooled_h, int pooled_w, int sample_num, float* out,
                           int* levels_out, bgs_stream_t stream);

/* RoIAlign backward (RoIAlignFunction.backward, mmdet/ops/roi_align/roi_align.py:31-53 ->
 *   src/roi_align_kernel.cu:149-266), `selectp = 0` path only.  dout [K,pooled_h,pooled_w,C];
 *   host_dfeats [L] HOST array of device pointers to the per-level gradient maps
 *   [num_images,H_l,W_l,C], which are ACCUMULATED INTO with fp32 hardware atomics (zero them
 *   first if nothing else contributed).  Same level rule / box semantics as the forward. */
int bgs_roi_align_nhwc_bwd(float* const* host_dfeats, const int* host_heights,
                           const int* host_widths, const float* host_scales, int num_levels,
                           int num_images, float finest_scale, const float* rois, int K, int C,
                           int pooled_h, int pooled_w, int sample_num, const float* dout,
                           bgs_stream_t stream);

/* The half instantiation of the reference's dtype dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 *   roi_align_kernel.cu:136,281): feature maps / `out` (forward) and `dout` (backward) are IEEE fp16, RoIs fp32
 *   (widen them: exact), arithmetic fp32, the gradient maps fp32 (the caller adds them into its fp16
 *   bottom_grad, mmdet/ops/roi_align/roi_align.py:45-52).  Any sample_num >= 0.  Same contract otherwise. */
int bgs_roi_align_nhwc_fwd_f16(const void* const* host_feats, const int* host_heights,
                               const int* host_widths, const float* host_scales, int num_levels,
                               int num_images, float finest_scale, const float* rois, int K, int C,
                               int pooled_h, int pooled_w, int sample_num, void* out, int* levels_out,
                               bgs_stream_t stream);
int bgs_roi_align_nhwc_bwd_f16(float* const* host_dfeats, const int* host_heights,
                               const int* host_widths, const float* host_scales, int num_levels,
                               int num_images, float finest_scale, const float* rois, int K, int C,
                               int pooled_h, int pooled_w, int sample_num, const void* dout,
                               bgs_stream_t stream);

/* RoIAlign with a fused average pool and accumulate (HTC semantic fusion,
 *   mmdet/models/detectors/htc.py:57-64,88-96: `semantic_roi_extractor([semantic_feat], rois)`
 *   at 14x14 -> `F.adaptive_avg_pool2d(., 7)` -> `bbox_feats += .`).  Same contract as
 *   bgs_roi_align_nhwc_fwd / _bwd plus:
 *   pool        1 or 2: every output bin is the mean of pool x pool bins of the
 *               (pooled_h*pool) x (pooled_w*pool) RoIAlign grid (pool == 2 needs C <= 256);
 *   accumulate  != 0: out += result (out holds the box / mask RoI features).
 *   (pool == 2 and accumulate exist for sample_num = 2 only: BGS_ERR_UNSUPPORTED otherwise.)
 *   The backward scatters dout / (4 * pool^2) per sample, atomically, into host_dfeats. */
int bgs_roi_align_nhwc_fwd_ex(const float* const* host_feats, const int* host_heights,
                              const int* host_widths, const float* host_scales, int num_levels,
                              int num_images, float finest_scale, const float* rois, int K, int C,
                              int pooled_h, int pooled_w, int sample_num, int pool,
                              int accumulate, float* out, int* levels_out, bgs_stream_t stream);
int bgs_roi_align_nhwc_bwd_ex(float* const* host_dfeats, const int* host_heights,
                              const int* host_widths, const float* host_scales, int num_levels,
                              int num_images, float finest_scale, const float* rois, int K, int C,
                              int pooled_h, int pooled_w, int sample_num, int pool,
                              const float* dout, bgs_stream_t stream);

/* Bilinear resize of NHWC maps, align_corners = 1 only (BGS_ERR_UNSUPPORTED otherwise).
 *   Replaces F.interpolate(feat, size, mode='bilinear', align_corners=True) of the HTC semantic
 *   head (mmdet/models/mask_heads/fused_semantic_head.py:88-93); arithmetic = torch's
 *   upsample_bilinear2d.  x [N,H,W,C] -> y [N,Ho,Wo,C], C % 4 == 0.
 *   _bwd: dy [N,Ho,Wo,C] is scattered with fp32 atomics INTO dx [N,H,W,C] (zero it first). */
int bgs_resize_bilinear_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int Ho,
                                 int Wo, int align_corners, bgs_stream_t stream);
int bgs_resize_bilinear_nhwc_bwd_f32(const float* dy, float* dx, int N, int H, int W, int C,
                                     int Ho, int Wo, int align_corners, bgs_stream_t stream);

/* Batched sorted top-k of fp32 rows: the proposal pre-selection of RPNHead.get_bboxes_single
 *   (mmdet/models/anchor_heads/rpn_head.py:79-83 `scores.topk(cfg.nms_pre)` per level, :99-103
 *   `scores.topk(num)` over the NMS survivors).  P <= 64 rows of different lengths / k in one set
 *   of launches.
 *   host_rows [P] HOST array of device pointers to float rows; host_len / host_k [P] HOST ints
 *   (k[p] <= kmax <= 4096; min(k, len) entries are produced); host_inner / host_pitch [P] HOST
 *   ints or both NULL: element i of row p is row[(i / inner) * pitch + i % inner] (inner == 0 or
 *   NULL: contiguous) — the objectness logits are the first A of the 5A channels of the fused RPN
 *   head output and are read in place;
 *   out_val / out_idx [P, kmax]: the largest values in DESCENDING order and their positions in the
 *   row, zero-filled beyond min(k, len).  Which of several elements equal to the k-th value are
 *   returned is unspecified (as for torch.topk).  workspace: bgs_topk_workspace_bytes(P, kmax). */
size_t bgs_topk_workspace_bytes(int P, int kmax);
int bgs_topk_sorted_f32(const float* const* host_rows, const int* host_len, const int* host_k,
                        const int* host_inner, const int* host_pitch, int P, int kmax,
                        float* out_val, long long* out_idx, void* workspace, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Batched greedy NMS, entirely on the device.  Replaces ops.nms / nms_cuda
 *   (mmdet/ops/nms/nms_wrapper.py:8-49, src/nms_kernel.cu:13-131; CPU variant nms_cpu.cpp:5-59)
 *   for P independent problems at once (image x FPN level in the RPN,
 *   mmdet/models/anchor_heads/rpn_head.py:92).
 *   boxes [P, nmax, 5] float (x1,y1,x2,y2,score), each problem sorted by DESCENDING score;
 *   counts [P] int32 (valid boxes per problem);  legacy +1 IoU;
 *   iou_mode 0: suppress when IoU > thr (nms_cuda), 1: IoU >= thr (nms_cpu);
 *   keep [P, nmax] int32: kept indices (into the sorted order, ascending), first
 *   keep_count[p] entries valid, at most max_keep (<= 0: no limit);
 *   workspace: bgs_nms_workspace_bytes(P, nmax).   nmax <= 4096.
 * ---------------------------------------------------------------------------------- */
size_t bgs_nms_workspace_bytes(int P, int nmax);
int bgs_nms_batched(const float* boxes, const int* counts, int P, int nmax, float iou_thr,
                    int iou_mode, int max_keep, int* keep, int* keep_count, void* workspace,
                    bgs_stream_t stream);

/* The gathers around the RPN's NMS (RPNHead.get_bboxes_single, rpn_head.py:92-103: `proposals = proposals[keep]`,
 * the concatenation over levels and the final `topk(max_num)` selection), fixed-shape, one launch each:
 *   bgs_nms_gather: out_boxes [R,nmax,5] = boxes[r, clamp(keep[r,slot], 0, nmax-1)]; out_scores [R,nmax] = that
 *     box's score for slot < keep_count[r], -1 otherwise (padding slots lose every later top-k);
 *   bgs_gather_boxes: props [N,num,5] = flat[n, idx[n,j]] (idx int64 row positions, e.g. of bgs_topk_sorted_f32),
 *     valid [N,num] uint8 = scores[n,j] >= 0. */
int bgs_nms_gather(const float* boxes, const int* keep, const int* keep_count, int R, int nmax,
                   float* out_boxes, float* out_scores, bgs_stream_t stream);
int bgs_gather_boxes(const float* flat, const long long* idx, const float* scores, int N, int T, int num,
                     float* props, unsigned char* valid, bgs_stream_t stream);
/* The proposal tail in ONE launch (replaces bgs_nms_gather + bgs_topk_sorted_f32 + bgs_gather_boxes there;
 *   mmdet/models/anchor_heads/rpn_head.py:99-103: cat of the levels, `scores.topk(num)`, gather): boxes
 *   [N * L, nmax, 5] (rows sorted by descending score), keep [N * L, nmax] / keep_count [N * L] of bgs_nms_batched
 *   -> props [N, num, 5] = the num best kept boxes of each image over its L levels in descending score order
 *   (ties: lower level first, then NMS order — the order of bgs_topk_sorted_f32 on the concatenated rows),
 *   valid [N, num] uint8 (0, with a zero box, past the number of kept boxes).  L <= 16, L * nmax <= 16384. */
int bgs_nms_merge_select(const float* boxes, const int* keep, const int* keep_count, int N, int L, int nmax,
                         int num, float* props, unsigned char* valid, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Target assignment without the [G, A] IoU matrix.  Replaces MaxIoUAssigner.assign /
 *   assign_wrt_overlaps (mmdet/core/bbox/assigners/max_iou_assigner.py:47-180, incl. its CPU
 *   fallback for > 50 GTs and the Python loop over GTs) and bbox_overlaps
 *   (mmdet/core/bbox/geometry.py:4-63, legacy +1 sizes), for N images at once.
 *   boxes: image n reads box i at boxes + n*box_img_stride + i*box_stride floats
 *          (box_img_stride = 0: the same anchors for every image);
 *   valid [N,A] uint8 or NULL (anchors outside the image take no part and get -1);
 *   gt [sum G,4] concatenated GT boxes, host_gt_offsets [N+1] (HOST);
 *   assigned [N,A] int32: -1 ignore, 0 negative (neg_iou_lo <= max IoU < neg_iou_hi),
 *   g+1 positive (max IoU >= pos_iou_thr -> argmax gt; then every gt with max >= min_pos_iou
 *   claims all boxes attaining its maximum, later gts overriding earlier ones);
 *   max_overlaps_out [N,A] float or NULL.  workspace: bgs_iou_assign_workspace_bytes().
 * ---------------------------------------------------------------------------------- */
size_t bgs_iou_assign_workspace_bytes(int N, int A, int G_total);
int bgs_iou_assign(const float* boxes, long long box_img_stride, int box_stride,
                   const uint8_t* valid, const float* gt, const int* host_gt_offsets, int N, int A,
                   float pos_iou_thr, float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                   int* assigned, float* max_overlaps_out, void* workspace, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * RPN loss over the sampled anchors.  Replaces the target encoding of anchor_target_single
 *   (mmdet/core/anchor/anchor_target.py:118-152) + AnchorHead.loss / loss_single
 *   (mmdet/models/anchor_heads/anchor_head.py:142-207) for the sigmoid-objectness RPN:
 *     loss_cls[l]  = w_cls  * sum BCEWithLogits(x, is_pos) * label_weight / num_total_samples
 *     loss_bbox[l] = w_bbox * sum SmoothL1(pred - bbox2delta(anchor, gt); beta) / num_total_samples
 *     num_total_samples = sum_n max(n_pos_n, 1) + max(n_neg_n, 1)
 *   host_level_outs [L] HOST array of device pointers to the fused head output of each level,
 *   [N, H_l*W_l, A + 4A] float (objectness logits first, then 4 deltas per anchor);
 *   host_level_hw [L] (H_l*W_l); anchors [A_total,4]; assigned [N,A_total] int32;
 *   pos_mask / neg_mask [N,A_total] uint8 (the sampled anchors); gt as in bgs_iou_assign.
 *   Outputs loss_cls_out [L], loss_bbox_out [L], num_total_out [1] or NULL.  Values only
 *   (the RPN is frozen in every shipped BAGS config).
 * ---------------------------------------------------------------------------------- */
size_t bgs_rpn_loss_workspace_bytes(int N, int A_total, int L);
int bgs_rpn_loss(const float* const* host_level_outs, const int* host_level_hw, int L,
                 int num_anchors, const float* anchors, const int* assigned,
                 const uint8_t* pos_mask, const uint8_t* neg_mask, const float* gt,
                 const int* host_gt_offsets, int N, const float* host_means,
                 const float* host_stds, float beta, float pos_weight, float loss_weight_cls,
                 float loss_weight_bbox, float* loss_cls_out, float* loss_bbox_out,
                 float* num_total_out, void* workspace, bgs_stream_t stream);

/* RandomSampler of the RPN in one launch (mmdet/core/bbox/samplers/base_sampler.py:35-78,
 *   random_sampler.py:19-53): assigned [N, A] int32 (-1 ignore / 0 negative / > 0 positive) ->
 *   pos_mask, neg_mask [N, A] uint8 with EXACTLY min(int(num * pos_fraction), #pos) positives and
 *   min(num - #pos_sampled [, int(neg_pos_ub * max(#pos_sampled, 1)) when neg_pos_ub >= 0], #neg)
 *   negatives per image, uniformly without replacement (radix select over bijective 32-bit
 *   keys of (seed, *draw_counter, image, anchor)).  draw_counter: device int64 [1] or NULL.
 *   workspace: bgs_sample_pos_neg_workspace_bytes(N) bytes (counters + per-image key lists; the
 *   three launches scan -> select -> mark communicate through it). */
size_t bgs_sample_pos_neg_workspace_bytes(int N);
int bgs_sample_pos_neg(const int* assigned, int N, int A, int num, float pos_fraction,
                       float neg_pos_ub, uint64_t seed, const long long* draw_counter,
                       uint8_t* pos_mask, uint8_t* neg_mask, void* workspace,
                       bgs_stream_t stream);

/* RandomSampler of the RoI head (two_stage.py:192-210; add_gt_as_proposals candidates = GT boxes
 *   followed by the proposals): host_assigned [N] HOST array of device pointers to each image's
 *   assignment vector (int32, host_counts[n] <= 4096 entries); per image inds [num] int64 into
 *   that vector: sampled positives first (<= int(num * pos_fraction), a uniform subset if there
 *   are more), then uniformly sampled negatives, then padding (valid = 0).  is_pos, valid
 *   [N, num] uint8. */
int bgs_sample_rois(const int* const* host_assigned, const int* host_counts, int N, int num,
                    float pos_fraction, uint64_t seed, const long long* draw_counter, long long* inds,
                    uint8_t* is_pos, uint8_t* valid, bgs_stream_t stream);
/* The same draw, also emitting what the mask branch and the cascade refinement read from the SamplingResult
 * (mmdet/core/bbox/samplers/sampling_result.py:7-24): gt_ind [N, num] int32 = assigned[inds] - 1
 * (`pos_assigned_gt_inds`; -1 for negatives) and is_gt [N, num] uint8 = inds < host_gt_counts[n] (`pos_is_gt`:
 * the GT boxes `add_gt_as_proposals` put in front of the candidates, base_sampler.py:49-53).  Either output and
 * host_gt_counts may be NULL. */
int bgs_sample_rois_ex(const int* const* host_assigned, const int* host_counts, const int* host_gt_counts,
                       int N, int num, float pos_fraction, uint64_t seed, const long long* draw_counter,
                       long long* inds, uint8_t* is_pos, uint8_t* valid, int* gt_ind, uint8_t* is_gt,
                       bgs_stream_t stream);

/* Sampling keys for RandomSampler (mmdet/core/bbox/samplers/random_sampler.py:19-53) drawn on the
 *   device: out[i] = 62-bit splitmix64(seed, *draw_counter, i), i < n.  draw_counter: device int64
 *   [1] (or NULL = 0) that the caller advances between draws — constant kernel arguments under
 *   hipGraph replay, fresh keys every replay. */
int bgs_random_keys(uint64_t seed, const long long* draw_counter, int n, long long* out,
                    bgs_stream_t stream);

/* Gradient of bgs_rpn_loss w.r.t. the fused head outputs (autograd of AnchorHead.loss_single,
 *   anchor_head.py:131-161; `selectp = 0` path).  host_level_douts [L] HOST array of device
 *   pointers to ZERO-FILLED maps shaped like the outputs; num_total = the normaliser written
 *   by bgs_rpn_loss; grad_loss_cls / grad_loss_bbox [L] device arrays (upstream gradients of
 *   the 2L loss scalars).  Only the sampled anchors' entries are written. */
int bgs_rpn_loss_grad(const float* const* host_level_outs, float* const* host_level_douts,
                      const int* host_level_hw, int L, int num_anchors, const float* anchors,
                      const int* assigned, const uint8_t* pos_mask, const uint8_t* neg_mask,
                      const float* gt, const int* host_gt_offsets, int N, const float* host_means,
                      const float* host_stds, float beta, float pos_weight, float loss_weight_cls,
                      float loss_weight_bbox, const float* num_total, const float* grad_loss_cls,
                      const float* grad_loss_bbox, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Proposal decode: gather + delta2bbox + clamp + sigmoid for the top-k anchors of every
 *   (image, level).  Replaces rpn_head.py:62-90 + transforms.py:34-111.
 *   top_idx / top_logit [N,L,nmax] (int64 / float; entries >= host_level_counts[l] ignored);
 *   host_img_hw [N,2] (img_shape h, w); boxes_out [N,L,nmax,5] (x1,y1,x2,y2,score; zero padded).
 * ---------------------------------------------------------------------------------- */
int bgs_decode_proposals(const float* const* host_level_outs, const int* host_level_hw,
                         const int* host_level_counts, int L, int num_anchors,
                         const float* anchors, const long long* top_idx, const float* top_logit,
                         int N, const int* host_img_hw, const float* host_means,
                         const float* host_stds, float wh_ratio_clip, int nmax, float* boxes_out,
                         bgs_stream_t stream);

/* Cascade stage hand-over in one launch: BBoxHead.refine_bboxes -> regress_by_class -> delta2bbox
 * (mmdet/models/bbox_heads/bbox_head.py:169-239, mmdet/core/bbox/transforms.py:34-111).  rois [K,5]
 * (image index, x1, y1, x2, y2), labels [K] int64 (NULL when pred_cols == 4: class-agnostic),
 * bbox_pred [K, pred_cols], host_img_hw = {h0, w0, h1, w1, ...} (N images: the clip bounds), wh_ratio_clip
 * as delta2bbox's (16 / 1000) -> out [K,4].  GT rows / padding slots are the caller's business (masks). */
int bgs_refine_boxes(const float* rois, const long long* labels, const float* bbox_pred, int K,
                     int pred_cols, const int* host_img_hw, int N, const float* host_means,
                     const float* host_stds, float wh_ratio_clip, float* out, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * RoI-head targets of the sampled RoIs.  Replaces bbox2roi (transforms.py:149-168) +
 *   bbox_target_single (mmdet/core/bbox/bbox_target.py:35-61).  Per image n (HOST pointer
 *   arrays): candidate boxes (row stride host_box_strides[n] floats), assigned [cand] int32,
 *   inds [num] int64 (sampled candidates, positives first), valid [num] uint8 or NULL,
 *   gt_labels [G_n] int64.  Outputs rois [N*num,5], labels [N*num] int64, label_weights,
 *   bbox_targets [N*num,4], bbox_weights [N*num,4].
 * ---------------------------------------------------------------------------------- */
int bgs_rcnn_targets(const float* const* host_boxes, const int* host_box_strides,
                     const int* const* host_assigned, const long long* const* host_inds,
                     const uint8_t* const* host_valid, const long long* const* host_gt_labels,
                     const float* gt, const int* host_gt_offsets, int N, int num,
                     const float* host_means, const float* host_stds, float pos_weight,
                     float* rois, long long* labels, float* label_weights, float* bbox_targets,
                     float* bbox_weights, bgs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mask branch (cfg 4: gs_mask_rcnn_r50_fpn_1x_lvis).
 *
 * bgs_mask_target: mask_target_single (mmdet/core/mask/mask_target.py:16-38) for all positive
 *   RoIs in one launch.  host_masks [num_images] HOST array of device pointers to the uint8 GT
 *   bitmaps [G_n, mask_h, mask_w] of each image; host_num_gt [num_images]; rois [P, roi_stride]
 *   float (batch_ind, x1, y1, x2, y2, ...); gt_inds [P] int32 = pos_assigned_gt_inds; valid [P]
 *   uint8 or NULL (fixed-shape padding slots -> all-zero target); out [P, S, S] float in {0,1}
 *   = cv2.resize(crop, (S, S), INTER_LINEAR) on uint8 (what mmcv.imresize calls).
 *
 * bgs_mask_gt_logits: logits_out[p, pix] = <feat[p,pix,:], weight[labels[p],:]> + bias[labels[p]]
 *   — FCNMaskHead.conv_logits (fcn_mask_head.py:82,101) restricted to the channel that
 *   mask_cross_entropy (cross_entropy_loss.py:54-61) / get_seg_masks (:162-165) read.
 *   feat [P, pixels, C] NHWC float (C % 4 == 0), weight [num_classes, C], bias or NULL.
 *
 * bgs_mask_bce: the same single-channel conv fused with binary_cross_entropy_with_logits and
 *   its backward.  partial_out [bgs_mask_bce_partials(P)] per-workgroup loss sums (the loss is
 *   norm * sum(partials)); norm [1] device float = 1 / (#valid RoIs * pixels) (the 'mean'
 *   reduction); dfeat [P,pixels,C] / dweight [num_classes,C] / dbias [num_classes] or NULL —
 *   dweight/dbias are ACCUMULATED INTO with fp32 atomics (zero them first).
 * ---------------------------------------------------------------------------------- */
int bgs_mask_target(const uint8_t* const* host_masks, const int* host_num_gt, int num_images,
                    int mask_h, int mask_w, const float* rois, int roi_stride, const int* gt_inds,
                    const uint8_t* valid, int P, int mask_size, float* out, bgs_stream_t stream);
int bgs_mask_gt_logits(const float* feat, const float* weight, const float* bias,
                       const long long* labels, int P, int pixels, int C, int num_classes,
                       float* logits_out, bgs_stream_t stream);
/* bgs_mask_paste_u8: the resize + threshold + paste of FCNMaskHead.get_seg_masks (fcn_mask_head.py:156-176; the RLE
 *   encoding that follows needs pycocotools and stays on the host): probs [K, S, S] float = sigmoid of every
 *   detection's own class channel; boxes [K, box_stride >= 4] float (x1, y1, x2, y2, ...), divided by scale_factor and
 *   truncated to int32 as :164 does; out [K, img_h, img_w] uint8 in {0, 1} (every byte written; 4-byte aligned):
 *   out[k, y1 : y1 + h, x1 : x1 + w] = cv2.resize(probs[k], (w, h), INTER_LINEAR) > thr, zero elsewhere; the part of a
 *   box outside the image is clipped. */
int bgs_mask_paste_u8(const float* probs, const float* boxes, int box_stride, int K, int S, float scale_factor,
                      float thr, int img_h, int img_w, unsigned char* out, bgs_stream_t stream);
int bgs_mask_bce_partials(int P);
int bgs_mask_bce(const float* feat, const float* weight, const float* bias, const long long* labels,
                 const float* target, const uint8_t* valid, const float* norm, int P, int pixels,
                 int C, int num_classes, float* partial_out, float* dfeat, float* dweight,
                 float* dbias, bgs_stream_t stream);

/* Gradient clipping + SGD update of ALL trainable tensors (csrc/optim.hip): the reference's optimizer hook
 * `DistOptimizerHook.after_train_iter` (mmdet/core/utils/dist_utils.py:51-58): `clip_grads` (max_norm = 35, L2:
 * torch.nn.utils.clip_grad_norm_) -> `optimizer.step()` (torch.optim.SGD with momentum and weight decay,
 * configs/bags/ `optimizer` entries), and the unscale step of `Fp16OptimizerHook` (mmdet/core/fp16/hooks.py:73-79)
 * through `grad_scale` = 1 / loss_scale.  host_params / host_grads / host_momentum / host_numel: HOST arrays of
 * n_tensors device pointers / element counts (fp32; momentum buffers zero before the first step); they are
 * read during the call and passed to the kernels by value (hipGraph-capturable).  max_norm <= 0: no clipping.
 *   g <- g * grad_scale * min(1, max_norm / (||g * grad_scale||_2 + 1e-6))   (in place, as the hook does)
 *   buf <- momentum * buf + (g + weight_decay * p);   p <- p - lr * buf      (torch's operation order)
 * total_norm_out: device float[1] receiving the unclipped norm, or NULL.
 * workspace >= bgs_sgd_clip_workspace_bytes(host_numel, n_tensors). */
size_t bgs_sgd_clip_workspace_bytes(const long long* host_numel, int n_tensors);
int bgs_sgd_clip_step(const void* const* host_params, const void* const* host_grads,
                      const void* const* host_momentum, const long long* host_numel, int n_tensors,
                      float max_norm, float grad_scale, float lr, float momentum, float weight_decay,
                      void* workspace, size_t workspace_bytes, float* total_norm_out, bgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BGS_H_ */
