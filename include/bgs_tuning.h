/* libbgs — tuning, A/B and launch-census entry points.
 *
 * These are NOT part of the drop-in boundary of include/bgs.h (nothing in the reference binds to
 * them): they exist so that tests can pin the kernel instantiation they mean to cover, and so that
 * tools/ can time one decomposition against another inside one process.  Every hook is process-wide
 * and not thread-safe; every arm of every hook produces results within the tolerance (mostly the
 * same bits) of the default.  The library exports them next to the product entry points.
 */
#ifndef BGS_TUNING_H_
#define BGS_TUNING_H_

#include "bgs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Wide-tile 1x1 kernel of the bf16x6 mode (csrc/conv_bfx_wide.hip: 128 x 128 tile, four M-stacked
 * waves, 24 MFMAs per wave and barrier; replaces the 64 x 64 operand ring on the wide-N 1x1 layers
 * of mmdet/models/backbones/resnet.py:220-266 / necks/fpn.py:101-141; bit-identical to it when K is
 * not sliced).  mode 0 = never | 1 = where measured faster (default) | 2 = every eligible layer;
 * nst 0 = auto | 2 | 3 ring stages; splitk -1 = auto | 1..16 K slices.
 * last_launch: bit 0 = the last bf16x6 1x1 launch took it; bits 4..7 ring stages; bits 8.. K slices. */
void bgs_conv_bfx_wide_tuning(int mode, int nst, int splitk);
int bgs_conv_bfx_wide_last_launch(void);

/* Row-per-workgroup GroupSoftmax loss kernel (csrc/gs_loss.hip, bgs_gs_loss_fwd_bwd; the bandwidth-bound form
 * of gs_bbox_head_with0.py:147-186 for N beyond the fused head's 4096 rows): prefetch 0 = every row pays its own
 * global-memory round trip (the round-3 kernel) | 1 = the next row of a workgroup is fetched into registers under the
 * current row's sweeps | 2 / 3 / 4 = 1 + non-temporal row loads / gradient stores / both.  3 is the default
 * (N = 65,536: 4.5 -> 5.3 TB/s).  Bit-identical results in every mode. */
void bgs_gs_loss_tuning(int prefetch);

#ifdef __cplusplus
}
#endif
#endif /* BGS_TUNING_H_ */
