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

/* ---- self-tests ---------------------------------------------------------------------------------
 * bgs_selftest_wave_reduce: in [64] float, out [4] float = {max, sum} by the build's wave64 reduction
 *   primitive (DPP) followed by {max, sum} by a ds_bpermute butterfly.
 * bgs_selftest_mfma_peak: `blocks` workgroups x 4 waves x (4 * iters) v_mfma_f32_32x32x2_f32 with register
 *   operands (4096 flop each) — the fp32 matrix rate the chip sustains under its power limit.
 * bgs_selftest_mfma_peak_bf16: the same with 24 x iters v_mfma_f32_32x32x16_bf16 per wave (32768 flop each) in the
 *   product order of the bf16x6 kernels; random_operands != 0: pseudo-random operand registers (mixed sign and
 *   mantissa, the three planes 2^-8 apart), 0: zero operands — the chip clocks to its power budget, the random
 *   figure / 6 is what a memory-free bf16x6 loop sustains. */
int bgs_selftest_wave_reduce(const float* in, float* out, bgs_stream_t stream);
int bgs_selftest_mfma_peak(int blocks, int iters, float* out, bgs_stream_t stream);
int bgs_selftest_mfma_peak_bf16(int blocks, int iters, int random_operands, float* out, bgs_stream_t stream);

/* ---- launch census ------------------------------------------------------------------------------
 * How often a kernel family was launched by this process since the last reset — lets a test ASSERT that the
 * instantiation it means to pin really ran (e.g. the 8-wave bf16 ring and the LDS-resident grouped conv inside a
 * whole X101 iteration).  Returns the count of `family` (-1 for an unknown id); reset != 0 zeroes every counter
 * after reading. */
#define BGS_CENSUS_BF16_RING8 0      /* conv_igemm_bf16_ring8_kernel                      */
#define BGS_CENSUS_GROUPED_LDS 1     /* grouped_conv3x3_lds_kernel                        */
#define BGS_CENSUS_HALO_BFX4 2       /* conv3x3_halo_bfx4_kernel                          */
#define BGS_CENSUS_DMA_RING64 3      /* conv_igemm_bfx_dma_kernel (64 x 64 operand ring)  */
#define BGS_CENSUS_GS_HEAD_FUSED 4   /* gs_head_fused_kernel                              */
#define BGS_CENSUS_CONV1X1_BRES 5    /* conv1x1_bres_kernel (filter-resident 1x1)         */
#define BGS_CENSUS_WGRAD_BFX 6       /* conv_wgrad_bfx_kernel                             */
#define BGS_CENSUS_ROI_BWD_GATHER 7  /* roi_align backward without global atomics         */
#define BGS_CENSUS_BF16S 8           /* conv_bf16s_kernel (bf16 activations in HBM)       */
#define BGS_CENSUS_GROUPED_BF16S 9   /* grouped 3x3 conv with bf16 activations in HBM     */
#define BGS_CENSUS_BFX_WIDE 10       /* conv1x1_bfx_wide_kernel (128 x 128, M-stacked waves) */
#define BGS_CENSUS_GS_SCALE_GRAD 11   /* gs_head_scale_grad_kernel (a non-unit upstream gradient) */
#define BGS_CENSUS_HALO_WIDE 12       /* conv3x3_halo_bfx7_kernel (16 x 16-pixel x 128-channel units) */
#define BGS_CENSUS_STEM_FUSED 13      /* stem_conv7x7s2_relu_maxpool_kernel (conv + ReLU + max-pool, NCHW in) */
#define BGS_CENSUS_FUSED_C3 14        /* conv3x3_c3_fused_bfx_kernel (conv2 -> conv3 + residual + ReLU of a frozen bottleneck) */
#define BGS_CENSUS_PLANES_3X3 15      /* conv3x3_planes_bfx_kernel (8 x 8 pixels, whole reduction per workgroup) */
#define BGS_CENSUS_PLANES_1X1 16      /* conv1x1_planes_bfx_kernel (A split once per K chunk into LDS planes)    */
#define BGS_CENSUS_FAMILIES 20
int bgs_launch_census(int family, int reset);

/* ---- fused GroupSoftmax head (csrc/gs_loss.hip) ---------------------------------------------------
 * bgs_gs_head_debug_timestamps: buf != NULL (device, [2048][8] uint64) makes every later launch record 8
 *   shader-clock marks per workgroup (tools/gs_phase_times.py); NULL (default) switches it off.
 * bgs_gs_head_tuning: rows per workgroup (0 = default).
 * bgs_gs_head_variant: 0 = one row per workgroup with per-row flag words | 1 = one row per workgroup with ballot
 *   bit planes | 2 / 3 = variant 1 with 2 / 4 rows per workgroup behind one shared prologue (N <= 2048) | 4 / 5 =
 *   2 / 3 with the gradient stored by the wave that owns the bin; < 0 (default): automatic — 4 for N < 1024, 5
 *   for N = 1024, 3 up to 2048, 1 beyond.  Bitwise the same results.  bgs_gs_head_variant_used(N): the variant a
 *   launch with N rows takes. */
void bgs_gs_head_debug_timestamps(unsigned long long* buf);
void bgs_gs_head_tuning(int rows_per_workgroup);
/* Round 6: bgs_gs_head_fold(1) (default 0; < 0 restores it; env BGS_GS_HEAD_FOLD=1 starts with it on): for N <= 2048
 * bgs_gs_head_step reduces its per-row partials INSIDE the main launch — the workgroup that takes the last ticket reads
 * them back and writes loss_out / total_out / the draw counter exactly as gs_head_reduce_kernel does (bitwise the
 * two-launch result): ONE launch per head step.  Measured 1 - 2 us SLOWER than the two launches (N = 1024: 16.6 vs 14.7
 * us; profiles/r10h_gs_head_fold_ab.txt), hence off by default: a tested A/B arm. */
void bgs_gs_head_fold(int on);
void bgs_gs_head_variant(int variant);
int bgs_gs_head_variant_used(int N);

/* ---- convolution kernels --------------------------------------------------------------------------
 * bgs_conv_tuning (fp32 MFMA kernel): tile 0 = auto | 11 | 21 | 22 (MB*10+NB blocks of 64), bk 0 = auto | 16 |
 *   32, splitk 0 = auto | 1..16, noswizzle 1 = plain tile order; bgs_conv_last_launch reports the instantiation
 *   the last launch used.
 * bgs_conv_bfx_tuning (bf16x6 / bf16 modes, csrc/conv_bfx.hip): tile 0 = auto | 11 | 12 | 21 | 22 (+ 0x100: the
 *   register-staged 64 x 64 kernel instead of the LDS-DMA ring; + 0x400 / 0x800: force the 3- / 4-stage ring),
 *   splitk -1 = auto | 1..16; bgs_conv_bfx_last_launch: *tile carries the tile | 0x200 (DMA ring) | 0x400
 *   (3 stages) | 0x1000 (filter-resident 1x1) | 0x2000 (parity-class data gradient) | 0x4000 (wide 1x1).
 * bgs_conv3x3_halo_bfx_tuning: splits -1 = auto | n; variant 0 = default = 4 (filter slices by LDS-DMA) | 2
 *   (register-staged slices) | 1 (first version) | 5 (4 + A-fragment prefetch); bits 16..19 of `variant`: the pixel
 *   tile of variant 4, 0 = default (8 x 16; env BGS_HALO_GEOM) | 1 = 8 x 16 | 2 = 10 x 12 | 3 = 5 x 21 | 4 = the
 *   tile with the fewest tiles per image;
 *   bgs_conv3x3_halo_bfx_last_launch reports the variant in bits 8..15 of *nb and the pixel tile in bits 16..
 *   (0: 8 x 16, 1: 10 x 12, 2: 5 x 21).  Bits 24..27 of `variant` (round 5): the wide pixel tile (variant 7: 16 x 16
 *   pixels x 128 channels per workgroup, whole rounds of 512 units in one launch + the left-over image rows in a
 *   variant-4 launch; bit-identical to variant 4), 0 = leave as is | 1 = off (default; env BGS_HALO_WIDE=0/1/2) | 2 = automatic |
 *   3 = every eligible layer; bgs_conv3x3_halo_bfx_tuning(-1, 0) restores the default.
 *   bgs_conv3x3_halo_bfx_last_wide: units of the variant-7 launch (0: it did not run) and of the variant-4 launch
 *   behind it (last_launch then reports variant 7).
 * bgs_conv1x1_bres_enable (filter-resident 1x1, csrc/conv1x1_bres.hip): 1 (default) = where measured faster,
 *   2 = every layer it can run, 0 = never; bgs_conv1x1_bres_last_launch: 1 when the last call that could have
 *   taken it did.
 * bgs_conv_dgrad_parity_enable: stride-2 data gradients with the GEMM rows grouped by output-pixel parity (1,
 *   default) or in the zero-upsampled form (0).  Bit-identical.
 * bgs_conv2d_wgrad_bfx_enable(0): every weight gradient on the fp32-MFMA kernel (A/B).
 * bgs_conv_bf16s_tuning (bf16-storage kernels): 0 = operands by LDS-DMA (default), 1 = through registers. */
void bgs_conv_tuning(int tile, int bk, int splitk, int noswizzle);
int bgs_conv_last_launch(int* tile, int* bk, int* up, int* splits);
void bgs_conv_bfx_tuning(int tile, int splitk);
int bgs_conv_bfx_last_launch(int* tile, int* splits);
void bgs_conv3x3_halo_bfx_tuning(int splits, int variant);
int bgs_conv3x3_halo_bfx_last_launch(int* nb, int* splits);
int bgs_conv3x3_halo_bfx_last_wide(int* wide_units, int* tail_units);
int bgs_conv3x3_halo_bfx_wide(int mode);      /* set the wide-pixel-tile mode alone (-1 = environment default); returns the previous one */
void bgs_conv1x1_bres_enable(int on);
int bgs_conv1x1_bres_last_launch(void);
void bgs_conv_dgrad_parity_enable(int on);
void bgs_conv2d_wgrad_bfx_enable(int on);
void bgs_conv_bf16s_tuning(int variant);

/* Wide-tile 1x1 kernel of the bf16x6 mode (csrc/conv_bfx_wide.hip: 128 x 128 tile, four M-stacked
 * waves, 24 MFMAs per wave and barrier; replaces the 64 x 64 operand ring on the wide-N 1x1 layers
 * of mmdet/models/backbones/resnet.py:220-266 / necks/fpn.py:101-141; bit-identical to it when K is
 * not sliced).  mode 0 = never | 1 = where measured faster (default) | 2 = every eligible layer;
 * nst 0 = auto | 2 | 3 ring stages; splitk -1 = auto | 1..16 K slices.
 * last_launch: bit 0 = the last bf16x6 1x1 launch took it; bits 4..7 ring stages; bits 8.. K slices. */
void bgs_conv_bfx_wide_tuning(int mode, int nst, int splitk);
int bgs_conv_bfx_wide_last_launch(void);
/* Round 6: the planes-in-LDS 1x1 kernel (csrc/conv1x1_planes.hip: 64 pixels x 256 or 128 channels per workgroup, the A
 * tile split ONCE per K chunk into bf16 planes in LDS, the MFMA phase free of VALU work, filter fragments by buffer loads;
 * bit-identical to the ring / wide kernels).  mode 0 off | 1 automatic (default; env BGS_BFX_PLANES, read at every call:
 * every eligible layer whose grid has at least one workgroup per CU) | 2 every eligible layer (1x1, stride 1 or 2, Cin %
 * 64 == 0, Cout % 128 == 0, fp32-faithful planes, no split-K, tensors < 2 GB; the ReLU-backward mask and the residual of the
 * data-gradient form are in its epilogue) | < 0: back to the environment's
 * value.  bgs_conv1x1_planes_last_launch: 0, or the channels per workgroup / 128 (1 | 2) when the last bf16x6 conv launch
 * took it.  Environment, read once: BGS_BFX_PLANES_NB = 1 / 2 forces the channels per workgroup, BGS_BFX_PLANES_ABLATE =
 * timing-only arms (tools/planes_ablate.py). */
void bgs_conv1x1_planes_enable(int mode);
int bgs_conv1x1_planes_last_launch(void);
/* Round 6: the 3x3 sibling for the small maps (csrc/conv3x3_planes.hip: 8 x 8 output pixels x 256 or 128 channels per
 * workgroup, the whole reduction in one workgroup — no K slices, no reduction launch; 10 x 10 patch planes in LDS per
 * 32-channel chunk, one barrier per 18 k steps, filter fragments by buffer loads; bit-identical to the halo kernel with ONE
 * K slice, i.e. within fp32 summation order of the sliced default).  First choice of bgs_conv3x3_halo_nhwc_f32_bfx unless
 * the halo tuning hook forces a slice count / variant / pixel tile.  mode 0 off | 1 automatic (env BGS_BFX_PLANES3, read
 * at every call: the layers whose halo plan slices K and whose own grid has a workgroup per CU) | 2 every eligible layer
 * (3x3 / stride 1 / pad 1, Cin % 32 == 0, Cout % 128 == 0, fp32-faithful planes, tensors < 2 GB; the ReLU-backward mask
 * of the data-gradient form is in its epilogue) | < 0: back to
 * the environment's value.  last_launch: 0, or the channels per workgroup / 128 of the last 3x3 launch that took it
 * (| 0x10: the stride-2 form, conv3x3s2_planes_bfx_kernel: forward 3x3 / stride 2 / pad 1 layers ahead of the operand ring —
 * the 17 x 17 patch as four parity sub-grids in LDS; same products as the ring in another fp32 summation order;
 * BGS_BFX_PLANES3_S2=0 switches that form alone off). */
void bgs_conv3x3_planes_enable(int mode);
int bgs_conv3x3_planes_last_launch(void);

/* Row-per-workgroup GroupSoftmax loss kernel (csrc/gs_loss.hip, bgs_gs_loss_fwd_bwd; the bandwidth-bound form
 * of gs_bbox_head_with0.py:147-186 for N beyond the fused head's 4096 rows): prefetch 0 = every row pays its own
 * global-memory round trip (the round-3 kernel) | 1 = the next row of a workgroup is fetched into registers under the
 * current row's sweeps | 2 / 3 / 4 = 1 + non-temporal row loads / gradient stores / both.  3 is the default
 * (N = 65,536: 4.5 -> 5.3 TB/s).  Bit-identical results in every mode. */
void bgs_gs_loss_tuning(int prefetch);
/* Round 6: prefetch 5 (the default) = the row-per-WAVE kernel (gs_loss_wavepriv_kernel: a wave owns a row in a private
 * LDS row, no workgroup barrier in the row loop) for 4096 < N < 12288 rows of 16-byte-aligned tables whose bins fit the
 * register sweep (8192 rows: 16.4 -> 14.2 us), mode 3 elsewhere (faster again from 16,384 rows); 6 / 7 = row-per-wave for
 * every N >= the threshold with plain / non-temporal row loads (A/B).  Gradient bit-identical to modes 0 - 4, the per-bin
 * losses are the same terms summed in a different order.  bgs_gs_loss_wavepriv_min_rows: the row threshold (< 0: default).
 * bgs_gs_merge_tuning (csrc/gs_merge.hip, _merge_score gs_bbox_head_with0.py:239-273): mode 1 (default) = 2 = row-per-wave
 * kernel with 16-byte score stores for N >= min_rows (< 0: default 4096; R = 65,536: 184 -> 124 - 132 us) | 3 = .. with
 * non-temporal row loads (A/B) | 0 = the 4-wave-per-row kernel.  Bit-identical scores in every mode. */
void bgs_gs_loss_wavepriv_min_rows(int rows);
void bgs_gs_merge_tuning(int mode, int min_rows);

#ifdef __cplusplus
}
#endif
#endif /* BGS_TUNING_H_ */
