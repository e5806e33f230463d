// ResNet stem in ONE launch on gfx950 (MI355X): 7x7 / stride 2 / pad 3 convolution of the NCHW image (3 -> 64
// channels, eval-BN folded) + ReLU + 3x3 / stride 2 / pad 1 max-pool, bf16x6 arithmetic (fp32-faithful: three-way
// bf16 split of both operands, six products on v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Replaces, for the frozen stem of every BAGS config (mmdet/models/backbones/resnet.py:522-533: conv1 -> norm1 ->
// relu -> maxpool), the chain  nchw_to_nhwc4 (17 us) -> implicit-GEMM conv with K = 7 * 7 * 4 (136 us, writes the
// 137 MB [2, 400, 672, 64] map) -> maxpool3x3s2 (40 us, re-reads it)  of the cfg[1] step (profiles/r8z): the conv
// output never leaves the CU.
//
// Workgroup = 4 waves = one 5 x 16 tile of POOLED pixels x 64 channels:
//   * the 11 x 33 conv outputs that the tile's 3x3 windows touch are the GEMM rows (363 of 384 = 12 MFMA row tiles; the
//     1.13x re-computation of the window overlap replaces the HBM round trip), 64 output channels = 2 column tiles;
//     wave w owns row tiles 3 w .. 3 w + 2 and BOTH column tiles (acc[3][2] = 96 accumulator registers): an A fragment
//     is split once for twelve MFMAs (a 2 x 2 wave grid splits every fragment twice: the first version of this kernel
//     was VALU-bound at 4800 VALU instructions per wave for 330 MFMAs, profiles/r9c);
//   * the 27 x 71 x 3 input patch is staged ONCE from the NCHW image (coalesced reads along W, zero padding of the
//     image border) and split ONCE into its three bf16 planes in LDS, [row][x][c0 c1 c2 0] (46.7 KB): an A fragment —
//     eight consecutive k of a conv pixel — is two neighbouring patch pixels = one aligned ds_read_b128 per plane,
//     because K is ordered (ky, kx, c padded to 4) with the eighth filter column zero: K = 7 x 32 = 14 MFMA steps.
//     (Versions 1 and 2 kept the patch in fp32 with K = (ky, 3 kx + c) — 11 steps — and split every fragment in
//     registers: 44 VALU instructions per fragment, each input value split ~12 times over; VALU and MFMA time ADD on
//     a SIMD, and the kernel ran VALU-bound: 135 / 105 us against 190 for the chain, profiles/r9c.)
//   * the filter (pre-split once: [plane][block][cout][8] bf16, 86 KB, L2-resident) is read straight into the MFMA's
//     B registers, one step ahead;
//   * epilogue, per 32-channel half: bias + ReLU, tile to LDS (aliasing the dead patch), 3x3 max over the conv pixels
//     that exist (a pixel outside the 400 x 672 map counts as 0: every pool window holds a real pixel and ReLU outputs
//     are >= 0, so 0 stands in for the pool's -inf padding), 16-byte NHWC stores.
// Arithmetic differs from the three-launch chain only in the ORDER of the fp32 accumulation (K order / padding):
// fp32-faithful either way (tests: == torch-CPU fp64 within 2e-6 of the map's scale, == the chain within 1e-5).
#include <stdlib.h>

#include "bgs_common.h"
#include "bfx_split.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kCout = 64;
constexpr int kBlocks = 28;                 // k blocks of 8 = two input pixels x (3 channels + 1 zero): 7 filter rows x 4 blocks (14 MFMA steps)
constexpr int PT_H = 5, PT_W = 16;          // pooled tile
constexpr int CT_H = 2 * PT_H + 1, CT_W = 2 * PT_W + 1, CT_PX = CT_H * CT_W;     // 11 x 33 = 363 conv pixels = 12 row tiles of 32 (384)
constexpr int IN_H = 2 * CT_H + 5, IN_W = 2 * CT_W + 5;                          // 27 x 71 input pixels
constexpr int IN_WP = IN_W + 1;             // 72 pixels per patch row (the zero-weight eighth filter column reads pixel 71)
constexpr int ROW_BYTES = IN_WP * 8;        // a patch pixel = 4 bf16 (c0, c1, c2, 0) = 8 bytes per plane
constexpr int PLANE_BYTES = IN_H * ROW_BYTES;                                     // 15,552
constexpr int PATCH_BYTES = 3 * PLANE_BYTES;                                      // 46,656
constexpr int OUT_LD = 32 + 4;              // conv tile in LDS, one 32-channel half at a time: [363][36] fp32 = 52,272 bytes
constexpr int OUT_BYTES = CT_PX * OUT_LD * 4;
constexpr int LDS_BYTES = OUT_BYTES > PATCH_BYTES ? OUT_BYTES : PATCH_BYTES;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// x (4 consecutive k) -> three planes of 4 packed bf16 each (the split of csrc/conv_bfx.hip)
__device__ __forceinline__ void split3(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
  const f32x4 r = {bfx_resid_lo(hi[0], v[0]), bfx_resid_hi(hi[0], v[1]), bfx_resid_lo(hi[1], v[2]), bfx_resid_hi(hi[1], v[3])};
  mid = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
  const f32x4 r2 = {bfx_resid_lo(mid[0], r[0]), bfx_resid_hi(mid[0], r[1]), bfx_resid_lo(mid[1], r[2]), bfx_resid_hi(mid[1], r[3])};
  lo = u32x2{pack_bf16(r2[0], r2[1]), pack_bf16(r2[2], r2[3])};
}

// w [64][7][7][cin_stride] fp32 (folded, channels 0..2 used) -> out [3 planes][28 blocks][64 cout][8] bf16:
// block = 4 ky + g, element j of it = filter column kx = 2 g + j / 4, channel c = j % 4 (kx = 7 and c = 3: zeros)
__global__ __launch_bounds__(256) void stem_split_weights_kernel(const float* __restrict__ w, int cin_stride,
                                                                 unsigned* __restrict__ out) {
  const int total = kBlocks * kCout * 4;                       // bf16 pairs per plane
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int jp = e & 3, cout = (e >> 2) % kCout, blk = (e >> 2) / kCout;
  const int ky = blk >> 2, kx = 2 * (blk & 3) + (jp >> 1);
  float v[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = (jp & 1) * 2 + u;
    v[u] = (kx < 7 && c < 3) ? w[((size_t)(cout * 7 + ky) * 7 + kx) * cin_stride + c] : 0.f;
  }
  const unsigned h = pack_bf16(v[0], v[1]);
  const float r0 = bfx_resid_lo(h, v[0]), r1 = bfx_resid_hi(h, v[1]);
  const unsigned m = pack_bf16(r0, r1);
  const unsigned l = pack_bf16(bfx_resid_lo(m, r0), bfx_resid_hi(m, r1));
  out[e] = h;
  out[total + e] = m;
  out[2 * total + e] = l;
}

struct StemArgs {
  const float* img;      // [N, 3, H, W]
  const unsigned* ws;    // split filter
  const float* bias;     // [64]
  float* out;            // [N, PH, PW, 64]
  int N, H, W, CH, CW, PH, PW;
  int tiles_y, tiles_x;
};

__global__ __launch_bounds__(kThreads, 2) void stem_conv7x7s2_relu_maxpool_kernel(StemArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-banded tile order: workgroup b runs on XCD b % 8 and walks its own band of the image
  const int tiles = p.N * p.tiles_y * p.tiles_x;
  const int chunk = (tiles + 7) / 8;
  const int t = (int)((blockIdx.x & 7) * chunk + (blockIdx.x >> 3));
  if (t >= tiles) return;
  const int n = t / (p.tiles_y * p.tiles_x);
  const int trem = t - n * (p.tiles_y * p.tiles_x);
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int py0 = ty * PT_H, px0 = tx * PT_W;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;              // conv pixel of tile row / column 0 (pool pad 1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;              // input pixel of patch row / column 0 (conv pad 3)

  // ---- the filter fragments of step 0 are requested before anything else: lane (frow, fk) holds the eight k of block
  //      2 s + fk for output channels frow and 32 + frow
  const int frow = lane & 31, fk = lane >> 5;
  const unsigned* wl = p.ws + ((size_t)fk * kCout + frow) * 4;                  // + (plane * 22 + 2 s) * 64 * 4 words (+ 32 * 4: column tile 1)
  constexpr int W_PLANE = kBlocks * kCout * 4, W_STEP = 2 * kCout * 4;
  u32x4 fbn[3][2];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int b = 0; b < 2; ++b) fbn[s3][b] = *reinterpret_cast<const u32x4*>(wl + s3 * W_PLANE + b * 32 * 4);

  // ---- stage the patch, split ONCE: every input value feeds ~12 conv pixels, and splitting it in the fragment path
  //      (the first two versions of this kernel) made the loop VALU-bound — VALU and MFMA time ADD on a SIMD
  //      (profiles/r9c: 3230 VALU instructions per wave beside 396 MFMAs: 17.8 + 17.5 us per round of workgroups).
  //      Three bf16 planes [row][x][c0 c1 c2 0]; a thread takes pixels x = lane (+ 64) of the rows wave, wave + 4, ...:
  //      three coalesced runs along W of the NCHW image per row; pixels outside the image (conv padding) are zeros.
  {
    const size_t plane = (size_t)p.H * p.W;
    const float* src = p.img + (size_t)n * 3 * plane;
    for (int row = wave; row < IN_H; row += 4) {                // wave-uniform
      const int iy = iy0 + row;
      const bool row_in = iy >= 0 && iy < p.H;
      const float* srow = src + (size_t)(row_in ? iy : 0) * p.W;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int x = lane + 64 * h;
        if (x >= IN_WP) continue;
        const int ix = ix0 + x;
        const bool in = row_in && x < IN_W && ix >= 0 && ix < p.W;
        const float v0 = in ? srow[ix] : 0.f, v1 = in ? srow[plane + ix] : 0.f, v2 = in ? srow[2 * plane + ix] : 0.f;
        u32x2 hh, mm, ll;
        split3(f32x4{v0, v1, v2, 0.f}, hh, mm, ll);
        unsigned char* d = lds + row * ROW_BYTES + x * 8;
        *reinterpret_cast<u32x2*>(d) = hh;
        *reinterpret_cast<u32x2*>(d + PLANE_BYTES) = mm;
        *reinterpret_cast<u32x2*>(d + 2 * PLANE_BYTES) = ll;
      }
    }
  }
  __syncthreads();

  // ---- fragment roles: lane frow of row tile mt = 3 wave + i owns conv pixel m = 32 mt + frow of the 11 x 33 tile (rows
  //      past 362: pixel 0, never stored); its eight k of block (ky, g) are patch pixels (2 cy + ky, 2 cx + 2 g .. + 1):
  //      one 16-byte aligned ds_read_b128 per plane
  int a_base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int m = (3 * wave + i) * 32 + frow;
    if (m >= CT_PX) m = 0;
    const int cy = m / CT_W, cx = m - cy * CT_W;
    a_base[i] = 2 * cy * ROW_BYTES + 2 * cx * 8;
    asm volatile("" : "+v"(a_base[i]));                        // (keep it in a register: hipcc re-derived the division per step)
  }
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

#pragma unroll
  for (int s = 0; s < kBlocks / 2; ++s) {
    // this step's filter fragments (requested one step ago) and the next step's request
    bf16x8 fb[3][2];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[s3][b] = __builtin_bit_cast(bf16x8, fbn[s3][b]);
    if (s + 1 < kBlocks / 2) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          fbn[s3][b] = *reinterpret_cast<const u32x4*>(wl + s3 * W_PLANE + (s + 1) * W_STEP + b * 32 * 4);
    }
    // block of this half-wave: 2 s + fk -> (ky, g); byte offset inside a plane
    const int blk0 = 2 * s, blk1 = 2 * s + 1;
    const int off0 = (blk0 >> 2) * ROW_BYTES + (blk0 & 3) * 16, off1 = (blk1 >> 2) * ROW_BYTES + (blk1 & 3) * 16;
    const int koff = fk ? off1 : off0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      bf16x8 fa[3];
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
        fa[s3] = *reinterpret_cast<const bf16x8*>(lds + a_base[i] + koff + s3 * PLANE_BYTES);
#pragma unroll
      for (int tt = 2; tt >= 0; --tt)
#pragma unroll
        for (int j = 0; j <= tt; ++j)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[tt - j][b], acc[i][b], 0, 0, 0);
    }
  }

  // ---- epilogue, one 32-channel half at a time: bias + ReLU, tile to LDS (over the patch: every wave is done reading),
  //      3x3 max over the conv pixels that exist (a pixel outside the conv map counts as 0: every window holds a real
  //      pixel and ReLU outputs are >= 0, so 0 stands in for the pool's -inf padding), 16-byte NHWC stores
  float* tile = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    __syncthreads();
    {
      const float b = p.bias ? p.bias[nh * 32 + (lane & 31)] : 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (3 * wave + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < CT_PX) tile[m * OUT_LD + (lane & 31)] = fmaxf(acc[i][nh][r] + b, 0.f);
        }
    }
    __syncthreads();
    const int q = tid & 7;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pp = (tid >> 3) + 32 * i;
      if (pp >= PT_H * PT_W) continue;
      const int ppy = pp >> 4, ppx = pp & 15;
      const int py = py0 + ppy, px = px0 + ppx;
      if (py >= p.PH || px >= p.PW) continue;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int gy = cy0 + 2 * ppy + dy;
        if (gy < 0 || gy >= p.CH) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int gx = cx0 + 2 * ppx + dx;
          if (gx < 0 || gx >= p.CW) continue;
          const f32x4 u = *reinterpret_cast<const f32x4*>(tile + ((2 * ppy + dy) * CT_W + 2 * ppx + dx) * OUT_LD + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], u[e]);
        }
      }
      *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.PH + py) * p.PW + px) * kCout + nh * 32 + 4 * q) = v;
    }
  }
}

}  // namespace

extern "C" size_t bgs_stem_fused_weight_bytes(void) { return (size_t)3 * kBlocks * kCout * 8 * sizeof(__bf16); }

extern "C" int bgs_stem_fused_split_weights(const float* w, int cin_stride, void* out, bgs_stream_t stream) {
  if (!w || !out || cin_stride < 3) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)out % 16 != 0) return BGS_ERR_INVALID_ARG;
  const int total = kBlocks * kCout * 4;
  hipLaunchKernelGGL(stem_split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, cin_stride, reinterpret_cast<unsigned*>(out));
  BGS_RETURN_LAUNCH_STATUS();
}

extern "C" int bgs_stem_conv7x7s2_relu_maxpool_nchw_f32(const float* img, const void* wsplit, const float* bias,
                                                        float* out, int N, int H, int W, bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0) return BGS_ERR_INVALID_ARG;
  if (!img || !wsplit || !out) return BGS_ERR_INVALID_ARG;
  if ((uintptr_t)wsplit % 16 != 0 || (uintptr_t)out % 16 != 0) return BGS_ERR_UNSUPPORTED;
  StemArgs p;
  p.img = img; p.ws = reinterpret_cast<const unsigned*>(wsplit); p.bias = bias; p.out = out;
  p.N = N; p.H = H; p.W = W;
  p.CH = (H + 6 - 7) / 2 + 1;
  p.CW = (W + 6 - 7) / 2 + 1;
  p.PH = (p.CH + 2 - 3) / 2 + 1;
  p.PW = (p.CW + 2 - 3) / 2 + 1;
  if (p.CH <= 0 || p.CW <= 0 || p.PH <= 0 || p.PW <= 0) return BGS_ERR_INVALID_ARG;
  if ((long long)N * 3 * H * W > 0x7fffffffLL * 4LL) return BGS_ERR_UNSUPPORTED;
  p.tiles_y = (p.PH + PT_H - 1) / PT_H;
  p.tiles_x = (p.PW + PT_W - 1) / PT_W;
  const long long tiles = (long long)N * p.tiles_y * p.tiles_x;
  if (tiles > 0x7fffffffLL / 8) return BGS_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)(8 * ((tiles + 7) / 8));
  bgs_internal_census_bump(BGS_CENSUS_STEM_FUSED);
  hipLaunchKernelGGL(stem_conv7x7s2_relu_maxpool_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, p);
  BGS_RETURN_LAUNCH_STATUS();
}
