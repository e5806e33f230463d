// 3x3 / stride 1 / pad 1 implicit-GEMM convolution with a halo-resident A operand, fp32 MFMA,
// NHWC, gfx950.  Design: DESIGN.md appendix A.  First version: used for the large-M layers
// (functional._use_halo_kernel); BGS_CONV_HALO=0|1 overrides.
//
// The general kernel (conv_igemm.hip) fetches every input pixel once per filter tap: nine global
// loads of a [pixels x BK] tile per channel chunk, 9x the input over the fabric
// (profiles/r3l_pmc_conv.md).  Here the workgroup's 8 x 16 output pixels plus their one-pixel halo
// (10 x 18 pixels x 16 channels) are staged in LDS ONCE per channel chunk and the nine taps read
// shifted rows of that patch; only the filter slice (B) is streamed per tap as before.
//
//   workgroup = 256 threads = 2 x 2 waves; BM = 128 pixels (8 x 16), BN = 128 channels, BK = 16;
//   wave tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x2_f32 accumulators;
//   loop: channel chunk (Cin / 16) outer, tap (9) inner — the same number of MFMA steps;
//   LDS rows keep the even/odd k-slot permutation of conv_igemm.hip (a lane's 16-byte fragment
//   feeds four consecutive MFMAs).
// Epilogue: bias + optional ReLU (what the 3x3 layers of the BAGS detectors need).
#include "bgs_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kT = 256;
constexpr int TH = 8, TW = 16, BM = TH * TW, BN = 128, BK = 16, LDK = BK + 4;
constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;      // 180 patch pixels
constexpr int AQ = PROWS * (BK / 4);                           // 720 16-byte quads per patch
constexpr int AQ_PER_THREAD = (AQ + kT - 1) / kT;              // 3

struct HaloArgs {
  const float* x;      // [N, H, W, Cin]
  const float* w;      // [Cout, 3, 3, Cin]
  const float* bias;   // [Cout] or null
  float* y;            // [N, H, W, Cout]
  int N, H, W, Cin, Cout, relu;
  int tiles_y, tiles_x, tiles_m, tiles_n, chunk;   // chunk: XCD-banded order as in conv_igemm.hip
};

__global__ __launch_bounds__(kT) void conv3x3_halo_f32_kernel(HaloArgs p) {
  __shared__ __attribute__((aligned(16))) float As[2][PROWS * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int vtile = p.chunk ? (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (vtile >= p.tiles_m * p.tiles_n) return;                  // workgroup-uniform
  const int tm = vtile / p.tiles_n, tn = vtile - tm * p.tiles_n;
  const int n = tm / (p.tiles_y * p.tiles_x);
  const int trem = tm - n * (p.tiles_y * p.tiles_x);
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int h0 = ty * TH - 1, w0 = tx * TW - 1;                // input coords of patch (0, 0)
  const int n0 = tn * BN;
  const int K = 9 * p.Cin;

  // ---- staging roles
  // A: patch quads idx = tid + 256 q;  prow = idx / 4, kq = idx % 4
  const float* a_src[AQ_PER_THREAD];
  int a_dst[AQ_PER_THREAD];
  bool a_use[AQ_PER_THREAD], a_in[AQ_PER_THREAD];
#pragma unroll
  for (int q = 0; q < AQ_PER_THREAD; ++q) {
    const int idx = tid + kT * q;
    a_use[q] = idx < AQ;
    const int prow = a_use[q] ? idx >> 2 : 0, kq = idx & 3;
    const int pr = prow / PW, pc = prow - pr * PW;
    const int hi = h0 + pr, wi = w0 + pc;
    a_in[q] = a_use[q] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    a_src[q] = p.x + (((size_t)n * p.H + (a_in[q] ? hi : 0)) * p.W + (a_in[q] ? wi : 0)) * p.Cin + kq * 4;
    a_dst[q] = prow * LDK + kq * 2;
  }
  // B: rows brow = tid / 4 + 64 q (q = 0, 1), kq = tid % 4
  const int bkq = tid & 3;
  const float* b_src[2];
  bool b_ok[2];
  int b_dst[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int brow = (tid >> 2) + 64 * q;
    b_ok[q] = n0 + brow < p.Cout;
    b_src[q] = p.w + (size_t)(b_ok[q] ? n0 + brow : 0) * K + bkq * 4;
    b_dst[q] = brow * LDK + bkq * 2;
  }

  f32x4 ra[AQ_PER_THREAD], rb[2];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < AQ_PER_THREAD; ++q) {
      ra[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_in[q]) ra[q] = *reinterpret_cast<const f32x4*>(a_src[q] + chunk * BK);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int q = 0; q < AQ_PER_THREAD; ++q) {
      if (!a_use[q]) continue;
      float* d = &As[buf][a_dst[q]];
      *reinterpret_cast<f32x2*>(d) = f32x2{ra[q][0], ra[q][2]};
      *reinterpret_cast<f32x2*>(d + BK / 2) = f32x2{ra[q][1], ra[q][3]};
    }
  };
  auto load_b = [&](int chunk, int tap) {
    const int kg = tap * p.Cin + chunk * BK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      rb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b_ok[q]) rb[q] = *reinterpret_cast<const f32x4*>(b_src[q] + kg);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float* d = &Bs[buf][b_dst[q]];
      *reinterpret_cast<f32x2*>(d) = f32x2{rb[q][0], rb[q][2]};
      *reinterpret_cast<f32x2*>(d + BK / 2) = f32x2{rb[q][1], rb[q][3]};
    }
  };

  // ---- fragment roles: lane frow of sub-tile a owns pixel m = 64 wm + 32 a + frow
  const int frow = lane & 31, fk = lane >> 5;
  int a_row0[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int m = wm * 64 + a * 32 + frow;
    a_row0[a] = (m >> 4) * PW + (m & 15);                      // patch row of tap (0, 0)
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunks = p.Cin / BK;
  const int nsteps = nchunks * 9;
  load_a(0);
  load_b(0, 0);
  store_a(0);
  store_b(0);
  __syncthreads();
  int chunk = 0, tap = 0;
  for (int t = 0; t < nsteps; ++t) {
    const int bbuf = t & 1, abuf = chunk & 1;
    const bool more = t + 1 < nsteps;
    int nchunk = chunk, ntap = tap + 1;
    if (ntap == 9) {
      ntap = 0;
      ++nchunk;
    }
    if (more) load_b(nchunk, ntap);                            // next filter slice in flight
    if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);    // next patch: held in registers

    // fragments of this step: both halves (k = 0..7 / 8..15 of the chunk), then 32 MFMAs
    const int tap_off = ((tap / 3) * PW + (tap % 3)) * LDK;
    f32x4 fa[2][2], fb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[h][a] = *reinterpret_cast<const f32x4*>(&As[abuf][a_row0[a] * LDK + tap_off + fk * (BK / 2) + 4 * h]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
        fb[h][b] = *reinterpret_cast<const f32x4*>(&Bs[bbuf][(wn * 64 + b * 32 + frow) * LDK + fk * (BK / 2) + 4 * h]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][a][j], fb[h][b][j], acc[a][b], 0, 0, 0);

    if (more) store_b(bbuf ^ 1);
    if (tap == 8 && chunk + 1 < nchunks) store_a(abuf ^ 1);   // loaded at tap 0 of this chunk
    __syncthreads();
    tap = ntap;
    chunk = nchunk;
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = wm * 64 + a * 32 + i;
      const int ho = ty * TH + (m >> 4), wo = tx * TW + (m & 15);
      if (ho >= p.H || wo >= p.W) continue;
      float* yrow = p.y + (((size_t)n * p.H + ho) * p.W + wo) * p.Cout;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = n0 + wn * 64 + b * 32 + (lane & 31);
        if (j >= p.Cout) continue;
        float v = acc[a][b][r];
        if (p.bias) v += p.bias[j];
        if (p.relu) v = fmaxf(v, 0.f);
        yrow[j] = v;
      }
    }
  }
}

}  // namespace

extern "C" int bgs_conv3x3_halo_nhwc_f32(const float* x, const float* w, const float* bias, float* y,
                                         int N, int H, int W, int Cin, int Cout, int relu,
                                         bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return BGS_ERR_INVALID_ARG;
  if (!x || !w || !y) return BGS_ERR_INVALID_ARG;
  if (Cin % BK != 0) return BGS_ERR_UNSUPPORTED;
  if ((uintptr_t)x % 16 != 0 || (uintptr_t)w % 16 != 0) return BGS_ERR_UNSUPPORTED;
  HaloArgs p;
  p.x = x;
  p.w = w;
  p.bias = bias;
  p.y = y;
  p.N = N;
  p.H = H;
  p.W = W;
  p.Cin = Cin;
  p.Cout = Cout;
  p.relu = relu;
  p.tiles_y = (H + TH - 1) / TH;
  p.tiles_x = (W + TW - 1) / TW;
  p.tiles_m = N * p.tiles_y * p.tiles_x;
  p.tiles_n = (Cout + BN - 1) / BN;
  p.chunk = (p.tiles_m * p.tiles_n + 7) / 8;
  hipLaunchKernelGGL(conv3x3_halo_f32_kernel, dim3((unsigned)(8 * p.chunk)), dim3(kT), 0,
                     (hipStream_t)stream, p);
  BGS_RETURN_LAUNCH_STATUS();
}
