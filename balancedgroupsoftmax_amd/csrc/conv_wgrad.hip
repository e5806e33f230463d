// Weight gradient of the NHWC implicit-GEMM convolution (and of nn.Linear) on the fp32 matrix
// cores of gfx950 — the `selectp = 0` (train everything) backward pass of the BAGS detector:
// what the reference gets from cuDNN's backward-filter through autograd for
//   ResNet bottlenecks     mmdet/models/backbones/resnet.py:220-266
//   FPN laterals / outputs mmdet/models/necks/fpn.py:101-141
//   RPN head               mmdet/models/anchor_heads/rpn_head.py:30-35
//   RoI head FCs           mmdet/models/bbox_heads/convfc_bbox_head.py:132-168
//
//   dW[co][r][s][ci] = sum_m dy[m][co] * x[n(m), ho(m)*stride - pad + r, wo(m)*stride - pad + s][ci]
//
// GEMM view: D[co][k] = sum_m A[m][co] * B[m][k], k = (r, s, ci): BOTH operands are stored with
// the reduction index m as the slow axis and the output index contiguous.  That is the layout
// v_mfma_f32_32x32x2_f32 wants if a lane's output rows are INTERLEAVED: lane l holds, for the
// two m of a K step (m parity = l >> 5), MB consecutive co starting at MB * (l & 31) — one
// ds_read_b64/b128 from the [m][co] tile feeds MB row tiles at once, and the global 16-byte loads
// go to LDS untransposed.  D row i of row tile t is co = co0 + MB * i + t; likewise for k.
//
// The reduction (M = N*Ho*Wo, up to 134,400 at cfg[1]) is split over gridDim.z; partial tiles go
// to a workspace [splits][Cout][K] and are summed in a fixed order by wgrad_reduce_kernel
// (bitwise reproducible, no atomics), which can also accumulate into dW (weights shared across
// the five RPN levels).
#include <stdlib.h>

#include "bgs_common.h"
#include "bfx_split.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kBKM = 16;  // reduction rows (m) per stage

struct WgradArgs {
  const float* x;   // [N, H, W, Cin]
  const float* dy;  // [N, Ho, Wo, Cout]
  float* out;       // workspace [splits][Cout][K] or dW itself when splits == 1 && !accumulate
  float* db_part;   // null, or [splits][Cout]: per-split column sums of dy (bias gradient), produced
                    // by the k-tile-0 workgroups from the dy tiles they stage anyway
  int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
  int M, K, m_per_split;
  // XCD-aware order (bf16x6 kernel): 1-D launch of 8 * chunk workgroups; workgroup b runs on XCD b % 8 and
  // takes tile (b % 8) * chunk + b / 8 of the (co tile, k tile, M slice) list, M slice slowest — all tiles of
  // one M slice (they re-read the same dy / x rows: 18 k tiles x 2 co tiles x 9 taps on the FPN convs) share
  // ONE XCD's L2 instead of pulling the slice through the fabric into all eight.  chunk == 0: 3-D grid.
  int gx, gy, gz, chunk;
};

template <int T>
struct FragVec;
template <>
struct FragVec<1> {
  typedef float type;
};
template <>
struct FragVec<2> {
  typedef f32x2 type;
};
template <>
struct FragVec<4> {
  typedef f32x4 type;
};

template <int T>
__device__ __forceinline__ float frag_get(const typename FragVec<T>::type& v, int t) {
  return v[t];
}
template <>
__device__ __forceinline__ float frag_get<1>(const float& v, int) {
  return v;
}

// Workgroup = 4 waves (2 x 2); wave tile = (32*MB co) x (32*NB k); block tile 64*MB x 64*NB.
template <int MB, int NB>
__global__ __launch_bounds__(kThreads) void conv_wgrad_f32_kernel(WgradArgs p) {
  constexpr int BM = 64 * MB, BN = 64 * NB;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int QA = BM / 4, QB = BN / 4;          // 16-byte quads per m row
  constexpr int PA = (kBKM * QA) / kThreads;        // quads per thread per stage (1 or 2)
  constexpr int PB = (kBKM * QB) / kThreads;
  constexpr int RA = kThreads / QA, RB = kThreads / QB;  // m rows covered per pass
  __shared__ __attribute__((aligned(16))) float As[2][kBKM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBKM * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int co0 = blockIdx.x * BM, k0 = blockIdx.y * BN;
  const int m_begin = blockIdx.z * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // ---- A staging: quad qa of m row (ra + RA*q)
  const int qa = tid % QA, ra = tid / QA;
  const bool a_col_ok = co0 + qa * 4 < p.Cout;     // Cout % 4 == 0: whole quad in or out
  // ---- B staging: quad qb (fixed filter tap / input channel for the whole kernel)
  const int qb = tid % QB, rb = tid / QB;
  const int kg = k0 + qb * 4;
  const bool b_col_ok = kg < p.K;
  int kr, ks, kc;
  {
    const int kk = b_col_ok ? kg : 0;
    const int rs = kk / p.Cin;
    kc = kk - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  // (n, ho, wo) of this thread's B rows, advanced incrementally by kBKM per stage
  int bn_[PB], bho[PB], bwo[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int m = m_begin + rb + RB * q;
    const int hw = p.Ho * p.Wo;
    bn_[q] = m / hw;
    const int rem = m - bn_[q] * hw;
    bho[q] = rem / p.Wo;
    bwo[q] = rem - bho[q] * p.Wo;
  }

  f32x4 va[PA], vb[PB];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool want_db = p.db_part != nullptr && blockIdx.y == 0;
  int m_stage = m_begin;
  auto load_stage = [&]() {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int m = m_stage + ra + RA * q;
      va[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_col_ok && m < m_end)
        va[q] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.Cout + co0 + qa * 4);
      if (want_db) bsum += va[q];
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int m = m_stage + rb + RB * q;
      const int hi = bho[q] * p.stride - p.pad + kr, wi = bwo[q] * p.stride - p.pad + ks;
      vb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b_col_ok && m < m_end && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
        vb[q] = *reinterpret_cast<const f32x4*>(
            p.x + (((size_t)bn_[q] * p.H + hi) * p.W + wi) * p.Cin + kc);
      // advance this row by kBKM output pixels
      bwo[q] += kBKM;
      while (bwo[q] >= p.Wo) {
        bwo[q] -= p.Wo;
        if (++bho[q] == p.Ho) {
          bho[q] = 0;
          ++bn_[q];
        }
      }
    }
    m_stage += kBKM;
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      *reinterpret_cast<f32x4*>(&As[buf][(ra + RA * q) * LDA + qa * 4]) = va[q];
#pragma unroll
    for (int q = 0; q < PB; ++q)
      *reinterpret_cast<f32x4*>(&Bs[buf][(rb + RB * q) * LDB + qb * 4]) = vb[q];
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nst = (m_end - m_begin + kBKM - 1) / kBKM;
  const int fi = lane & 31, fk = lane >> 5;
  typedef typename FragVec<MB>::type AV;
  typedef typename FragVec<NB>::type BV;
  if (nst > 0) {
    load_stage();
    store_stage(0);
  }
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) load_stage();
    const float* Ab = &As[buf][fk * LDA + wm * 32 * MB + fi * MB];
    const float* Bb = &Bs[buf][fk * LDB + wn * 32 * NB + fi * NB];
#pragma unroll
    for (int kk = 0; kk < kBKM / 2; ++kk) {
      const AV av = *reinterpret_cast<const AV*>(Ab + 2 * kk * LDA);
      const BV bv = *reinterpret_cast<const BV*>(Bb + 2 * kk * LDB);
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_get<MB>(av, a), frag_get<NB>(bv, b),
                                                            acc[a][b], 0, 0, 0);
    }
    if (st + 1 < nst) store_stage(buf ^ 1);
    __syncthreads();
  }

  if (want_db) {   // block-uniform; the last loop iteration ended with a barrier: As is free
    f32x4* red = reinterpret_cast<f32x4*>(&As[0][0]);     // [RA][QA] quads (RA*QA = 256)
    red[ra * QA + qa] = bsum;
    __syncthreads();
    if (ra == 0 && a_col_ok) {
      f32x4 t = red[qa];
      for (int r = 1; r < RA; ++r) t += red[r * QA + qa];
      *reinterpret_cast<f32x4*>(p.db_part + (size_t)blockIdx.z * p.Cout + co0 + qa * 4) = t;
    }
  }

  // ---- write the partial tile: D row i of row tile a -> co = co0 + wm*32*MB + MB*i + a,
  //      D col (lane & 31) of col tile b -> k = k0 + wn*32*NB + NB*(lane & 31) + b
  float* out = p.out + (size_t)blockIdx.z * p.Cout * p.K;
  const int kcol = k0 + wn * 32 * NB + NB * (lane & 31);
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int co = co0 + wm * 32 * MB + MB * i + a;
      if (co >= p.Cout) continue;
      float* row = out + (size_t)co * p.K;
      if (NB == 2 && kcol + 1 < p.K) {
        *reinterpret_cast<f32x2*>(row + kcol) = f32x2{acc[a][0][r], acc[a][NB - 1][r]};
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (kcol + b < p.K) row[kcol + b] = acc[a][b][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix cores with fp32-faithful results ("bf16x6", see
// conv_bfx.hip): D[co][k] = sum_m dy[m][co] * x[m][k] with every product formed from the exact
// three-way bf16 split of BOTH operands (six v_mfma_f32_32x32x16_bf16 per 16 reduction rows instead
// of eight v_mfma_f32_32x32x2_f32 per 16: 6/16 of the matrix-pipe time).
// The MFMA wants, per lane, 8 consecutive reduction indices m of ONE output row (co or k) in 16
// contiguous bytes, but both operands are stored m-major.  The transpose happens in registers on the
// way to LDS: a staging thread loads a 4 (m) x 4 (co or k) block with four 16-byte loads, and for
// each of its 4 columns splits the 4 m-values into the three bf16 planes and writes them with one
// ds_write_b64 per plane into the [row][16 m] image (48-byte rows as in conv_igemm_bfx_kernel) —
// 12 LDS writes per 64 loaded bytes, the same ratio as the forward kernel's A staging, no
// ds_read_tr / ds_bpermute.  Threads 0-127 stage dy (and accumulate the bias gradient), 128-255 the
// im2col rows of x.  128 x 128 tile, 2 x 2 waves, 24 MFMAs per wave and stage, LDS double-buffered
// (72 KB: two workgroups per CU); reduction split over gridDim.z and summed in a fixed order as above.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_wgrad_zero[4];

__device__ __forceinline__ unsigned wg_pack_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float wg_bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float wg_bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void wg_split3(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{wg_pack_bf16(v[0], v[1]), wg_pack_bf16(v[2], v[3])};
  const f32x4 r = {bfx_resid_lo(hi[0], v[0]), bfx_resid_hi(hi[0], v[1]), bfx_resid_lo(hi[1], v[2]),
                   bfx_resid_hi(hi[1], v[3])};
  mid = u32x2{wg_pack_bf16(r[0], r[1]), wg_pack_bf16(r[2], r[3])};
  const f32x4 r2 = {bfx_resid_lo(mid[0], r[0]), bfx_resid_hi(mid[0], r[1]), bfx_resid_lo(mid[1], r[2]),
                    bfx_resid_hi(mid[1], r[3])};
  lo = u32x2{wg_pack_bf16(r2[0], r2[1]), wg_pack_bf16(r2[2], r2[3])};
}

// NS = 3: fp32-faithful; NS = 1: operands rounded to bf16 (the bf16 mode of cfg[4])
// ABL (only instantiated != 0 under -DBGS_ABLATE, tools/wgrad_ablate.py): timing-only variants that drop one
// component of the loop — 1: MFMAs, 2: global loads after the first stage, 4: the operand split, 8: LDS stores,
// 16: fragment ds_reads.
// NBUF = 2: two LDS stage buffers (74 KB: two workgroups per CU), one barrier per stage.  NBUF = 1: one buffer
// (37 KB: THREE workgroups per CU at 160 VGPRs), two barriers per stage — the component ablation (profiles/r5m_wgrad_ablate.txt)
// shows the operand loads (0.70 ms alone on fpn.out0) and the MFMAs (0.64 ms alone) NOT overlapping (1.50 ms
// together): with two workgroups per CU the memory path idles whenever both are multiplying.
template <int NS, int ABL = 0, int NBUF = 2>
__global__ __launch_bounds__(kThreads, NBUF == 1 ? 3 : 2) void conv_wgrad_bfx_kernel(WgradArgs p) {
  constexpr int BT = 128;                       // tile: 128 co x 128 k
  constexpr int LDR = 48;                       // LDS row: 16 m as bf16 (32 B) + 16 B pad
  constexpr int PLANE = BT * LDR;               // one operand plane
  constexpr int BUF = 2 * NS * PLANE;           // A planes, then B planes
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.chunk) {
    const int v = (int)((blockIdx.x & 7) * p.chunk + (blockIdx.x >> 3));
    if (v >= p.gx * p.gy * p.gz) return;          // workgroup-uniform
    bx = v % p.gx;
    const int t = v / p.gx;
    by = t % p.gy;
    bz = t / p.gy;
  }
  const int co0 = bx * BT, k0 = by * BT;
  const int m_begin = bz * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // ---- staging roles: part 0 (threads 0-127) = dy, part 1 = x; block (mg, q): m rows 4 mg .. 4 mg + 3
  //      of the stage, column quad q (4 consecutive co, or 4 consecutive k = channels of one tap)
  const bool is_b = tid >= 128;
  const int u = tid & 127;
  const int mg = u & 3, cq = u >> 2;
  const int colq = (is_b ? k0 : co0) + cq * 4;
  const bool col_ok = colq < (is_b ? p.K : p.Cout);     // Cout % 4 == 0, K % 4 == 0: whole quad in or out
  int kr = 0, ks = 0, kc = 0;
  if (is_b && col_ok) {
    const int rs = colq / p.Cin;
    kc = colq - rs * p.Cin;
    kr = rs / p.S;
    ks = rs - kr * p.S;
  }
  // (n, ho, wo) of this thread's 4 rows (x part), advanced by kBKM per stage
  int bn_[4], bho[4], bwo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_begin + 4 * mg + i;
    const int hw = p.Ho * p.Wo;
    bn_[i] = m / hw;
    const int rem = m - bn_[i] * hw;
    bho[i] = rem / p.Wo;
    bwo[i] = rem - bho[i] * p.Wo;
  }
  const int dst = (is_b ? NS * PLANE : 0) + (cq * 4) * LDR + mg * 8;     // + j * LDR + s * PLANE

  f32x4 v[4];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool want_db = p.db_part != nullptr && by == 0;
  int m_stage = m_begin;
  auto load_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_stage + 4 * mg + i;
      const float* src = g_wgrad_zero;
      if (!is_b) {
        if (col_ok && m < m_end) src = p.dy + (size_t)m * p.Cout + colq;
      } else {
        const int hi = bho[i] * p.stride - p.pad + kr, wi = bwo[i] * p.stride - p.pad + ks;
        if (col_ok && m < m_end && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
          src = p.x + (((size_t)bn_[i] * p.H + hi) * p.W + wi) * p.Cin + kc;
        bwo[i] += kBKM;
        while (bwo[i] >= p.Wo) {
          bwo[i] -= p.Wo;
          if (++bho[i] == p.Ho) {
            bho[i] = 0;
            ++bn_[i];
          }
        }
      }
      v[i] = *reinterpret_cast<const f32x4*>(src);
    }
    m_stage += kBKM;
  };
  auto store_stage = [&](int buf) {
    unsigned char* base = lds + buf * BUF + dst;
    if (want_db && !is_b) bsum += (v[0] + v[1]) + (v[2] + v[3]);
    if (ABL & 8) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 h, m, l;
      if (ABL & 4) {
        h = u32x2{__float_as_uint(v[0][j]), __float_as_uint(v[1][j])};
        m = u32x2{__float_as_uint(v[2][j]), __float_as_uint(v[3][j])};
        l = h;
      } else {
        wg_split3(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]}, h, m, l);
      }
      *reinterpret_cast<u32x2*>(base + j * LDR) = h;
      if (NS >= 2) *reinterpret_cast<u32x2*>(base + j * LDR + PLANE) = m;
      if (NS >= 3) *reinterpret_cast<u32x2*>(base + j * LDR + 2 * PLANE) = l;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  const int a_frag = (wm * 64 + frow) * LDR + fk * 16;
  const int b_frag = NS * PLANE + (wn * 64 + frow) * LDR + fk * 16;
  const int nst = (m_end - m_begin + kBKM - 1) / kBKM;
  if (nst > 0) {
    load_stage();
    store_stage(0);
  }
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = NBUF == 2 ? (st & 1) : 0;
    const bool more = st + 1 < nst;
    if (more && !((ABL & 2) && st > 0)) load_stage();   // global loads of stage st + 1 in flight under the MFMAs
    const unsigned char* base = lds + buf * BUF;
    bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (ABL & 16) fa[s][a] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)st, (unsigned)s, (unsigned)a, 0u});
        else fa[s][a] = *reinterpret_cast<const bf16x8*>(base + a_frag + s * PLANE + a * 32 * LDR);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (ABL & 16) fb[s][b] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)st, (unsigned)s, (unsigned)b, 1u});
        else fb[s][b] = *reinterpret_cast<const bf16x8*>(base + b_frag + s * PLANE + b * 32 * LDR);
      }
    }
    // products (i, j) with i + j <= NS - 1, smallest terms first
#pragma unroll
    for (int t = NS - 1; t >= 0; --t)
#pragma unroll
      for (int i = 0; i <= t; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
          {
            if (ABL & 1) acc[a][b][(i + t) & 15] += __builtin_bit_cast(u32x4, fa[i][a])[0] * 1e-30f +
                                                    __builtin_bit_cast(u32x4, fb[t - i][b])[1] * 1e-30f;
            else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][a], fb[t - i][b], acc[a][b], 0, 0, 0);
          }
    if (NBUF == 1) __syncthreads();             // every wave has read the (only) buffer
    if (more) store_stage(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  if (want_db) {   // block-uniform; the loop ended with a barrier: the LDS is free
    f32x4* red = reinterpret_cast<f32x4*>(lds);      // [4 mg][32 cq]
    if (!is_b) red[mg * 32 + cq] = bsum;
    __syncthreads();
    if (!is_b && mg == 0 && col_ok) {
      const f32x4 t = (red[cq] + red[32 + cq]) + (red[64 + cq] + red[96 + cq]);
      *reinterpret_cast<f32x4*>(p.db_part + (size_t)bz * p.Cout + colq) = t;
    }
  }

  // ---- partial tile: C/D layout col = lane & 31 (k), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co)
  float* out = p.out + (size_t)bz * p.Cout * p.K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int co = co0 + wm * 64 + a * 32 + i;
      if (co >= p.Cout) continue;
      float* row = out + (size_t)co * p.K;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int k = k0 + wn * 64 + b * 32 + (lane & 31);
        if (k < p.K) row[k] = acc[a][b][r];
      }
    }
}

// dW[i] (+)= scale-free sum over the splits, fixed order; the last `gb` workgroups of the launch do the same
// for the bias-gradient partials (ONE launch for both: 63 launches less per `selectp=0` backward).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ dw, size_t n4,
                                                           const float* __restrict__ ws_b,
                                                           float* __restrict__ db, size_t c4, unsigned gb,
                                                           int splits, int accumulate) {
  const unsigned gw = gridDim.x - gb;
  if (blockIdx.x >= gw) {
    for (size_t i = (size_t)(blockIdx.x - gw) * 256 + threadIdx.x; i < c4; i += (size_t)gb * 256) {
      f32x4 s = *reinterpret_cast<const f32x4*>(ws_b + i * 4);
      for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(ws_b + (z * c4 + i) * 4);
      if (accumulate) s += *reinterpret_cast<const f32x4*>(db + i * 4);
      *reinterpret_cast<f32x4*>(db + i * 4) = s;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gw * 256) {
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + i * 4);
    for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(ws + (z * n4 + i) * 4);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(dw + i * 4);
    *reinterpret_cast<f32x4*>(dw + i * 4) = s;
  }
}

// the reduction launch shared by both kernels' entry points: weight slabs (unless the kernel wrote dw
// directly) and bias partials
int launch_wgrad_reduce(const float* ws, float* dw, size_t n, bool direct, const float* db_part, float* db,
                        int Cout, int splits, int accumulate, hipStream_t st) {
  const size_t n4 = direct ? 0 : n / 4;     // Cin % 4 == 0 -> n % 4 == 0
  size_t g = (n4 + 255) / 256;
  if (g > 4096) g = 4096;
  const size_t c4 = db ? (size_t)Cout / 4 : 0;
  const unsigned gb = (unsigned)((c4 + 255) / 256);
  if (g + gb == 0) return BGS_OK;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g + gb), dim3(256), 0, st, ws, dw, n4, db_part, db,
                     c4, gb, splits, accumulate);
  return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH;
}

struct WgradPlan {
  int tile;     // 22 / 11
  int splits, m_per_split;
};

WgradPlan plan_wgrad(int M, int Cout, int K, bool bfx = false) {
  WgradPlan pl;
  pl.tile = (Cout >= 128 && K >= 128) ? 22 : 11;
  if (const char* e = getenv("BGS_WGRAD_TILE")) {
    const int f = atoi(e);
    if (f == 22 || f == 11) pl.tile = f;
  }
  if (bfx) pl.tile = 22;          // the bf16x6 kernel has the 128 x 128 tile only
  const int bm = pl.tile == 22 ? 128 : 64;
  const long long tiles = (long long)((Cout + bm - 1) / bm) * ((K + bm - 1) / bm);
  // Split count from the per-layer sweep of the cfg[1] shapes (tools/wgrad_sweep.py,
  // profiles/r2q_wgrad_sweep.txt): aim at ~2300 workgroups (9 per CU), but keep >= 512 reduction
  // rows per slice on the big maps (>= 256 on maps below 16k pixels, where there are few tiles
  // to begin with), never more than 128 slices (partial-sum traffic), and at most 2 when the
  // output alone already gives >= 256 tiles (fc1).
  long long splits = (2304 + tiles - 1) / tiles;
  // (M <= 2048 — the FC heads at 1024 RoIs: fc_cls of the shipped selectp = 1 step, 80 tiles: 8 slices of 128 rows
  //  52.6 us, 4 of 256 68.4, 3 of 352 (the round-down below) 69.5; tools/fc_cls_wgrad_ab.py, profiles/r9a)
  long long min_rows = M >= 16384 ? 512 : (M <= 2048 ? 128 : 256);
  if (const char* e = getenv("BGS_WGRAD_MINROWS")) {            // A/B
    const int f = atoi(e);
    if (f >= 16) min_rows = f;
  }
  const long long max_splits = (M + min_rows - 1) / min_rows;
  if (splits > max_splits) splits = max_splits;
  if (splits > 128) splits = 128;
  if (tiles >= 256 && splits > 2) splits = 2;
  // few rounds of workgroups: a total just above a multiple of the 256 CUs costs a whole extra
  // round (l2.c1: 264 workgroups 0.125 ms, 256 workgroups 0.104 ms) -> round down
  if (tiles * splits > 256 && tiles * splits < 2048) {
    const long long rounded = (tiles * splits / 256) * 256 / tiles;
    if (rounded >= 1) splits = rounded;
  }
  if (splits < 1) splits = 1;
  if (const char* e = getenv("BGS_WGRAD_SPLITS")) {
    const int f = atoi(e);
    if (f >= 1 && f <= 4096) splits = f;
  }
  int mps = (int)((M + splits - 1) / splits);
  mps = (mps + kBKM - 1) / kBKM * kBKM;
  pl.m_per_split = mps;
  pl.splits = (M + mps - 1) / mps;
  return pl;
}

}  // namespace

extern "C" size_t bgs_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R,
                                                   int S, int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0)
    return 0;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long long M = (long long)N * Ho * Wo;
  const int K = R * S * Cin;
  const WgradPlan pl = plan_wgrad((int)M, Cout, K);
  // weight partials + bias partials
  return ((size_t)pl.splits * Cout * K + (size_t)pl.splits * Cout) * sizeof(float) + 256;
}

extern "C" int bgs_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, float* db,
                                         int N, int H, int W, int Cin, int Cout, int R, int S,
                                         int stride, int pad, int accumulate, void* workspace,
                                         bgs_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!x || !dy || !dw || !workspace) return BGS_ERR_INVALID_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) % 16 != 0)
    return BGS_ERR_INVALID_ARG;
  WgradArgs p;
  p.gx = p.gy = p.gz = p.chunk = 0;
  p.x = x; p.dy = dy;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return BGS_ERR_INVALID_ARG;
  const long long M = (long long)N * p.Ho * p.Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  p.M = (int)M;
  p.K = R * S * Cin;
  const WgradPlan pl = plan_wgrad(p.M, Cout, p.K);
  p.m_per_split = pl.m_per_split;
  hipStream_t st = (hipStream_t)stream;
  float* ws = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const bool direct = pl.splits == 1 && !accumulate;
  p.out = direct ? dw : ws;
  const size_t n = (size_t)Cout * p.K;
  p.db_part = db ? ws + (size_t)pl.splits * n : nullptr;
  const int bm = pl.tile == 22 ? 128 : 64;
  dim3 grid((unsigned)((Cout + bm - 1) / bm), (unsigned)((p.K + bm - 1) / bm), (unsigned)pl.splits);
  if (pl.tile == 22)
    hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 2>), grid, dim3(kThreads), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_f32_kernel<1, 1>), grid, dim3(kThreads), 0, st, p);
  if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
  return launch_wgrad_reduce(ws, dw, n, direct, p.db_part, db, Cout, pl.splits, accumulate, st);
}

// bf16x6 / bf16 flavour (conv_wgrad_bfx_kernel): same contract and workspace as
// bgs_conv2d_wgrad_nhwc_f32; planes = 3 (fp32-faithful) or 1 (operands rounded to bf16).  Layers
// with Cout < 96 or K < 96 (the 128 x 128 tile would be mostly padding: stem, RPN heads) are routed
// to the fp32-MFMA kernel, which is exact.
int g_wgrad_bfx_enabled = -1;
int g_wgrad_xcd = -1;      // BGS_WGRAD_XCD=0: the plain 3-D grid (A/B)
static bool wgrad_xcd_order() {
  if (g_wgrad_xcd < 0) {
    const char* e = getenv("BGS_WGRAD_XCD");
    g_wgrad_xcd = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_wgrad_xcd != 0;
}
int g_wgrad_nbuf = -1;     // BGS_WGRAD_NBUF = 1 | 2 (LDS stage buffers of conv_wgrad_bfx_kernel<3>)
static int wgrad_nbuf() {
  if (g_wgrad_nbuf < 0) {
    const char* e = getenv("BGS_WGRAD_NBUF");
    g_wgrad_nbuf = (e && atoi(e) == 2) ? 2 : 1;      // default 1: 8.91 -> 8.16 ms over a selectp=0 backward
  }
  return g_wgrad_nbuf;
}

extern "C" size_t bgs_conv2d_wgrad_bfx_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R,
                                                       int S, int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0)
    return 0;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long long M = (long long)N * Ho * Wo;
  const int K = R * S * Cin;
  const WgradPlan a = plan_wgrad((int)M, Cout, K, true), b = plan_wgrad((int)M, Cout, K, false);
  const int splits = a.splits > b.splits ? a.splits : b.splits;
  return ((size_t)splits * Cout * K + (size_t)splits * Cout) * sizeof(float) + 256;
}

extern "C" int bgs_conv2d_wgrad_nhwc_f32_bfx(const float* x, const float* dy, float* dw, float* db,
                                             int N, int H, int W, int Cin, int Cout, int R, int S,
                                             int stride, int pad, int accumulate, int planes,
                                             void* workspace, bgs_stream_t stream) {
  if (planes != 1 && planes != 3) return BGS_ERR_INVALID_ARG;
  if (g_wgrad_bfx_enabled < 0) {
    const char* e = getenv("BGS_WGRAD_BFX");
    g_wgrad_bfx_enabled = (e && atoi(e) == 0) ? 0 : 1;
  }
  // (M <= 1536 rows — the FC heads at 1024 RoIs: the reduction is too short for the 128 x 128 tile's
  //  staging to amortise; fp32 kernel 0.755 vs 0.96 ms over fc1 / fc2 / fc_cls / fc_reg, profiles/r5n_*)
  const long long m_rows = (long long)N * ((H + 2 * pad - R) / (stride > 0 ? stride : 1) + 1) *
                           ((W + 2 * pad - S) / (stride > 0 ? stride : 1) + 1);
  if (!g_wgrad_bfx_enabled || Cout < 96 || R * S * Cin < 96 || (m_rows <= 1536 && g_wgrad_bfx_enabled != 2))
    return bgs_conv2d_wgrad_nhwc_f32(x, dy, dw, db, N, H, W, Cin, Cout, R, S, stride, pad, accumulate,
                                     workspace, stream);
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
      pad < 0)
    return BGS_ERR_INVALID_ARG;
  if (!x || !dy || !dw || !workspace) return BGS_ERR_INVALID_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0) return BGS_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) % 16 != 0)
    return BGS_ERR_INVALID_ARG;
  WgradArgs p;
  p.x = x; p.dy = dy;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return BGS_ERR_INVALID_ARG;
  const long long M = (long long)N * p.Ho * p.Wo;
  if (M > 0x7fffffffLL) return BGS_ERR_UNSUPPORTED;
  p.M = (int)M;
  p.K = R * S * Cin;
  const WgradPlan pl = plan_wgrad(p.M, Cout, p.K, true);
  p.m_per_split = pl.m_per_split;
  hipStream_t st = (hipStream_t)stream;
  float* ws = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const bool direct = pl.splits == 1 && !accumulate;
  p.out = direct ? dw : ws;
  const size_t n = (size_t)Cout * p.K;
  p.db_part = db ? ws + (size_t)pl.splits * n : nullptr;
  dim3 grid((unsigned)((Cout + 127) / 128), (unsigned)((p.K + 127) / 128), (unsigned)pl.splits);
  p.gx = (int)grid.x; p.gy = (int)grid.y; p.gz = (int)grid.z;
  p.chunk = 0;
  if (wgrad_xcd_order()) {
    p.chunk = (p.gx * p.gy * p.gz + 7) / 8;
    grid = dim3((unsigned)(8 * p.chunk));
  }
  bgs_internal_census_bump(BGS_CENSUS_WGRAD_BFX);
#ifdef BGS_ABLATE
  int abl = 0;
  if (const char* e = getenv("BGS_WGRAD_ABLATE")) abl = atoi(e);     // read at every launch (tools/wgrad_ablate.py)
#define WG_ABL(A_) case A_: hipLaunchKernelGGL((conv_wgrad_bfx_kernel<3, A_>), grid, dim3(kThreads), 0, st, p); break;
  if (planes == 3 && abl) {
    switch (abl) { WG_ABL(1) WG_ABL(2) WG_ABL(4) WG_ABL(8) WG_ABL(16) WG_ABL(17) WG_ABL(12) WG_ABL(14) WG_ABL(31) WG_ABL(29)
      default: return BGS_ERR_UNSUPPORTED; }
  } else
#undef WG_ABL
#endif
  if (planes == 3 && wgrad_nbuf() == 1) hipLaunchKernelGGL((conv_wgrad_bfx_kernel<3, 0, 1>), grid, dim3(kThreads), 0, st, p);
  else if (planes == 3) hipLaunchKernelGGL((conv_wgrad_bfx_kernel<3>), grid, dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL((conv_wgrad_bfx_kernel<1>), grid, dim3(kThreads), 0, st, p);
  if (hipGetLastError() != hipSuccess) return BGS_ERR_LAUNCH;
  return launch_wgrad_reduce(ws, dw, n, direct, p.db_part, db, Cout, pl.splits, accumulate, st);
}

// 0 = fp32-MFMA kernel everywhere, 1 = default routing, 2 = the bf16x6 kernel also on short reductions (tests)
extern "C" void bgs_conv2d_wgrad_bfx_enable(int on) { g_wgrad_bfx_enabled = on < 0 ? 0 : (on > 2 ? 2 : on); }
