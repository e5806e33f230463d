// Device-side helpers shared by the libbgs kernels (gfx950 / CDNA4, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgs.h"
#include "bgs_tuning.h"

#define BGS_WAVE 64

// Every launch is checked the way the reference checks its own
// (THCudaCheck(cudaGetLastError()), roi_align_kernel.cu:145) — but reported through
// the return code instead of aborting.
#define BGS_RETURN_LAUNCH_STATUS()                         \
  do {                                                     \
    return hipGetLastError() == hipSuccess ? BGS_OK : BGS_ERR_LAUNCH; \
  } while (0)

// launch census (capi.hip): bgs_census_bump(BGS_CENSUS_*) at the launch site of a kernel family
extern "C" void bgs_internal_census_bump(int family);

namespace bgs {

// ---- wave64 all-reduce -------------------------------------------------------------
// Default: DPP row operations (quad_perm / row_ror / row_bcast, the gfx9 recipe also used by
// rocPRIM's warp_reduce_dpp) — 6 dependent VALU ops, then v_readlane of lane 63; the result
// is wave-uniform (SGPR).  -DBGS_NO_DPP builds the ds_bpermute butterfly instead (A/B + safety).
#ifndef BGS_NO_DPP
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define BGS_DPP_REDUCE(v, OP)                                   \
  do {                                                          \
    v = OP(v, dpp_mov<0xb1>(v));  /* quad_perm:[1,0,3,2] */     \
    v = OP(v, dpp_mov<0x4e>(v));  /* quad_perm:[2,3,0,1] */     \
    v = OP(v, dpp_mov<0x124>(v)); /* row_ror:4 */               \
    v = OP(v, dpp_mov<0x128>(v)); /* row_ror:8 */               \
    v = OP(v, dpp_mov<0x142>(v)); /* row_bcast:15 */            \
    v = OP(v, dpp_mov<0x143>(v)); /* row_bcast:31 */            \
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); \
  } while (0)
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_max(float v) {
  BGS_DPP_REDUCE(v, fmaxf);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  BGS_DPP_REDUCE(v, op_add);
  return v;
}
#else
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, BGS_WAVE));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, BGS_WAVE);
  return v;
}
#endif

// ds_bpermute butterflies (always available; used by the self-test and for int/double).
__device__ __forceinline__ float wave_max_shfl(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, BGS_WAVE));
  return v;
}
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, BGS_WAVE);
  return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, BGS_WAVE);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, BGS_WAVE);
  return v;
}

// Wave-uniform value -> SGPR (lets the compiler use scalar loads / uniform branches).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Native clang vectors: a 16-byte access is ONE global_load/store_dwordx4 or ds_read/write_b128
// (HIP's float4 struct assigns member-wise and its stores were split into 4 dword stores).
template <int VEC>
struct VecT;
template <>
struct VecT<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <>
struct VecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct VecT<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&out)[VEC]) {
  using V = typename VecT<VEC>::type;
  const V v = *reinterpret_cast<const V*>(p);
  if constexpr (VEC == 1) {
    out[0] = v;
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = v[j];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&in)[VEC]) {
  using V = typename VecT<VEC>::type;
  V v;
  if constexpr (VEC == 1) {
    v = in[0];
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = in[j];
  }
  *reinterpret_cast<V*>(p) = v;
}

// The same with the non-temporal hint (global_load / global_store ... nt): data streamed through once
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* p, float (&out)[VEC]) {
  using V = typename VecT<VEC>::type;
  const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
  if constexpr (VEC == 1) {
    out[0] = v;
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = v[j];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec_nt(float* p, const float (&in)[VEC]) {
  using V = typename VecT<VEC>::type;
  V v;
  if constexpr (VEC == 1) {
    v = in[0];
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = in[j];
  }
  __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
}

// splitmix64 finaliser: counter-based RNG keyed by (seed, stream, index).
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint32_t stream, uint32_t idx) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)idx + 1ull) +
               0xD1B54A32D192ED03ull * ((uint64_t)stream + 1ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

// Sampling keys of the GroupSoftmax "others" draw: one 64-bit hash per (draw, bin) gives a 32-bit
// salt, the per-row key is the lowbias32 mixer (a bijection of 32-bit words, two multiplies) of
// row ^ salt — an order of magnitude fewer VALU instructions than a 64-bit hash per row (the fused
// head kernel evaluates N keys per row and sampled bin), and distinct rows always have distinct
// keys (no ties inside a bin).
__device__ __forceinline__ uint32_t gs_bin_salt(uint64_t seed, uint32_t bin) {
  return hash_u32(seed, bin, 0x5bd1e995u);
}
__device__ __forceinline__ uint32_t gs_key(uint32_t salt, uint32_t row) {
  uint32_t x = row ^ salt;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// Exact-k sampling without replacement in O(1) per row: the candidates of a bin are numbered in
// row order (pos = number of candidate rows before this one, a prefix count), a keyed pseudo-random
// PERMUTATION of [0, M) is applied to the position, and the row is selected iff the image is < k —
// exactly k of the M candidates, every k-subset being the image of the first k positions under the
// permutation.  The permutation is a 6-round Feistel network on 2h bits (4^h >= M, round function =
// the lowbias32 mixer keyed by salt and round) made a bijection of [0, M) by cycle walking.  This
// replaces "rank of the row's key among all candidates" (a radix select in the prepare kernel, an
// O(N) hash scan per row and bin in the fused head kernel).
__device__ __forceinline__ uint32_t gs_perm(uint32_t salt, uint32_t x, uint32_t M) {
  uint32_t h = 1;
  while (h < 16 && (1u << (2 * h)) < M) ++h;
  const uint32_t mask = (1u << h) - 1u;
  for (int guard = 0; guard < 4096; ++guard) {        // cycle walking: expected < 4 rounds (2^(2h) < 4M)
    uint32_t L = x >> h, R = x & mask;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const uint32_t F = gs_key(salt + 0x9E3779B9u * (uint32_t)(i + 1), R) & mask;
      const uint32_t t = L ^ F;
      L = R;
      R = t;
    }
    x = (L << h) | R;
    if (x < M) break;
  }
  return x;
}

// Bitonic networks in LDS with one thread per compare-exchange pair (pair j: lo = 2 j - (j & (stride - 1)),
// hi = lo + stride): for stride <= 64 the 64 pairs of a wave lie in ONE contiguous 128-element block that no
// other wave touches in that stage, so the stages with stride <= 64 only need the wave's own LDS writes to be
// visible to its own lanes — a wave-scope fence instead of a workgroup barrier (51 of the 66 stages of a
// 2048-element sort).  A workgroup barrier is needed after a stage with stride >= 128, and after the last
// stage (stride 1) of a size whose successor starts with stride >= 128.
__device__ __forceinline__ void bitonic_stage_sync(int size, int stride) {
  const bool block = stride >= 128 || (stride == 1 && size >= 128);
  if (block) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// integer wave sum on the DPP path (wave_sum_i above is six dependent ds_bpermute round trips)
#ifndef BGS_NO_DPP
// (same recipe as BGS_DPP_REDUCE: only lane 63's value is the full sum)
__device__ __forceinline__ int wave_sum_i_fast(int v) {
  v += __builtin_amdgcn_update_dpp(v, v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(v, v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);  // row_ror:8
  v += __builtin_amdgcn_update_dpp(v, v, 0x142, 0xf, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(v, v, 0x143, 0xf, 0xf, false);  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}
#else
__device__ __forceinline__ int wave_sum_i_fast(int v) { return wave_sum_i(v); }
#endif

}  // namespace bgs
