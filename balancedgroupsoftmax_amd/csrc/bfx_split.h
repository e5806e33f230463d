// The residual step of the exact three-way bf16 split (x = hi + mid + lo, every plane a bf16): r = x - bf16(x).
//
// `pk` holds two RNE-rounded bf16 values (v_cvt_pk_bf16_f32 of two neighbouring k).  Up to round 4 the residual
// was formed as  x - float(half of pk): one shift or mask to widen the half, then (packed) subtraction — 9 VALU
// instructions per two elements for the whole split.  gfx950 has v_dot2c_f32_bf16 (D += A.lo * B.lo + A.hi * B.hi on
// packed bf16 operands, fp32 accumulator): with B = (-1, 0) resp. (0, -1) and D = x it yields the same residual
// straight from the PACKED word — 7 instructions per two elements.  The result is the same number: the one non-zero
// product is exact, and x - bf16(x) is representable in fp32 (Sterbenz-style: the residual has at most 16
// significant bits), so no rounding happens in either formulation.  Differences are confined to non-finite input
// (0 * inf of the neighbouring element); fp32-subnormal residuals are kept, not flushed
// (tests/test_gpu_det_ops.py::test_bfx_split_planes_match_the_numpy_restatement pins the planes).  -DBGS_SPLIT_SUB builds the subtract form (python -m ...csrc.build --variant splitsub: the A/B arm).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bgs_split_bf16x2 __attribute__((ext_vector_type(2)));

// x - float(low bf16 half of pk)
__device__ __forceinline__ float bfx_resid_lo(unsigned pk, float x) {
#ifdef BGS_SPLIT_SUB
  return x - __builtin_bit_cast(float, pk << 16);
#else
  // The multiplier (-1, 0) is kept in an SGPR behind an empty asm: as a compile-time constant hipcc 7.2 encodes
  // the packed pair {bf16 -1.0, 0} as the INLINE constant -1.0, which the hardware reads as the 32-bit pattern
  // 0xbf800000 = (0, -1) — the other element (tools/dot2_probe.hip: 64 of 64 residuals wrong that way).
  unsigned neg_lo = 0x0000bf80u;
  asm("" : "+s"(neg_lo));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bgs_split_bf16x2, pk),
                                         __builtin_bit_cast(bgs_split_bf16x2, neg_lo), x, false);
#endif
}

// x - float(high bf16 half of pk)
__device__ __forceinline__ float bfx_resid_hi(unsigned pk, float x) {
#ifdef BGS_SPLIT_SUB
  return x - __builtin_bit_cast(float, pk & 0xffff0000u);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bgs_split_bf16x2, pk),
                                         __builtin_bit_cast(bgs_split_bf16x2, 0xbf800000u), x, false);
#endif
}
