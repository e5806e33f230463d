// Self-test / calibration: sustained v_mfma_f32_32x32x2_f32 issue rate of this chip under its
// power limit, with operands in registers (no memory traffic).  bench.py --mfma-peak reports it
// next to the 157.3 TFLOP/s data-sheet figure so that the conv kernel's roofline fraction can be
// read against what the silicon sustains (MI355X_MICROARCH.md, "DVFS give-back").
#include "bgs_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* __restrict__ out) {
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = 1.0f + (float)(threadIdx.x & 7) * 0.125f, y = 0.5f + (float)(threadIdx.x & 3) * 0.25f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    x += 1e-9f;      // keep the loop from being collapsed
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 123.456f) out[0] = s;    // never true: keeps the accumulators alive
}

// The same for v_mfma_f32_32x32x16_bf16, the instruction of the bf16x6 / bf16 conv kernels: six operand registers per
// side (the hi / mid / lo planes of a split fragment), 24 MFMAs per loop trip on four accumulators in the product
// order of conv_bfx.hip.  `random` != 0 fills the operand registers with pseudo-random bf16 bit patterns of mixed
// sign and magnitude (what a split activation / filter fragment looks like); 0 = all-zero operands.  The chip clocks
// to its power budget: the two differ (MI355X_MICROARCH.md "DVFS give-back": +15-21 % TF on zero-filled operands),
// and the RANDOM figure / 6 is what a memory-free bf16x6 loop can sustain on this chip.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_peak_bf16_kernel(int iters, int random, float* __restrict__ out) {
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 fa[3][2], fb[3][2];
  unsigned h = 0x9E3779B9u * (threadIdx.x + 1u) + 0x85EBCA6Bu * (blockIdx.x + 1u);
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4 ua, ub;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        h = h * 1664525u + 1013904223u;
        // two bf16 per word: random sign / mantissa, exponent around 2^-1 .. 2^1 scaled down by 2^-8 per plane
        const unsigned e = (126u - 8u * s) << 7;
        ua[w] = random ? (((h & 0x807fu) | e) | ((((h >> 16) & 0x807fu) | e) << 16)) : 0u;
        h = h * 1664525u + 1013904223u;
        ub[w] = random ? (((h & 0x807fu) | e) | ((((h >> 16) & 0x807fu) | e) << 16)) : 0u;
      }
      fa[s][t] = __builtin_bit_cast(bf16x8, ua);
      fb[s][t] = __builtin_bit_cast(bf16x8, ub);
    }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 2; t >= 0; --t)
#pragma unroll
      for (int k = 0; k <= t; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[2 * a + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k][a], fb[t - k][b], acc[2 * a + b], 0, 0, 0);
    asm volatile("" : "+v"(fa[0][0]));      // keep the loop from being collapsed
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 123.456f) out[0] = s;
}

}  // namespace

// Launches `blocks` workgroups of 4 waves, each wave issuing 24 * iters v_mfma_f32_32x32x16_bf16 (32768 flop each).
extern "C" int bgs_selftest_mfma_peak_bf16(int blocks, int iters, int random_operands, float* out,
                                           bgs_stream_t stream) {
  if (blocks <= 0 || iters <= 0 || !out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mfma_peak_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters,
                     random_operands, out);
  BGS_RETURN_LAUNCH_STATUS();
}

// Launches `blocks` workgroups of 4 waves, each wave issuing 4 * iters MFMAs (4096 flop each).
extern "C" int bgs_selftest_mfma_peak(int blocks, int iters, float* out, bgs_stream_t stream) {
  if (blocks <= 0 || iters <= 0 || !out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
  BGS_RETURN_LAUNCH_STATUS();
}
