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

}  // namespace

// Launches `blocks` workgroups of 4 waves, each wave issuing 4 * iters MFMAs (4096 flop each).
extern "C" int bgs_selftest_mfma_peak(int blocks, int iters, float* out, bgs_stream_t stream) {
  if (blocks <= 0 || iters <= 0 || !out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
  BGS_RETURN_LAUNCH_STATUS();
}
