// Library-level entry points of libbgs.so (version, error text, device self-test).
#include "bgs_common.h"

extern "C" int bgs_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* bgs_error_string(int code) {
  switch (code) {
    case BGS_OK: return "ok";
    case BGS_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or misaligned buffer)";
    case BGS_ERR_UNSUPPORTED: return "unsupported shape for this build";
    case BGS_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    default: return "unknown error code";
  }
}

namespace {
int g_census[BGS_CENSUS_FAMILIES];
}  // namespace

extern "C" void bgs_internal_census_bump(int family) {
  if (family >= 0 && family < BGS_CENSUS_FAMILIES) ++g_census[family];
}

extern "C" int bgs_launch_census(int family, int reset) {
  const int v = (family >= 0 && family < BGS_CENSUS_FAMILIES) ? g_census[family] : -1;
  if (reset)
    for (int i = 0; i < BGS_CENSUS_FAMILIES; ++i) g_census[i] = 0;
  return v;
}

namespace {
// out[0..1] = DPP wave max / sum, out[2..3] = ds_bpermute butterflies, for lane values in[lane].
__global__ void selftest_wave_reduce_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const float v = in[threadIdx.x & 63];
  const float a = bgs::wave_max(v), b = bgs::wave_sum(v);
  const float c = bgs::wave_max_shfl(v), d = bgs::wave_sum_shfl(v);
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = b;
    out[2] = c;
    out[3] = d;
  }
}
}  // namespace

// Device self-test of the wave-reduction primitive every kernel relies on.
//   in [64] float, out [4] float: {wave_max, wave_sum} by the build's primitive, then by the
//   ds_bpermute butterfly.  The GPU tests require out[0]==out[2] and out[1]~=out[3].
extern "C" int bgs_selftest_wave_reduce(const float* in, float* out, bgs_stream_t stream) {
  if (!in || !out) return BGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(selftest_wave_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in,
                     out);
  BGS_RETURN_LAUNCH_STATUS();
}
