// How many 256-thread workgroups with B bytes of static LDS fit on one CU (hipOccupancyMaxActiveBlocksPerMultiprocessor)?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occ tools/lds_occupancy.hip && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int B>
__global__ __launch_bounds__(256) void k(float* o) {
  __shared__ unsigned char lds[B];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  o[threadIdx.x] = lds[(threadIdx.x * 7) % B];
}
template <int B>
void probe() {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<B>, 256, 0);
  printf("%6d B -> %d workgroups / CU (%d B total)\n", B, n, B * n);
}
int main() {
  probe<51712>(); probe<53248>(); probe<53760>(); probe<54144>(); probe<54272>(); probe<54528>(); probe<54613>(); probe<55296>();
  probe<40960>(); probe<41984>(); probe<32768>(); probe<30720>();
  return 0;
}
