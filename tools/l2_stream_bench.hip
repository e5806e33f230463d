// Micro-benchmark: how many bytes per clock does a CU get out of L2 when all 256 CUs stream at once, as a
// function of the access pattern of one wave instruction?  (Round 3: every operand-streaming kernel of this
// repo — the bf16-storage 1x1 conv, the operand ring, the weight gradient — saturates at ~18 B/clk/CU.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2b tools/l2_stream_bench.hip && /tmp/l2b
//
// Every workgroup (256 threads, 3 per CU) loops over a private window of `win` bytes of a buffer that fits L2
// (window reuse = L2 hits after the first pass; windows of neighbouring workgroups overlap like the A tile of
// 8 column tiles when `share` > 1).  Patterns of one wave instruction (64 lanes x 16 B = 1 KB):
//   0  contiguous 1 KB                                   (the filter slices: 8 full 128-B lines)
//   1  16 rows x 64 B, row stride `pitch` bytes           (fp32 A rows of a K16 step / bf16 rows of K32: half lines)
//   2  8 rows x 128 B, row stride `pitch`                 (bf16 rows of a K64 step: full lines)
//   3  32 rows x 32 B                                     (quarter lines)
// Destination: VGPRs (global_load_dwordx4) or LDS (global_load_lds_dwordx4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool TO_LDS>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ buf, size_t buf_bytes,
                                                     int pattern, int pitch, int win, int iters, int share,
                                                     unsigned* __restrict__ sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[16 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // window of this workgroup (workgroups b, b+1, .., b+share-1 share one)
  const size_t wbase = ((size_t)(blockIdx.x / share) * (size_t)win) % (buf_bytes - (size_t)win);
  size_t lane_off;
  int rows_per_instr;
  if (pattern == 0) { lane_off = (size_t)lane * 16; rows_per_instr = 0; }
  else if (pattern == 1) { lane_off = (size_t)(lane >> 2) * pitch + (lane & 3) * 16; rows_per_instr = 16; }
  else if (pattern == 2) { lane_off = (size_t)(lane >> 3) * pitch + (lane & 7) * 16; rows_per_instr = 8; }
  else { lane_off = (size_t)(lane >> 1) * pitch + (lane & 1) * 16; rows_per_instr = 32; }
  u32x4 acc = {0u, 0u, 0u, 0u};
  // per iteration every wave issues 8 instructions = 8 KB; the window is walked in 32 KB steps per workgroup
  const int row_bytes = pattern == 1 ? 64 : (pattern == 2 ? 128 : 32);
  for (int it = 0; it < iters; ++it) {
    // relative offset of this wave's instruction u inside the window (everything wraps inside the window, so the
    // footprint of a workgroup is exactly `win` bytes)
    size_t rel0;
    if (pattern == 0) rel0 = (size_t)it * 32768 + (size_t)wave * 8192;
    else rel0 = ((size_t)it * row_bytes) % (size_t)pitch + (size_t)wave * 8 * rows_per_instr * pitch;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t rel = rel0 + lane_off + (pattern == 0 ? (size_t)u * 1024 : (size_t)u * rows_per_instr * pitch);
      const unsigned char* src = buf + wbase + (rel & (size_t)(win - 1));       // win is a power of two
      if (TO_LDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + wave * 4096 + (u & 3) * 1024),
                                         16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src);
        acc ^= v;
      }
    }
    if (TO_LDS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if (TO_LDS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc[0] = *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
  const size_t buf_bytes = 24u << 20;     // 24 MB: within the 32 MB of aggregate L2 / the 256 MB Infinity Cache
  unsigned char* buf;
  unsigned* sink;
  hipMalloc(&buf, buf_bytes + (1 << 20));
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, buf_bytes + (1 << 20));
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk_ghz = prop.clockRate * 1e-6;
  printf("# %s: %d CUs, %.2f GHz nominal; 3 workgroups of 256 threads per CU, each wave 8 x 1 KB per iteration\n",
         prop.name, cus, clk_ghz);
  printf("# pattern | dest | window KB | share | GB/s total | B/clk/CU (at nominal clock)\n");
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = cus * 3, iters = 2000;
  const char* names[4] = {"1 KB contiguous", "16 rows x 64 B", "8 rows x 128 B", "32 rows x 32 B"};
  for (int to_lds = 0; to_lds < 2; ++to_lds)
    for (int pattern = 0; pattern < 4; ++pattern)
      for (int share = 1; share <= 8; share *= 8)
        for (int win_kb = 64; win_kb <= 1024; win_kb *= 16) {
          const int pitch = 2048;           // row pitch of a [M][1024] bf16 (or [M][512] fp32) activation matrix
          const int win = win_kb * 1024;
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (to_lds) hipLaunchKernelGGL(stream_kernel<true>, dim3(grid), dim3(256), 0, 0, buf, buf_bytes, pattern, pitch, win, iters, share, sink);
            else hipLaunchKernelGGL(stream_kernel<false>, dim3(grid), dim3(256), 0, 0, buf, buf_bytes, pattern, pitch, win, iters, share, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
              const double bytes = (double)grid * 4 * 8 * 1024 * iters;
              const double gbs = bytes / (ms * 1e-3) * 1e-9;
              printf("%-16s | %-4s | %5d | %d | %9.0f | %6.1f\n", names[pattern], to_lds ? "LDS" : "VGPR", win_kb, share,
                     gbs, gbs / cus / clk_ghz);
            }
          }
        }
  return 0;
}
