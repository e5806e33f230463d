// Micro-benchmark (round 6): do LDS fragment reads overlap with MFMA execution on a CU, or do their times add?
// Every 3x3 halo / 1x1 kernel of this repo spends, per k step of a wave, 24 x v_mfma_f32_32x32x16_bf16 (768 matrix-pipe
// cycles) and 12 x ds_read_b128 (12 KB; 4 waves = 48 KB = 384 LDS cycles per workgroup step at 128 B/clk), and measures
// step times close to the SUM of the two (tools/halo_small_map_probe.py, tools/planes_ablate.py).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mlo tools/mfma_lds_overlap.hip && /tmp/mlo
//
// Workgroups of 256 threads, W per CU (1..3), every wave loops `iters` times over a step of
//   mode 0: 24 MFMAs only                      mode 1: 12 ds_read_b128 only (conflict-free, consumed by an empty asm)
//   mode 2: the 12 reads, then the 24 MFMAs on OTHER registers (reads of step i + 1 issued before the MFMAs of step i:
//           a software-pipelined wave), one lgkmcnt(0) per step
//   mode 3: as the kernels do it — reads, lgkmcnt(0), MFMAs ON the registers just read (dependent), no barrier
//   mode 4: mode 3 + one workgroup barrier per step
// and prints ns per step and the fraction of the MFMA-only rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(int iters, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[48 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 48 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;   // bf16 1.0 pairs
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int off = lane * 16;                        // 64 lanes x 16 B contiguous: conflict-free
  bf16x8 f[12], g[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    f[j] = *reinterpret_cast<const bf16x8*>(lds + off + j * 1024);
    g[j] = f[j];
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 12; ++j) g[j] = *reinterpret_cast<const bf16x8*>(lds + off + j * 1024 + (it & 3) * 12288);
    }
    if (MODE == 3 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < 12; ++j) f[j] = *reinterpret_cast<const bf16x8*>(lds + off + j * 1024 + (it & 3) * 12288);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE != 1) {
#pragma unroll
      for (int m = 0; m < 24; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m % 6], f[6 + (m % 6)], acc[m & 3], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 12; ++j) asm volatile("" ::"v"(g[j]));
    }
    if (MODE == 4) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  if (s == 123.456f) sink[0] = s;
}

// ---- the halo kernel's k step with its filter delivery: per step and workgroup 12 KB of filter (three planes x 128 rows x
//      32 B) from an L2-resident buffer (`wbytes`, cycled), 6 KB of A fragments + 6 KB of B fragments per wave from LDS,
//      24 MFMAs per wave, one barrier.  DELIV: 0 = no delivery (B fragments from a static buffer) | 1 = LDS-DMA
//      (global_load_lds_dwordx4, double buffer, vmcnt(0) before the barrier: conv3x3_halo_bfx4_kernel) | 2 = ring of three
//      slices, counted wait (the slice gets two steps) | 3 = no filter LDS: every wave loads ITS six B fragments from L2
//      into registers one step ahead (two waves load the same rows: 24 KB per workgroup step)
__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// SAME: every workgroup walks the slices in the same order from the same start (the real launch: all co-resident
// workgroups are at the same (chunk, tap) at about the same time and pull the SAME 12 KB from the L2) instead of from
// a start of its own
template <int DELIV, bool SAME = false>
__global__ __launch_bounds__(256, 3) void halo_step(int iters, const unsigned char* __restrict__ w, int wbytes, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[(DELIV == 2 ? 36 : 24) * 1024 + 18 * 1024];
  constexpr int A_OFF = (DELIV == 2 ? 36 : 24) * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (int)sizeof(lds) / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int off = lane * 16;
  const int wm = wave >> 1, wn = wave & 1;
  // slice s of this workgroup: a different 12 KB window per step, different start per workgroup
  const int nsl = wbytes / 12288;
  int sl = SAME ? (int)(blockIdx.x & 1) * 144 : (int)(blockIdx.x * 37u) % nsl;
  auto issue = [&](int slice, int buf) {                         // wave w carries rows 32 w .. of the three planes
#pragma unroll
    for (int p = 0; p < 3; ++p)
      glds16(w + (size_t)slice * 12288 + p * 4096 + wave * 1024 + lane * 16, lds + buf * 12288 + p * 4096 + wave * 1024);
  };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w), 0, wbytes, 0x00020000);
  bf16x8 fb[6], fbn[6];
  auto loadb = [&](int slice, bf16x8 (&d)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      d[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, slice * 12288 + (j >> 1) * 4096 + wn * 2048 + (j & 1) * 1024, 0));
  };
  if (DELIV == 1 || DELIV == 2) {
    issue(sl, 0);
    if (DELIV == 2) issue((sl + 1) % nsl, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (DELIV == 3) loadb(sl, fb);
  for (int it = 0; it < iters; ++it) {
    int rd = 0;
    if (DELIV == 1) {
      rd = it & 1;
      issue((sl + it + 1) % nsl, rd ^ 1);
    } else if (DELIV == 2) {
      rd = it % 3;
      issue((sl + it + 2) % nsl, (it + 2) % 3);
    } else if (DELIV == 3) {
      loadb((sl + it + 1) % nsl, fbn);
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 fa[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(lds + A_OFF + off + (wm * 6 + j) * 1024 + (it % 9 / 3) * 64);
    if (DELIV != 3) {
#pragma unroll
      for (int j = 0; j < 6; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lds + rd * 12288 + off + (j >> 1) * 4096 + wn * 2048 + (j & 1) * 1024);
    }
#pragma unroll
    for (int m = 0; m < 24; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m % 6], fb[(m * 5 + 1) % 6], acc[m & 3], 0, 0, 0);
    if (DELIV == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DELIV == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (DELIV == 3) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 6; ++j) fb[j] = fbn[j];
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  if (s == 123.456f) sink[0] = s;
}

template <int DELIV, bool SAME = false>
float run_halo(int wgs, int iters, const unsigned char* w, int wbytes) {
  float* sink;
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((halo_step<DELIV, SAME>), dim3(wgs), dim3(256), 0, 0, iters, w, wbytes, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((halo_step<DELIV, SAME>), dim3(wgs), dim3(256), 0, 0, iters, w, wbytes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(sink);
  return ms;
}

template <int MODE>
float run(int wgs, int iters) {
  float* sink;
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(sink);
  return ms;
}

int main() {
  const int iters = 4000;
  const char* names[5] = {"MFMAs only", "ds_reads only", "reads (independent, pipelined) + MFMAs",
                          "reads -> wait -> MFMAs (dependent)", "dependent + barrier per step"};
  for (int w = 1; w <= 3; ++w) {
    const int wgs = 256 * w;
    float t[5];
    t[0] = run<0>(wgs, iters);
    t[1] = run<1>(wgs, iters);
    t[2] = run<2>(wgs, iters);
    t[3] = run<3>(wgs, iters);
    t[4] = run<4>(wgs, iters);
    printf("%d workgroup(s) per CU (%d waves per SIMD):\n", w, w);
    for (int m = 0; m < 5; ++m)
      printf("  mode %d %-42s %8.1f ns per step  (x %.2f of MFMAs only; MFMA + reads = %.1f)\n", m, names[m],
             t[m] * 1e6f / iters, t[m] / t[0], (t[0] + t[1]) * 1e6f / iters);
  }
  // the halo step with its filter delivery; filter buffer 3.5 MB (the P2 layer's split filter: 288 slices of 12 KB)
  const int wbytes = 288 * 12288;
  unsigned char* w;
  hipMalloc(&w, wbytes);
  hipMemset(w, 0x3f, wbytes);
  const char* dn[4] = {"no filter delivery", "LDS-DMA, double buffer, vmcnt(0) (the kernel)", "LDS-DMA, ring of 3, counted wait",
                       "filter fragments L2 -> registers, one step ahead"};
  for (int wg = 1; wg <= 3; ++wg) {
    const int wgs = 256 * wg;
    float t[4];
    t[0] = run_halo<0>(wgs, iters, w, wbytes);
    t[1] = run_halo<1>(wgs, iters, w, wbytes);
    t[2] = run_halo<2>(wgs, iters, w, wbytes);
    t[3] = run_halo<3>(wgs, iters, w, wbytes);
    const float ts1 = run_halo<1, true>(wgs, iters, w, wbytes), ts3 = run_halo<3, true>(wgs, iters, w, wbytes);
    printf("halo step, %d workgroup(s) per CU:\n", wg);
    printf("  %-50s %8.1f ns per step = %6.1f ns per workgroup step per CU\n", "LDS-DMA, every workgroup on the SAME slice", ts1 * 1e6f / iters, ts1 * 1e6f / iters / wg);
    printf("  %-50s %8.1f ns per step = %6.1f ns per workgroup step per CU\n", "L2 -> registers, every workgroup on the SAME slice", ts3 * 1e6f / iters, ts3 * 1e6f / iters / wg);
    for (int m = 0; m < 4; ++m)
      printf("  %-50s %8.1f ns per step = %6.1f ns per workgroup step per CU\n", dn[m], t[m] * 1e6f / iters, t[m] * 1e6f / iters / wg);
  }
  hipFree(w);
  return 0;
}
