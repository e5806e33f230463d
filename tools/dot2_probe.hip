// probe: v_dot2c_f32_bf16 with an inline-constant packed operand against the same operand held in a register
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* o) {
  const int i = threadIdx.x;
  const f32x2 v = {x[2 * i], x[2 * i + 1]};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const unsigned pk = __builtin_bit_cast(unsigned, h);
  unsigned c0 = 0x0000bf80u, c1 = 0xbf800000u;
  asm volatile("" : "+v"(c0), "+v"(c1));   // opaque: stays in registers
  o[8 * i + 0] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, 0x0000bf80u), v[0], false);
  o[8 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, 0xbf800000u), v[1], false);
  o[8 * i + 2] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, c0), v[0], false);
  o[8 * i + 3] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, c1), v[1], false);
  o[8 * i + 4] = v[0] - __builtin_bit_cast(float, pk << 16);
  o[8 * i + 5] = v[1] - __builtin_bit_cast(float, pk & 0xffff0000u);
  o[8 * i + 6] = __builtin_bit_cast(float, pk << 16);
  o[8 * i + 7] = __builtin_bit_cast(float, pk & 0xffff0000u);
}
int main() {
  float hx[128], ho[512], *dx, *dox;
  for (int i = 0; i < 128; ++i) hx[i] = (i % 2 ? -1.f : 1.f) * (1.2345678f + 0.37f * i) * (i % 7 == 0 ? 1e-3f : 1.f);
  hx[2] = 1e-37f; hx[3] = 3e-38f; hx[4] = 1e-40f; hx[5] = 2e-39f;
  hipMalloc(&dx, sizeof hx); hipMalloc(&dox, sizeof ho);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dox);
  hipMemcpy(ho, dox, sizeof ho, hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i) {
    for (int j = 0; j < 4; ++j) {
      unsigned a, b; memcpy(&a, &ho[8 * i + j], 4); memcpy(&b, &ho[8 * i + 4 + (j & 1)], 4);
      if (a != b) ++bad[j];
    }
    if (i < 4) printf("x %.9g %.9g | inline %.9g %.9g | reg %.9g %.9g | sub %.9g %.9g | bf16 %.9g %.9g\n", hx[2 * i], hx[2 * i + 1],
                      ho[8 * i], ho[8 * i + 1], ho[8 * i + 2], ho[8 * i + 3], ho[8 * i + 4], ho[8 * i + 5], ho[8 * i + 6], ho[8 * i + 7]);
  }
  printf("mismatches vs subtract: inline lo %d, inline hi %d, reg lo %d, reg hi %d (of 64)\n", bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
