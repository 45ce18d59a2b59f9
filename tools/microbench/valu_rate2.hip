// Diagnostic microbenchmark #2: issue cost of shift / select / d16 forms at 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint32_t seed, uint32_t sv)
{
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (OP == 0) { asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a) : "v"(b)); asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(c) : "v"(d)); }
      if (OP == 1) { asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a) : "v"(b)); asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(c) : "v"(d)); }
      if (OP == 2) { asm volatile("v_lshrrev_b32 %0, 11, %1" : "=v"(a) : "v"(b)); asm volatile("v_lshrrev_b32 %0, 11, %1" : "=v"(c) : "v"(d)); }
      if (OP == 3) { asm volatile("v_lshrrev_b32 %0, %2, %1" : "=v"(a) : "v"(b), "s"(sv)); asm volatile("v_lshrrev_b32 %0, %2, %1" : "=v"(c) : "v"(d), "s"(sv)); }
      if (OP == 4) { asm volatile("v_and_b32 %0, %2, %1" : "=v"(a) : "v"(b), "s"(sv)); asm volatile("v_and_b32 %0, %2, %1" : "=v"(c) : "v"(d), "s"(sv)); }
      if (OP == 5) { asm volatile("v_add_u32 %0, %2, %1" : "=v"(a) : "v"(b), "s"(sv)); asm volatile("v_add_u32 %0, %2, %1" : "=v"(c) : "v"(d), "s"(sv)); }
      if (OP == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(c) : "v"(d)); }
      if (OP == 7) { asm volatile("v_bfe_u32 %0, %1, 8, 12" : "=v"(a) : "v"(b)); asm volatile("v_bfe_u32 %0, %1, 8, 12" : "=v"(c) : "v"(d)); }
      if (OP == 8) { asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a) : "v"(b), "s"(sv)); asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(c) : "v"(d), "s"(sv)); }
      if (OP == 9) { asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "s"(sv)); asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "s"(sv)); }
      if (OP == 10) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (OP == 11) { asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); }
      if (OP == 12) { asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_lshrrev_b32 %0, 11, %1" : "=v"(c) : "v"(d)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int OP>
void run(const char *name, uint32_t *d, int per_iter = 32)
{
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, 10, 1u, 11u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, iters, 1u, 11u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * per_iter * 8);
  printf("%-34s %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, ns, ns * 2.4);
}

int main()
{
  uint32_t *out;
  hipMalloc(&out, 512 * 1024 * 4);
  run<0>("v_lshlrev_b32 d, 3, s", out);
  run<1>("v_lshlrev_b32 d, 16, s", out);
  run<2>("v_lshrrev_b32 d, 11, s", out);
  run<3>("v_lshrrev_b32 d, sgpr, s", out);
  run<4>("v_and_b32 d, sgpr, s", out);
  run<5>("v_add_u32 d, sgpr, s", out);
  run<6>("v_cndmask_b32 (sgpr mask, e64)", out);
  run<7>("v_bfe_u32", out);
  run<8>("v_lshl_add_u32 d, s, 3, sgpr", out);
  run<9>("v_mov_b32 d, sgpr", out);
  run<10>("v_xor_b32", out);
  run<11>("v_cmp+v_cndmask pair (per pair)", out, 16);
  run<12>("mad + lshrrev mix", out);
  return 0;
}
