// Diagnostic microbenchmark #3: issue cost of the compare / count / multiply forms the decode loop could use (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint32_t seed, uint32_t sv)
{
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11, kc = 0x8000u + (seed & 1);
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (OP == 0) { asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(kc), "v"(a) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(kc), "v"(c) : "vcc"); }
      if (OP == 1) { asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" :: "s"(sv), "v"(a) : "s20", "s21"); asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" :: "s"(sv), "v"(c) : "s20", "s21"); }
      if (OP == 2) { asm volatile("v_cmp_gt_u32 vcc, 0x8000, %0" :: "v"(a) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, 0x8000, %0" :: "v"(c) : "vcc"); }
      if (OP == 3) { asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a) : "v"(b), "v"(d) : "vcc"); asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(c) : "v"(d), "v"(b) : "vcc"); }
      if (OP == 4) { asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 5) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c) : "v"(d), "v"(b)); }
      if (OP == 6) { asm volatile("v_mbcnt_lo_u32_b32 %0, vcc_lo, 0" : "=v"(a)); asm volatile("v_mbcnt_hi_u32_b32 %0, vcc_hi, %0" : "+v"(a)); }
      if (OP == 7) { asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(d)); }
      if (OP == 8) { asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 9) { asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 10) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_and_b32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 11) { asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(c) : "v"(d), "v"(b)); }
      if (OP == 12) { asm volatile("v_subrev_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_sub_u32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 13) { asm volatile("v_cmp_gt_i32 vcc, 0, %0" :: "v"(a) : "vcc"); asm volatile("v_cmp_gt_i32 vcc, 0, %0" :: "v"(c) : "vcc"); }
      if (OP == 14) { asm volatile("v_bfe_u32 %0, %1, 0, 11" : "=v"(a) : "v"(b)); asm volatile("v_bfe_u32 %0, %1, 0, 11" : "=v"(c) : "v"(d)); }
      if (OP == 15) { asm volatile("v_alignbit_b32 %0, %1, %2, 11" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_alignbit_b32 %0, %1, %2, 11" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 16) { asm volatile("v_lshrrev_b32 %0, 11, %1" : "=v"(a) : "v"(b)); asm volatile("v_lshrrev_b32 %0, 11, %1" : "=v"(c) : "v"(d)); }
      if (OP == 17) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(d)); }
      if (OP == 18) { asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b)); }
      if (OP == 19) { asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d)); asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b)); }
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
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, 10, 1u, 0x8000u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, iters, 1u, 0x8000u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * per_iter * 8);
  printf("%-44s %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, ns, ns * 2.4);
}

int main()
{
  uint32_t *out;
  hipMalloc(&out, 512 * 1024 * 4);
  run<0>("v_cmp_gt_u32 vcc, vgpr, vgpr (e32)", out);
  run<1>("v_cmp_gt_u32 sgpr_pair, sgpr, vgpr (e64)", out);
  run<2>("v_cmp_gt_u32 vcc, literal, vgpr (e32)", out);
  run<13>("v_cmp_gt_i32 vcc, 0, vgpr (e32 inline const)", out);
  run<3>("v_add_co_u32 d, vcc, vgpr, vgpr", out);
  run<4>("v_mul_u32_u24 (VOP2)", out);
  run<5>("v_mad_u32_u24 (VOP3)", out);
  run<19>("v_mul_hi_u32", out);
  run<6>("v_mbcnt_lo + v_mbcnt_hi (per instruction)", out);
  run<7>("v_lshlrev_b32 + v_add_u32 (per instruction)", out);
  run<8>("v_lshl_add_u32 vgprs", out);
  run<9>("v_lshl_or_b32 vgprs", out);
  run<10>("v_and_b32 vgprs", out);
  run<12>("v_sub_u32 vgprs", out);
  run<14>("v_bfe_u32", out);
  run<15>("v_alignbit_b32", out);
  run<16>("v_lshrrev_b32", out);
  run<11>("v_perm_b32", out);
  run<17>("v_mov_b32_dpp quad_perm", out);
  run<18>("v_pk_add_u16", out);
  return 0;
}
