// Microbenchmark (diagnostic, not part of the product): issue cost of the integer VALU ops the decode loop uses,
// at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint32_t seed)
{
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (OP == 0) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (OP == 1) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(c) : "v"(d)); }
      if (OP == 2) { asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(c) : "v"(d)); }
      if (OP == 3) { asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(c) : "v"(d)); }
      if (OP == 4) { asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a) : "v"(b)); asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(c) : "v"(d)); }
      if (OP == 5) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c) : "v"(d)); }
      if (OP == 6) { asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(a)); asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(c)); }
      if (OP == 7) { asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(c), "v"(d) : "vcc"); }
      if (OP == 8) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d)); }
      if (OP == 9) { asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(c) : "v"(d)); }
      if (OP == 10) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(c) : "v"(d)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd)
{
  const int iters = 2000;
  const int threads = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd); // one block per CU: waves spread over the 4 SIMDs
  const int blocks_per_cu = waves_per_simd > 4 ? waves_per_simd / 4 : 1;
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, d, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 32 * waves_per_simd; // wave-instructions issued on one SIMD
  const double ns_per_inst = ms * 1e6 / insts_per_simd;
  printf("%-16s waves/SIMD %d: %.3f ms  %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
}

int main()
{
  uint32_t *d;
  hipMalloc(&d, 256 * 8 * 1024 * 4);
  for (int w : {1, 2, 4, 8})
  {
    run<0>("v_and_b32", d, w);
    run<1>("v_mad_u32_u24", d, w);
    run<2>("v_perm_b32", d, w);
    run<3>("v_mbcnt_lo", d, w);
    run<4>("v_lshl_or_b32", d, w);
    run<5>("v_fma_f32", d, w);
    run<6>("v_lshrrev_b32", d, w);
    run<7>("v_cmp_gt_u32", d, w);
    run<8>("v_mov_b32_dpp", d, w);
    run<9>("v_and_or_b32", d, w);
    run<10>("v_lshl_add_u32", d, w);
  }
  return 0;
}
