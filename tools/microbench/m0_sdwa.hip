// Diagnostics: (1) does LDS-DMA (buffer_load_dwordx4 ... lds) accept an M0 base above 64 KiB?  (2) issue cost of SDWA ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) dma_test(const uint32_t *src, uint32_t *out, uint32_t lds_off)
{
  extern __shared__ u32x4 smem[];
  uint8_t *base = (uint8_t *)smem;
  u32x4 rs;
  const uint64_t addr = (uint64_t)(uintptr_t)src;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xFFFF);
  rs.z = 4096;
  rs.w = 0x00020000;
  const uint32_t voff = threadIdx.x * 16;
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)base + lds_off);
  for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4; i += 64)
    ((uint32_t *)base)[i] = 0xdeadbeef;
  __syncthreads();
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(dst), "s"(rs) : "memory");
  __syncthreads();
  // report where the first 4 source dwords landed
  uint32_t found = 0xffffffff;
  for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4; i += 64)
    if (((uint32_t *)base)[i] == src[0] && ((uint32_t *)base)[i + 1] == src[1])
      found = i * 4;
  for (int d = 32; d >= 1; d >>= 1)
  {
    const uint32_t o = __shfl_xor(found, d, 64);
    found = found < o ? found : o;
  }
  if (threadIdx.x == 0)
    out[0] = found;
}

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint32_t seed)
{
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
      if (OP == 0) { asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(c) : "v"(d)); }
      if (OP == 1) { asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a) : "v"(b)); asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(c) : "v"(d)); }
      if (OP == 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (OP == 3) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a)); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(c)); }
      if (OP == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d)); }
      if (OP == 5) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (OP == 6) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (OP == 7) { asm volatile("v_or_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_or_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d)); }
      if (OP == 8) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(c) : "v"(d)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int OP>
void run(const char *name, uint32_t *d)
{
  const int iters = 2000, waves_per_simd = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * 32 * waves_per_simd);
  printf("%-28s 8 waves/SIMD: %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, ns, ns * 2.4);
}

int main()
{
  uint32_t *src, *out, h[1024];
  for (int i = 0; i < 1024; i++)
    h[i] = 0x1000000u + i;
  hipMalloc(&src, 4096);
  hipMalloc(&out, 512 * 1024 * 4);
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)dma_test, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (uint32_t off : {0u, 4096u, 65536u - 1024u, 65536u, 65536u + 4096u, 131072u, 160u * 1024u - 1024u})
  {
    hipLaunchKernelGGL(dma_test, dim3(1), dim3(64), 160 * 1024, 0, src, out, off);
    uint32_t r = 0;
    hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA with M0 = %6u: data landed at LDS byte %d %s\n", off, (int)r, r == off ? "(ok)" : "(MISMATCH)");
  }
  run<0>("v_mov_b32_sdwa byte insert", out);
  run<1>("v_or_b32_sdwa src byte", out);
  run<2>("v_add_u32", out);
  run<3>("v_lshlrev_b32", out);
  run<4>("v_cndmask_b32 (vcc)", out);
  run<5>("v_mul_u32_u24", out);
  run<6>("v_or_b32", out);
  run<7>("v_or_b32_dpp", out);
  run<8>("mix mad+and", out);
  return 0;
}
