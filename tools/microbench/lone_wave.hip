// Diagnostic microbenchmark: what ONE wavefront alone on its SIMD pays per instruction (the mt_ encoder's situation: one wavefront per
// block, one dependent chain).  Every case is a sequence of 16 copies inside a loop, run by one wavefront per CU on every fourth CU's
// worth of workgroups (grid 64, block 64), timed with s_memrealtime (100 MHz) around the loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X
#define GROUP "v_cmp_ge_u32 vcc, %0, %2\n\t" \
                         "s_bcnt1_i32_b64 s22, vcc\n\t" \
                         "v_mbcnt_lo_u32_b32 %5, vcc_lo, 0\n\t" \
                         "v_mbcnt_hi_u32_b32 %5, vcc_hi, %5\n\t" \
                         "s_sub_u32 %1, %1, s22\n\t" \
                         "v_add_lshl_u32 %5, %5, %1, 1\n\t" \
                         "v_and_or_b32 %5, %5, %9, %7\n\t" \
                         "v_cndmask_b32 %5, %8, %5, vcc\n\t" \
                         "ds_write_b16 %5, %0\n\t" \
                         "v_cndmask_b32_sdwa %0, %0, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
                         "v_mul_hi_u32 %6, %0, %3\n\t" \
                         "v_lshrrev_b32 %6, 3, %6\n\t" \
                         "v_mul_u32_u24 %6, %6, %4\n\t" \
                         "v_add3_u32 %0, %0, %4, %6\n\t"
#define GROUP2 "v_cmp_ge_u32 vcc, %0, %2\n\t" \
                         "s_bcnt1_i32_b64 s22, vcc\n\t" \
                         "v_mbcnt_lo_u32_b32 %5, vcc_lo, 0\n\t" \
                         "v_mbcnt_hi_u32_b32 %5, vcc_hi, %5\n\t" \
                         "s_sub_u32 %1, %1, s22\n\t" \
                         "v_add_lshl_u32 %5, %5, %1, 1\n\t" \
                         "v_and_or_b32 %5, %5, %9, %7\n\t" \
                         "v_cndmask_b32 %5, %8, %5, vcc\n\t" \
                         "ds_write_b16 %5, %0\n\t" \
                         "v_cndmask_b32_sdwa %0, %0, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
                         "v_mul_hi_u32 %6, %0, %3\n\t" \
                         "v_lshrrev_b32_sdwa %6, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
                         "v_mul_u32_u24 %6, %6, %4\n\t" \
                         "v_add3_u32 %0, %0, %4, %6\n\t"

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint64_t *ticks, int iters, uint32_t seed)
{
  __shared__ uint32_t lds[4096];
  uint32_t x = threadIdx.x * 2654435761u + seed, a = x ^ 0x9e3779b9u, b = 0x12345u + seed, c = 3, e = 7, t0 = 0, t1 = 0;
  uint32_t p = 4096 + seed;
  const uint32_t ring = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)lds;
  lds[threadIdx.x] = x;
  __syncthreads();
  const uint64_t s = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++)
  {
    if (OP == 0) // dependent VALU chain
      asm volatile(REP16("v_add_u32 %0, %0, %1\n\t") : "+v"(x) : "v"(a));
    if (OP == 1) // two independent VALU chains
      asm volatile(REP16("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\t") : "+v"(x), "+v"(b) : "v"(a));
    if (OP == 2) // dependent v_mul_hi_u32 chain
      asm volatile(REP16("v_mul_hi_u32 %0, %0, %1\n\t") : "+v"(x) : "v"(a));
    if (OP == 3) // VALU -> SGPR pair -> SALU -> SALU -> VALU
      asm volatile(REP16("v_cmp_ge_u32 s[20:21], %0, %1\n\ts_bcnt1_i32_b64 s22, s[20:21]\n\ts_sub_u32 %2, %2, s22\n\tv_add_u32 %0, %0, %2\n\t") : "+v"(x), "+v"(a), "+s"(p) : : "s20", "s21", "s22", "scc");
    if (OP == 4) // independent SALU
      asm volatile(REP16("s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\t") : : : "s20", "s21", "scc");
    if (OP == 5) // dependent SALU
      asm volatile(REP16("s_add_u32 s20, s20, 1\n\ts_add_u32 s20, s20, 1\n\t") : : : "s20", "scc");
    if (OP == 6) // masked LDS write between two writes of EXEC
      asm volatile(REP16("v_cmp_ge_u32 s[20:21], %0, %1\n\ts_mov_b64 s[22:23], exec\n\ts_and_b64 exec, s[20:21], exec\n\tds_write_b16 %2, %0\n\ts_mov_b64 exec, s[22:23]\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(x) : "v"(a), "v"(ring + 2 * threadIdx.x) : "s20", "s21", "s22", "s23", "scc", "memory");
    if (OP == 7) // the same through a per-lane address select (no EXEC writes)
      asm volatile(REP16("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %3, %4, %2, vcc\n\tds_write_b16 %3, %0\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(x) : "v"(a), "v"(ring + 2 * threadIdx.x), "v"(t0), "v"(ring + 512 + 2 * threadIdx.x) : "vcc", "memory");
    if (OP == 8) // the encoder's group as compiled in round 5 (18 instructions, LDS form)
      asm volatile(REP16("v_cmp_ge_u32 s[20:21], %0, %2\n\t"
                         "s_bcnt1_i32_b64 s22, s[20:21]\n\t"
                         "v_mbcnt_lo_u32_b32 %5, s20, 0\n\t"
                         "s_lshl_b32 s22, s22, 1\n\t"
                         "v_mbcnt_hi_u32_b32 %5, s21, %5\n\t"
                         "s_sub_u32 %1, %1, s22\n\t"
                         "v_lshl_add_u32 %5, %5, 1, %1\n\t"
                         "v_lshrrev_b32 %6, 16, %0\n\t"
                         "v_and_b32 %5, 0x3ff, %5\n\t"
                         "v_add_u32 %5, %7, %5\n\t"
                         "s_mov_b64 s[22:23], exec\n\t"
                         "s_and_b64 exec, s[20:21], exec\n\t"
                         "ds_write_b16 %5, %0\n\t"
                         "s_mov_b64 exec, s[22:23]\n\t"
                         "v_cndmask_b32 %0, %0, %6, s[20:21]\n\t"
                         "s_nop 0\n\t"
                         "v_mul_hi_u32 %6, %0, %3\n\t"
                         "v_lshrrev_b32 %6, 3, %6\n\t"
                         "v_mul_u32_u24 %6, %6, %4\n\t"
                         "v_add3_u32 %0, %0, %4, %6\n\t")
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring) : "s20", "s21", "s22", "s23", "scc", "memory");
    if (OP == 9) // a leaner group: word cursor (no shift), v_add_lshl + v_and_or for the ring address, address select instead of EXEC writes
      asm volatile(REP16("v_cmp_ge_u32 vcc, %0, %2\n\t"
                         "v_mbcnt_lo_u32_b32 %5, vcc_lo, 0\n\t"
                         "s_bcnt1_i32_b64 s22, vcc\n\t"
                         "v_mbcnt_hi_u32_b32 %5, vcc_hi, %5\n\t"
                         "s_sub_u32 %1, %1, s22\n\t"
                         "v_add_lshl_u32 %5, %5, %1, 1\n\t"
                         "v_lshrrev_b32 %6, 16, %0\n\t"
                         "v_and_or_b32 %5, %5, %9, %7\n\t"
                         "v_cndmask_b32 %5, %8, %5, vcc\n\t"
                         "ds_write_b16 %5, %0\n\t"
                         "v_cndmask_b32 %0, %0, %6, vcc\n\t"
                         "v_mul_hi_u32 %6, %0, %3\n\t"
                         "v_lshrrev_b32 %6, 3, %6\n\t"
                         "v_mul_u32_u24 %6, %6, %4\n\t"
                         "v_add3_u32 %0, %0, %4, %6\n\t")
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu) : "vcc", "s22", "scc", "memory");
    if (OP == 14) // a whole set as the encoder's loop has it: one wait, the next set's 4 entries and 4 symbols fetched, 4 lean groups, the flush test
      asm volatile(REP16("s_waitcnt lgkmcnt(0)\n\t"
                         "v_lshl_add_u32 v100, %10, 4, 0\n\tv_lshl_add_u32 v101, %10, 4, 0\n\tds_read_b128 v[104:107], v100\n\tds_read_b128 v[108:111], v101\n\t"
                         "v_lshl_add_u32 v100, %10, 4, 0\n\tv_lshl_add_u32 v101, %10, 4, 0\n\tds_read_b128 v[112:115], v100\n\tds_read_b128 v[116:119], v101\n\t"
                         "s_add_i32 s23, s23, 0x100\n\ts_and_b32 s24, s23, 0x1f00\n\tv_add_u32 v100, s24, %10\n\t"
                         "ds_read_u8 v120, v100\n\tds_read_u8 v121, v100 offset:64\n\tds_read_u8 v122, v100 offset:128\n\tds_read_u8 v123, v100 offset:192\n\t"
                         GROUP GROUP GROUP GROUP
                         "s_nop 0\n\ts_cmp_gt_u32 %1, 0\n\ts_cbranch_scc0 1f\n\t1:\n\t")
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu), "v"((threadIdx.x * 37) & 0xff)
                   : "vcc", "s22", "s23", "s24", "scc", "memory", "v100", "v101", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
    if (OP == 15) // the same without the LDS reads
      asm volatile(REP16("s_waitcnt lgkmcnt(0)\n\t"
                         "v_lshl_add_u32 v100, %10, 4, 0\n\tv_lshl_add_u32 v101, %10, 4, 0\n\t"
                         "v_lshl_add_u32 v100, %10, 4, 0\n\tv_lshl_add_u32 v101, %10, 4, 0\n\t"
                         "s_add_i32 s23, s23, 0x100\n\ts_and_b32 s24, s23, 0x1f00\n\tv_add_u32 v100, s24, %10\n\t"
                         GROUP GROUP GROUP GROUP
                         "s_nop 0\n\ts_cmp_gt_u32 %1, 0\n\ts_cbranch_scc0 1f\n\t1:\n\t")
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu), "v"((threadIdx.x * 37) & 0xff)
                   : "vcc", "s22", "s23", "s24", "scc", "memory", "v100", "v101");
    if (OP == 16) // four lean groups with the shift count read out of byte 3 of the table word (SDWA), nothing else
      asm volatile(REP16(GROUP2 GROUP2 GROUP2 GROUP2)
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu), "v"((threadIdx.x * 37) & 0xff)
                   : "vcc", "s22", "s23", "s24", "scc", "memory");
    if (OP == 17) // four lean groups, plain shift
      asm volatile(REP16(GROUP GROUP GROUP GROUP)
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu), "v"((threadIdx.x * 37) & 0xff)
                   : "vcc", "s22", "s23", "s24", "scc", "memory");
    if (OP == 18) // four lean groups + wait for the writes after every four
      asm volatile(REP16(GROUP GROUP GROUP GROUP "s_waitcnt lgkmcnt(0)\n\t")
                   : "+v"(x), "+s"(p) : "v"(a), "v"(b), "v"(c), "v"(t0), "v"(t1), "v"(ring), "v"(ring + 2048 + 2 * threadIdx.x), "s"(0x3ffu), "v"((threadIdx.x * 37) & 0xff)
                   : "vcc", "s22", "s23", "s24", "scc", "memory");
    if (OP == 10) // heap step: two v_readlane, scalar compare, one-lane select
      asm volatile(REP16("v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %0, 6\n\ts_lshr_b32 s22, s20, 8\n\ts_lshr_b32 s23, s21, 8\n\ts_cmp_gt_u32 s23, s22\n\ts_cselect_b32 s20, s21, s20\n\tv_cmp_eq_u32 vcc, 2, %1\n\tv_mov_b32 %2, s20\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t") : "+v"(x) : "v"(threadIdx.x), "v"(t0) : "s20", "s21", "s22", "s23", "vcc", "scc");
    if (OP == 11) // dependent LDS round trip
      asm volatile(REP16("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0xffc, %0\n\t") : "+v"(x), "+v"(e) : : "memory");
    if (OP == 12) // ds_bpermute round trip
      asm volatile(REP16("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(x) : "v"(4 * ((threadIdx.x * 5 + 1) & 63)) : "memory");
    if (OP == 13) // DPP move chain
      asm volatile(REP16("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t") : "+v"(x));
  }
  const uint64_t f = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] = x + a + b + p + t0 + t1 + e;
  if (threadIdx.x == 0)
    ticks[blockIdx.x] = f - s;
}

template <int OP>
void run(const char *name, uint32_t *d, uint64_t *t, int instr)
{
  const int iters = 4000;
  hipLaunchKernelGGL(k<OP>, dim3(64), dim3(64), 0, 0, d, t, 10, 1u);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<OP>, dim3(64), dim3(64), 0, 0, d, t, iters, 1u);
  hipDeviceSynchronize();
  uint64_t h[64];
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 64; i++)
    sum += (double)h[i];
  const double ns = sum / 64 * 10.0 / ((double)iters * 16);
  printf("%-78s %7.2f ns per copy, %2d instructions: %5.2f ns each\n", name, ns, instr, ns / instr);
}

#include <unistd.h>
// the same lean group, but as the encoder meets it: a short kernel (about 100 us) after the device sat idle for 20 ms
void cold(uint32_t *d, uint64_t *t)
{
  for (int trial = 0; trial < 4; trial++)
  {
    usleep(20000);
    hipLaunchKernelGGL(k<9>, dim3(64), dim3(64), 0, 0, d, t, 180, 1u);
    hipDeviceSynchronize();
    uint64_t h[64];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 64; i++)
      sum += (double)h[i];
    printf("lean group, 180 iterations after 20 ms idle: %7.2f ns per copy\n", sum / 64 * 10.0 / (180.0 * 16));
  }
  for (int trial = 0; trial < 3; trial++)
  {
    for (int r = 0; r < 200; r++)
      hipLaunchKernelGGL(k<9>, dim3(64), dim3(64), 0, 0, d, t, 180, 1u);
    hipDeviceSynchronize();
    uint64_t h[64];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 64; i++)
      sum += (double)h[i];
    printf("lean group, 180 iterations, the 200th launch back to back: %7.2f ns per copy\n", sum / 64 * 10.0 / (180.0 * 16));
  }
}

int main()
{
  uint32_t *out;
  uint64_t *t;
  hipMalloc(&out, 64 * 64 * 4);
  hipMalloc(&t, 64 * 8);
  run<0>("dependent v_add_u32", out, t, 1);
  run<1>("two independent v_add_u32", out, t, 2);
  run<2>("dependent v_mul_hi_u32", out, t, 1);
  run<3>("v_cmp -> s_bcnt1 -> s_sub -> v_add (VALU -> SGPR -> SALU -> VALU)", out, t, 4);
  run<4>("two independent s_add_u32", out, t, 2);
  run<5>("two dependent s_add_u32", out, t, 2);
  run<6>("v_cmp, save EXEC, narrow EXEC, ds_write_b16, restore EXEC, v_add", out, t, 6);
  run<7>("v_cmp, v_cndmask address, ds_write_b16, v_add", out, t, 4);
  run<8>("encoder group as compiled (LDS form)", out, t, 20);
  run<9>("encoder group, lean form", out, t, 15);
  run<10>("heap step: 2 v_readlane, 4 SALU, v_cmp, v_mov, v_cndmask", out, t, 9);
  run<11>("ds_read_b32 -> wait -> v_and (LDS round trip)", out, t, 3);
  run<12>("ds_bpermute_b32 -> wait", out, t, 2);
  run<13>("v_mov_b32_dpp row_shr + s_nop 1", out, t, 2);
  run<14>("a whole set: wait, 4 entries + 4 symbols fetched, 4 lean groups, flush test", out, t, 81);
  run<15>("the same without the 8 LDS reads", out, t, 73);
  run<17>("four lean groups", out, t, 56);
  run<16>("four lean groups, shift count by SDWA from byte 3", out, t, 56);
  run<18>("four lean groups + s_waitcnt lgkmcnt(0)", out, t, 57);
  cold(out, t);
  return 0;
}
