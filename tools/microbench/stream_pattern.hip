// What does the decoder's stream-read pattern cost by itself?  8,192 wavefronts (512 workgroups x 16), each reading its own
// contiguous region of a buffer through a 4-slot LDS ring in chunks of LANES x 16 bytes by LDS-DMA, AHEAD chunks requested in
// front of the one in use, with PACE cycles of s_sleep per chunk standing in for the decode work.  Buffers are rotated (4 x 64 MB
// > Infinity Cache with the 400 MB "output" written between runs) or replayed.  Prints GB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern && ./stream_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// STORES: 0 none; 1 three 256-byte streaming stores per chunk (the decoder writes 1.55 bytes per byte it reads), waits exact
// (vmcnt counts the stores issued after the awaited request); 2 the same stores, waits strict (vmcnt(AHEAD): every store drained)
template <int LANES, int AHEAD, int STORES = 0>
__global__ void __launch_bounds__(1024) k_stream(const uint8_t *src, uint64_t bytes_per_wave, uint32_t pace, uint32_t *sink, uint8_t *dst = nullptr)
{
  extern __shared__ u32x4 smem[];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + wave;
  constexpr uint32_t kChunk = LANES * 16;
  const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + wave * 4 * kChunk);
  const uint64_t addr = (uint64_t)(uintptr_t)src + (uint64_t)w * bytes_per_wave;
  u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xFFFF);
  rs.z = (uint32_t)bytes_per_wave;
  rs.w = 0x00020000;
  const uint32_t chunks = (uint32_t)(bytes_per_wave / kChunk);
  auto request = [&](uint32_t c) {
    const uint32_t voff = c * kChunk + lane * 16;
    const uint32_t dst = lds + (c & 3) * kChunk;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_mov_b64 exec, -1"
                 :
                 : "v"(voff), "s"(dst), "s"(rs), "s"(LANES == 64 ? ~0ull : (1ull << LANES) - 1)
                 : "memory");
  };
  for (uint32_t c = 0; c < AHEAD && c < chunks; c++)
    request(c);
  uint32_t acc = 0;
  for (uint32_t c = 0; c < chunks; c++)
  {
    if (c + AHEAD < chunks)
      request(c + AHEAD);
    // chunk c has landed when at most AHEAD younger requests are outstanding
    if (STORES == 1)
    { // younger than chunk c's request: AHEAD requests and 3 stores per chunk since
      if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (AHEAD == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    else
    {
      if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (AHEAD == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    if (STORES)
    {
      uint8_t *o = dst + ((uint64_t)w * (bytes_per_wave / kChunk) + c) * 768;
      for (int k = 0; k < 3; k++)
        asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(lane * 4 + k * 256), "v"(acc), "s"(o) : "memory");
    }
    acc += ((const uint32_t *)((const uint8_t *)smem + wave * 4 * kChunk + (c & 3) * kChunk))[lane % (LANES * 4)];
    for (uint32_t p = 0; p < pace; p += 64)
      __builtin_amdgcn_s_sleep(1); // 64 cycles
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678)
    sink[w] = acc;
}

template <int LANES, int AHEAD, int STORES = 0>
static void run(const char *label, std::vector<uint8_t *> &bufs, uint8_t *scratch, size_t scratch_bytes, size_t bytes, uint32_t pace, bool rotate, uint32_t *sink, uint8_t *dst = nullptr)
{
  const uint32_t grid = 512, waves = 16;
  const uint64_t per_wave = bytes / (grid * waves) / (LANES * 16) * (LANES * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f, sum = 0;
  const int reps = 12;
  for (int i = 0; i < reps + 2; i++)
  {
    if (rotate && dst == nullptr)
      hipMemsetAsync(scratch, i, scratch_bytes, 0); // 400 MB written: what was in the Infinity Cache is gone
    // (with an output buffer the rotation over the 4 sources alone is what the decoder benchmark does: 256 MB of streams + the output)
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_stream<LANES, AHEAD, STORES>), dim3(grid), dim3(waves * 64), waves * 4 * LANES * 16, 0, bufs[rotate ? i % bufs.size() : 0], per_wave, pace, sink, dst);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2)
    {
      best = ms < best ? ms : best;
      sum += ms;
    }
  }
  const double moved = (double)per_wave * grid * waves;
  printf("%-58s lanes %2d ahead %d pace %4u %s: mean %7.1f us (%.2f TB/s)  best %7.1f us (%.2f TB/s)\n", label, LANES, AHEAD, pace, rotate ? "cold" : "warm", sum / reps * 1e3,
         moved / (sum / reps * 1e-3) / 1e12, best * 1e3, moved / (best * 1e-3) / 1e12);
}

int main()
{
  const size_t bytes = 64u << 20;
  std::vector<uint8_t *> bufs(4);
  for (auto &b : bufs)
  {
    hipMalloc((void **)&b, bytes);
    hipMemset(b, 1, bytes);
  }
  uint8_t *scratch;
  const size_t scratch_bytes = 400u << 20;
  hipMalloc((void **)&scratch, scratch_bytes);
  uint32_t *sink;
  hipMalloc((void **)&sink, 8192 * 4);
  hipFuncSetAttribute((const void *)k_stream<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rotate = 0; rotate < 2; rotate++)
  {
    // pace 0: as fast as the memory system goes; pace 4800 cycles per 512-byte chunk = the decoder's own rate (2.3 us per chunk)
    for (uint32_t pace : {0u, 2400u, 4800u})
    {
      run<32, 1>("512 B chunks", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink);
      run<32, 2>("512 B chunks", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink);
      run<32, 3>("512 B chunks", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink);
      run<64, 2>("1 KiB chunks", bufs, scratch, scratch_bytes, bytes, pace * 2, rotate, sink);
      run<64, 3>("1 KiB chunks", bufs, scratch, scratch_bytes, bytes, pace * 2, rotate, sink);
    }
  }
  // with the decoder's stores (to ONE 100 MB output buffer that stays in the Infinity Cache, like "streams rotated, one output")
  uint8_t *dst;
  hipMalloc((void **)&dst, 100u << 20);
  for (int rotate = 0; rotate < 2; rotate++)
    for (uint32_t pace : {0u, 2400u})
    {
      run<32, 2, 1>("512 B chunks + stores, exact waits", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink, dst);
      run<32, 2, 2>("512 B chunks + stores, strict waits", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink, dst);
      run<32, 3, 1>("512 B chunks + stores, exact waits", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink, dst);
      run<32, 3, 2>("512 B chunks + stores, strict waits", bufs, scratch, scratch_bytes, bytes, pace, rotate, sink, dst);
    }
  return 0;
}
