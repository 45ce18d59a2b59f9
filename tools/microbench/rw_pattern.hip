// What can HBM sustain for the decoder's traffic SHAPE when nothing is cache-resident?  8,192 wavefronts (512 workgroups x 16),
// each reading its own contiguous 8 KB region in CHUNK-byte LDS-DMA requests (AHEAD in flight) and writing 1.5 bytes per byte
// read to its own contiguous output region, either as the decoder does (256 B per store instruction: one dword per lane) or as
// 768 B per instruction (dwordx4 on 48 lanes).  PACE cycles of s_sleep per 512 B read stand in for the decode work.
// Modes: one pair replayed (everything Infinity-Cache resident); 4 sources rotated, one output; 4 sources and 4 outputs rotated;
// the same with 512 MB written between launches.
//   hipcc --offload-arch=gfx950 -O3 rw_pattern.hip -o rw_pattern && ./rw_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// WIDE: 0 = three 256-byte stores (a dword per lane) per 512 bytes read; 1 = one 768-byte store (dwordx4 on 48 lanes)
// Requests AHEAD chunks in front and waits for "all but the operations younger than chunk c's request" (exact count: the in-order
// rule the decoder lives with too); stores are fire-and-forget.
// SPLIT: even waves only read (their own and their odd neighbour's region), odd waves only write (both outputs), each at half
// the pace per chunk: the same traffic in the same time, but no wave has loads and stores in one in-order vmcnt queue
template <int LANES, int AHEAD, int WIDE, bool SPLIT = false>
__global__ void __launch_bounds__(1024) k_rw_relaxed(const uint8_t *src, uint64_t bytes_per_wave, uint32_t pace, uint32_t *sink, uint8_t *dst)
{
  extern __shared__ u32x4 smem[];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t w = blockIdx.x * (blockDim.x >> 6) + wave;
  const bool reader = !SPLIT || (w & 1) == 0, writer = !SPLIT || (w & 1) == 1;
  if (SPLIT)
  {
    w >>= 1;
    bytes_per_wave *= 2;
    pace /= 2;
  }
  constexpr uint32_t kChunk = LANES * 16;
  constexpr uint32_t kSlots = AHEAD < 4 ? 4 : 8;
  const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + wave * kSlots * kChunk);
  const uint64_t addr = (uint64_t)(uintptr_t)src + (uint64_t)w * bytes_per_wave;
  u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xFFFF);
  rs.z = (uint32_t)bytes_per_wave;
  rs.w = 0x00020000;
  const uint32_t chunks = (uint32_t)(bytes_per_wave / kChunk);
  auto request = [&](uint32_t c) {
    const uint32_t voff = c * kChunk + lane * 16;
    const uint32_t d = lds + (c & (kSlots - 1)) * kChunk;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_mov_b64 exec, -1"
                 :
                 : "v"(voff), "s"(d), "s"(rs), "s"(LANES == 64 ? ~0ull : (1ull << LANES) - 1)
                 : "memory");
  };
  for (uint32_t c = 0; reader && c < AHEAD && c < chunks; c++)
    request(c);
  uint32_t acc = 0;
  uint8_t *o = dst + (uint64_t)w * (bytes_per_wave / 2 * 3);
  uint32_t units = 0;
  constexpr int kStoresPerChunk = WIDE == 2 ? 0 : WIDE ? 1 : 3; // per 512-byte unit
  constexpr int kYoung = AHEAD * (1 + kStoresPerChunk * (int)(kChunk / 512));
  for (uint32_t c = 0; c < chunks; c++)
  {
    if (reader && c + AHEAD < chunks)
      request(c + AHEAD);
    // younger than chunk c's request: AHEAD requests and the stores of AHEAD chunks
    static_assert(kYoung <= 63, "vmcnt is 6 bits");
    if (SPLIT)
    {
      if (reader)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
    }
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYoung) : "memory");
    acc += ((const uint32_t *)((const uint8_t *)smem + wave * kSlots * kChunk + (c & (kSlots - 1)) * kChunk))[lane % (LANES * 4)];
    for (uint32_t u = 0; u < kChunk / 512; u++, units++)
    {
      if (WIDE == 2 || !writer)
      {
      }
      else if (WIDE == 0)
      {
        for (int k = 0; k < 3; k++)
          asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(lane * 4 + k * 256), "v"(acc), "s"(o) : "memory");
        o += 768;
      }
      else
      {
        const u32x4 v = {acc, acc, acc, acc};
        asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_mov_b64 exec, -1" : : "v"(lane * 16), "v"(v), "s"(o), "s"((1ull << 48) - 1) : "memory");
        o += 768;
      }
      for (uint32_t p = 0; p < pace; p += 64)
        __builtin_amdgcn_s_sleep(1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678)
    sink[w] = acc;
}

struct Bufs
{
  std::vector<uint8_t *> src, dst;
  uint8_t *scratch;
  size_t scratch_bytes;
  uint32_t *sink;
};

template <int LANES, int AHEAD, int WIDE, bool SPLIT = false>
static void run(const char *label, Bufs &b, size_t bytes, uint32_t pace, int mode)
{
  const uint32_t grid = 512, waves = 16;
  const uint64_t per_wave = bytes / (grid * waves) / 1024 * 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void *)k_rw_relaxed<LANES, AHEAD, WIDE, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  float best = 1e30f, sum = 0;
  const int reps = 10;
  for (int i = 0; i < reps + 2; i++)
  {
    if (mode == 3)
      hipMemsetAsync(b.scratch, i, b.scratch_bytes, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rw_relaxed<LANES, AHEAD, WIDE, SPLIT>), dim3(grid), dim3(waves * 64), waves * (AHEAD < 4 ? 4 : 8) * LANES * 16, 0, b.src[mode >= 1 ? i % b.src.size() : 0], per_wave, pace, b.sink,
                       b.dst[mode >= 2 ? i % b.dst.size() : 0]);
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
  const double moved = (double)per_wave * grid * waves * 2.5;
  printf("%-34s read %4d B x ahead %d, pace %4u, %-28s: mean %6.1f us (%.2f TB/s read+written)  best %6.1f us\n", label, LANES * 16, AHEAD, pace,
         mode == 0 ? "warm" : mode == 1 ? "sources rotated" : mode == 2 ? "sources and outputs rotated" : "rotated + 512 MB written", sum / reps * 1e3,
         moved / (sum / reps * 1e-3) / 1e12, best * 1e3);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main()
{
  const size_t bytes = 64u << 20;
  Bufs b;
  b.src.resize(4);
  b.dst.resize(4);
  for (auto &p : b.src)
  {
    hipMalloc((void **)&p, bytes);
    hipMemset(p, 1, bytes);
  }
  for (auto &p : b.dst)
  {
    hipMalloc((void **)&p, bytes / 2 * 3 + (1 << 20));
    hipMemset(p, 2, bytes / 2 * 3);
  }
  b.scratch_bytes = 512u << 20;
  hipMalloc((void **)&b.scratch, b.scratch_bytes);
  hipMalloc((void **)&b.sink, 8192 * 4);
  for (int mode = 0; mode < 4; mode++)
    for (uint32_t pace : {0u, 2400u})
    {
      run<32, 3, 2>("no stores", b, bytes, pace, mode);
      run<32, 3, 0>("256 B stores", b, bytes, pace, mode);
      run<32, 3, 1>("768 B stores", b, bytes, pace, mode);
      run<64, 3, 0>("256 B stores, 1 KiB reads", b, bytes, pace, mode);
      run<32, 6, 0>("256 B stores, deeper ring", b, bytes, pace, mode);
      run<32, 3, 0, true>("256 B stores, split waves", b, bytes, pace, mode);
    }
  return 0;
}
