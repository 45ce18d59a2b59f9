// gfx950 encoder for mt_ streams with independent fixed-size blocks (include/hsrans_hip.h: hsrans_encode_device).
//
// The reference's encoders are scalar CPU loops (src/mt_rANS32x64_16w_encode.cpp:140-356); their only serial dependency
// between blocks is the coder state that is carried across block boundaries.  With HSRANS_ENC_INDEPENDENT_BLOCKS every
// block starts from fresh states, so a block is one self-contained job:
//
//   K_enc    one wavefront per block: byte histogram (LDS atomics) -> normalisation identical to hist.cpp:16-215 ->
//            backward rANS pass, lane j = coder state j, words stored top-down into the block's scratch slot, then the
//            block header [size][skip][states][counts] is put in front of them: the slot ends with the block's image
//   K_scan   exclusive scan of the image sizes -> position of every block in the stream, file header [n][total]
//   K_gather copies every image to its place
//
// The result is byte-identical to the host encoder (hsrans_host.cpp encode(), same flag).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hsrans_encode.h"

namespace hsrans
{
namespace
{

constexpr uint32_t kWavesPerWG = 4;
constexpr uint32_t kChunk = 2048; // input bytes staged in LDS per step (two buffers)

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }
__device__ __forceinline__ uint32_t enc_lane_to_byte(uint32_t j) { return (j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1); }
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct __attribute__((packed, aligned(2))) U64a2
{
  uint64_t v;
};
struct __attribute__((packed, aligned(2))) U32a2
{
  uint32_t v;
};

struct WaveLds
{
  uint32_t scaled[256];  // raw counts, then the normalised counts
  uint32_t order[256];   // count << 8 | symbol, heap-sorted by count
  uint4 table[256];      // {x_max, bias, rcp, cmpl | shift << 16}
  uint8_t stage[2][kChunk];
};

// ---- hist.cpp:16-215 on one lane ----------------------------------------------------------------------------------
// textbook heap sort by count; the order of equal counts it happens to produce decides which symbols are adjusted,
// so it is replayed exactly (hsrans_host.cpp heap_sift); the sifted element stays in a register
__device__ void heap_sift(uint32_t *a, int n, int root)
{
  const uint32_t val = a[root];
  while (true)
  {
    const int l = 2 * root + 1, r = l + 1;
    if (l >= n)
      break;
    const uint32_t al = a[l];
    const uint32_t ar = r < n ? a[r] : 0;
    int big = root;
    uint32_t kb = val >> 8, vb = val;
    if ((al >> 8) > kb)
    {
      big = l;
      kb = al >> 8;
      vb = al;
    }
    if (r < n && (ar >> 8) > kb)
    {
      big = r;
      vb = ar;
    }
    if (big == root)
      break;
    a[root] = vb;
    root = big;
  }
  a[root] = val;
}

__device__ int first_at_least_two(const uint32_t *order, const uint32_t *scaled, int from, int fallback)
{
  for (int i = from; i < 256; i++)
    if (scaled[order[i] & 0xFF] >= 2)
      return i;
  return fallback;
}

__device__ void adjust_counts_one_lane(uint32_t *scaled, uint32_t *order, uint32_t sum, uint32_t target)
{
  for (int i = 127; i >= 0; i--)
    heap_sift(order, 256, i);
  for (int i = 255; i > 0; i--)
  {
    const uint32_t t = order[0];
    order[0] = order[i];
    order[i] = t;
    heap_sift(order, i, 0);
  }
  int lo = first_at_least_two(order, scaled, 0, 0);
  while (sum > target)
  {
    bool done = false;
    for (int i = lo; i < 256 && !done; i++)
    {
      scaled[order[i] & 0xFF]--;
      done = --sum == target;
    }
    if (!done)
      lo = first_at_least_two(order, scaled, lo, lo);
  }
  while (sum < target)
  {
    bool done = false;
    for (int i = 255; i >= lo && !done; i--)
    {
      scaled[order[i] & 0xFF]++;
      done = ++sum == target;
    }
    if (!done)
      lo = first_at_least_two(order, scaled, lo, lo);
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
  for (int d = 32; d >= 1; d >>= 1)
    v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ uint4 load16_guarded(const uint8_t *in, uint64_t pos, uint64_t n)
{
  if (pos + 16 <= n)
    return *(const uint4 *)(in + pos);
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < 16; k++)
    if (pos + k < n)
      w[k >> 2] |= (uint32_t)in[pos + k] << (8 * (k & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <uint32_t S, bool FULL>
__device__ __forceinline__ void encode_group(uint32_t &x, const WaveLds &L, const uint8_t *stage, uint32_t group_off, uint32_t valid, uint8_t *&p, uint32_t lane,
                                             uint32_t byte_in_group)
{
  // FULL: all S lanes code a symbol; else only those whose byte exists (the file's last, partial group)
  const bool active = lane < S && (FULL || byte_in_group < valid);
  const uint32_t sym = stage[group_off + byte_in_group];
  const uint4 e = L.table[sym];
  const bool emit = active && x >= e.x;
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(emit);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
  p -= 2 * (uint32_t)__builtin_popcountll(mask);
  uint32_t v = x;
  if (emit)
  {
    *(uint16_t *)(p + 2 * rank) = (uint16_t)x; // lane S-1's word goes last in memory (rANS32x64_16w.cpp:65-99)
    v = x >> 16;
  }
  const uint32_t q = __umulhi(v, e.z) >> (e.w >> 16);
  const uint32_t nx = v + e.y + q * (e.w & 0xFFFFu);
  x = active ? nx : x;
}

template <uint32_t S>
__global__ void __launch_bounds__(64 * kWavesPerWG) k_encode_blocks(EncParams ep)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t b = blockIdx.x * kWavesPerWG + wave;
  if (b >= ep.n_blocks)
    return;
  WaveLds &L = *(WaveLds *)(lds_raw + (size_t)wave * sizeof(WaveLds));
  const uint64_t begin = (uint64_t)b * ep.block;
  const uint64_t end = b + 1 == ep.n_blocks ? ep.n : begin + ep.block;
  const uint32_t size = (uint32_t)(end - begin);
  const uint8_t *in = ep.in;
  uint8_t *slot_end = ep.scratch + (uint64_t)(b + 1) * ep.slot_bytes;

  // ---- byte histogram ----
  for (uint32_t k = 0; k < 4; k++)
    L.scaled[lane * 4 + k] = 0;
  wave_sync();
  for (uint32_t off = lane * 16; off < size; off += 1024)
  {
    const uint4 d = load16_guarded(in, begin + off, end);
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    const uint32_t have = size - off < 16 ? size - off : 16;
    for (uint32_t k = 0; k < 16; k++)
      if (k < have)
        atomicAdd(&L.scaled[(w[k >> 2] >> (8 * (k & 3))) & 0xFF], 1u);
  }
  wave_sync();
  uint32_t raw[4];
  uint32_t present = 0;
  for (uint32_t k = 0; k < 4; k++)
  {
    raw[k] = L.scaled[lane * 4 + k];
    present += raw[k] != 0;
  }
  const uint32_t distinct = wave_sum(present);
  if (distinct == 1) // single-symbol block: only the marker word (mt_rANS32x64_16w_encode.cpp:289-295)
  {
    if (present)
    {
      uint32_t sym = lane * 4;
      for (uint32_t k = 0; k < 4; k++)
        if (raw[k])
          sym = lane * 4 + k;
      const uint64_t marker = (uint64_t)size | ((uint64_t)1 << 63) | ((uint64_t)sym << 54);
      ((U64a2 *)(slot_end - 8))->v = marker;
      ep.image_bytes[b] = 8;
    }
    return;
  }

  // ---- normalisation (hist.cpp:16-215; hsrans_host.cpp normalize_counts) ----
  const uint32_t target = 1u << ep.bits;
  const float factor = (float)target / (float)(uint64_t)size;
  uint32_t sc[4];
  uint32_t part = 0;
  for (uint32_t k = 0; k < 4; k++)
  {
    const float v = __fmul_rn((float)raw[k], factor); // one rounding per operation, like the reference's build
    uint32_t c = (uint32_t)(uint16_t)__fadd_rn(v, 0.5f);
    if (c == 0 && raw[k] != 0)
      c = 1;
    sc[k] = c;
    part += c;
    L.scaled[lane * 4 + k] = c;
    L.order[lane * 4 + k] = (c << 8) | (lane * 4 + k);
  }
  const uint32_t sum = wave_sum(part);
  wave_sync();
  if (sum != target)
  {
    if (lane == 0)
      adjust_counts_one_lane(L.scaled, L.order, sum, target);
    wave_sync();
    part = 0;
    for (uint32_t k = 0; k < 4; k++)
    {
      sc[k] = L.scaled[lane * 4 + k];
      part += sc[k];
    }
  }
  // exclusive prefix over the 256 counts: lane-local then across lanes
  uint32_t incl = part;
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t o = __shfl_up(incl, d, 64);
    if ((int)lane >= d)
      incl += o;
  }
  uint32_t cum = incl - part;
  for (uint32_t k = 0; k < 4; k++)
  {
    const uint32_t freq = sc[k];
    uint4 e;
    e.x = freq << (31 - ep.bits); // emit when x >= ((2^15 >> bits) << 16) * freq
    if (freq < 2)
    {
      e.y = cum + target - 1; // q = x - 1 for freq == 1
      e.z = 0xFFFFFFFFu;
      e.w = (target - freq);
      if (freq == 0)
        e.x = 0xFFFFFFFFu;
    }
    else
    {
      uint32_t shift = 32 - __clz(freq - 1); // smallest shift with freq <= 1 << shift
      e.y = cum;
      e.z = (uint32_t)((((uint64_t)1 << (shift + 31)) + freq - 1) / freq);
      e.w = (target - freq) | ((shift - 1) << 16);
    }
    L.table[lane * 4 + k] = e;
    cum += freq;
  }

  // ---- backward rANS pass over the block (rANS32x64_16w.cpp:34-166) ----
  const uint32_t n_chunks = (size + kChunk - 1) / kChunk;
  const uint32_t byte_in_group = enc_lane_to_byte(lane) & (S - 1);
  uint32_t x = 1u << 15;
  uint8_t *p = slot_end;
  uint4 pre0, pre1;
  {
    const uint32_t c = n_chunks - 1;
    pre0 = load16_guarded(in, begin + (uint64_t)c * kChunk + lane * 16, end);
    pre1 = load16_guarded(in, begin + (uint64_t)c * kChunk + 1024 + lane * 16, end);
    *(uint4 *)(L.stage[c & 1] + lane * 16) = pre0;
    *(uint4 *)(L.stage[c & 1] + 1024 + lane * 16) = pre1;
  }
  for (uint32_t c = n_chunks; c-- > 0;)
  {
    if (c > 0)
    {
      pre0 = load16_guarded(in, begin + (uint64_t)(c - 1) * kChunk + lane * 16, end);
      pre1 = load16_guarded(in, begin + (uint64_t)(c - 1) * kChunk + 1024 + lane * 16, end);
    }
    wave_sync();
    const uint8_t *stage = L.stage[c & 1];
    const uint32_t bytes = size - c * kChunk < kChunk ? size - c * kChunk : kChunk;
    uint32_t g = (bytes + S - 1) / S; // groups in this chunk
    if (bytes % S != 0)               // only the file's last group can be partial
    {
      g--;
      encode_group<S, false>(x, L, stage, g * S, bytes - g * S, p, lane, byte_in_group);
    }
    while (g % 4 != 0)
    {
      g--;
      encode_group<S, true>(x, L, stage, g * S, S, p, lane, byte_in_group);
    }
    while (g != 0)
    {
      g -= 4;
#pragma unroll
      for (int k = 3; k >= 0; k--)
        encode_group<S, true>(x, L, stage, (g + k) * S, S, p, lane, byte_in_group);
    }
    if (c > 0)
    {
      *(uint4 *)(L.stage[(c - 1) & 1] + lane * 16) = pre0;
      *(uint4 *)(L.stage[(c - 1) & 1] + 1024 + lane * 16) = pre1;
    }
  }

  // ---- block header in front of the words: [size u64][skip u64][states S x u32][counts 256 x u16] ----
  const uint32_t words_bytes = (uint32_t)(slot_end - p);
  constexpr uint32_t kHeader = 16 + 4 * S + 512;
  uint8_t *h = p - kHeader;
  // skip: uint16 units from the state array to the next block header, minus one; the last block's is one less
  // (hsrans_host.cpp encode(); mt_rANS32x64_16w_encode.cpp:149,277)
  const uint64_t skip = (uint64_t)(4 * S + 512 + words_bytes) / 2 - 1 - (b + 1 == ep.n_blocks ? 1 : 0);
  if (lane == 0)
  {
    ((U64a2 *)h)->v = (uint64_t)size;
    ((U64a2 *)(h + 8))->v = skip;
    ep.image_bytes[b] = (uint64_t)(slot_end - h);
  }
  if (lane < S)
    ((U32a2 *)(h + 16 + 4 * lane))->v = x;
  for (uint32_t k = 0; k < 4; k++)
    *(uint16_t *)(h + 16 + 4 * S + 2 * (lane * 4 + k)) = (uint16_t)sc[k];
}

// ---- K_scan: one workgroup ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_scan_images(EncParams ep)
{
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry_s;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry_s = 16; // file header
  __syncthreads();
  for (uint32_t base = 0; base < ep.n_blocks; base += 1024)
  {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < ep.n_blocks ? ep.image_bytes[i] : 0;
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint64_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d)
        incl += o;
    }
    if (lane == 63)
      wave_tot[wave] = incl;
    __syncthreads();
    uint64_t before = carry_s;
    for (uint32_t w = 0; w < wave; w++)
      before += wave_tot[w];
    if (i < ep.n_blocks)
      ep.image_off[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    const uint64_t total = carry_s;
    ep.result[0] = total;
    ep.result[1] = total <= ep.out_cap ? 1 : 0;
    if (total <= ep.out_cap)
    {
      ((uint64_t *)ep.out)[0] = ep.n;
      ((uint64_t *)ep.out)[1] = total;
    }
  }
}

// ---- K_gather: one workgroup per block image; source and destination are only 2-byte aligned -------------------------
struct __attribute__((packed, aligned(2))) U128a2
{
  uint32_t v[4];
};

__global__ void __launch_bounds__(256) k_gather_images(EncParams ep)
{
  if (ep.result[1] == 0)
    return;
  const uint32_t b = blockIdx.x;
  const uint64_t bytes = ep.image_bytes[b];
  const uint8_t *src = ep.scratch + (uint64_t)(b + 1) * ep.slot_bytes - bytes;
  uint8_t *dst = ep.out + ep.image_off[b];
  // head: up to the first 16-byte boundary of dst
  uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  if (head > bytes)
    head = bytes;
  for (uint64_t i = threadIdx.x * 2; i < head; i += 512)
    *(uint16_t *)(dst + i) = *(const uint16_t *)(src + i);
  const uint64_t body = (bytes - head) / 16;
  for (uint64_t i = threadIdx.x; i < body; i += 256)
  {
    const U128a2 v = *(const U128a2 *)(src + head + i * 16);
    *(uint4 *)(dst + head + i * 16) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
  }
  for (uint64_t i = head + body * 16 + threadIdx.x * 2; i < bytes; i += 512)
    *(uint16_t *)(dst + i) = *(const uint16_t *)(src + i);
}

} // namespace

uint32_t encode_block_count(uint64_t n, uint64_t block, uint32_t S)
{
  // hsrans_host.cpp split_blocks: the last block absorbs a remainder shorter than one group
  uint64_t count = (n + block - 1) / block;
  if (count > 1 && n - (count - 1) * block < S)
    count--;
  return count > 0xFFFFFFFFull ? 0 : (uint32_t)count;
}

uint64_t encode_slot_bytes(uint64_t block, uint32_t S)
{
  // every symbol emits at most one 16-bit word; the last block can be up to S-1 symbols longer
  const uint64_t need = 2 * (block + S) + 16 + 4 * (uint64_t)S + 512;
  return (need + 255) / 256 * 256;
}

hipError_t launch_encode(const EncParams &ep, hipStream_t stream)
{
  static bool prepared = false;
  const size_t lds = sizeof(WaveLds) * kWavesPerWG;
  if (!prepared)
  {
    hipError_t e = hipFuncSetAttribute((const void *)k_encode_blocks<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_encode_blocks<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
      return e;
    prepared = true;
  }
  const uint32_t grid = (ep.n_blocks + kWavesPerWG - 1) / kWavesPerWG;
  if (ep.S == 64)
    hipLaunchKernelGGL(k_encode_blocks<64>, dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  else
    hipLaunchKernelGGL(k_encode_blocks<32>, dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  hipLaunchKernelGGL(k_scan_images, dim3(1), dim3(1024), 0, stream, ep);
  hipLaunchKernelGGL(k_gather_images, dim3(ep.n_blocks), dim3(256), 0, stream, ep);
  return hipGetLastError();
}

} // namespace hsrans
