// Host SIMD decoders for the hypersonic-rANS 32-bit-state / 16-bit-word formats, with runtime dispatch.
//
// The GPU is the product's decode path for everything that has parallelism to offer; what it cannot speed up is a stream
// that is ONE dependent chain (a raw or block_ stream without a sidecar index): one wavefront decodes that at ~0.65 GB/s,
// a third of what one CPU core does.  This file is the counterpart of the reference's host decoders and of its runtime
// dispatcher (block_rANS32x64_16w_decode.cpp:130-152: `_DetectCPUFeatures()` then AVX-512 / AVX2 / scalar): written from
// the format (SURVEY.md §8 "Wire formats"), not from the reference's SIMD code, and used for
//   (1) the `*_decode_auto_N` drop-in entries: single-chain streams without an index go to the host, everything else to the GPU;
//   (2) hsrans_index_build_host: the pass that records checkpoints of a foreign raw stream (3x faster than one wavefront);
//   (3) hsrans_decode_cpu: an in-run CPU comparator (bench.py's cpu_baseline kind "port"), single- or multi-threaded.
// The GPU entries (hsrans_decode_host / hsrans_decode_device, `*_decode_hip_N`) never come here: they fail without a GPU.
// Nothing under oracle/ is used.  Tests: tests/test_cpu_decoder.py (all dispatch levels against the oracle and the golden
// vectors of the real reference).
//
// The decode step (rANS32x64_16w.cpp:17-30,223-250), per group of S symbols, for state j = 0..S-1 in order:
//     slot = x & (2^b - 1);  sym = cumulInv[slot];  x = (x >> b) * freq[sym] + slot - cumul[sym];
//     if (x < 2^15) x = x << 16 | *readHead++;            out[i + idx2idx[j]] = sym
// SIMD mapping: 8 (AVX2) / 16 (AVX-512) consecutive states per vector.  The renormalisation is a "compress/expand":
// the lanes whose x fell below 2^15 take the next words of the stream in lane order = an expand of a contiguous load
// (AVX-512: vpexpandd under the compare mask; AVX2: vpermd by a 256-entry table indexed with the movemask).
// idx2idx is the bit permutation j -> (j&0x23)|((j&4)<<2)|((j&0x18)>>1): two saturating packs (dword -> word -> byte) of
// the vectors' symbols put them in output order (AVX-512: plus one dword permute).
#include "hsrans_cpu.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "hsrans_host.h"

namespace hsrans
{
namespace cpu
{

namespace
{
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t lane_to_byte(uint32_t j) { return (j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1); }

// ---------------------------------------------------------------------------------------------------------------
// decode tables (chosen per histogram width and state count; `mode` tells the loops which one was built)
//   kPacked    bits <= 11:  uint32 per slot = sym | (slot - cumul) << 8 | freq << 20: freq is one shift away from the gather
//   kPackedM1  bits == 12:  the same with freq - 1 (4096 must fit 12 bits)
//   kCompact   bits >= 13:  uint8 sym[2^bits] + uint32 {freq | cumul << 16}[256] — the reference's dec2 layout (hist.h:42-47):
//                           two dependent gathers, both from tables that stay in L1 (8..32 KiB + 1 KiB)
//   kBySlot    bits >= 13, 32-state chains: uint32 per slot = sym | (slot - cumul) << 8 and uint16 (freq - 1) per slot, gathered
//                           side by side: a 32-state loop is bound by the latency of gather -> multiply -> renormalise, and a
//                           second gather BESIDE the first is shorter than one behind it (64 states are throughput-bound: there
//                           the bigger tables cost more in cache misses than the shorter chain gains — measured both ways)
// ---------------------------------------------------------------------------------------------------------------
enum TableMode { kPacked = 1, kPackedM1 = 3, kCompact = 0, kBySlot = 2 };

inline bool cpu_is_amd()
{
  static const bool amd = [] { __builtin_cpu_init(); return __builtin_cpu_is("amd") != 0; }();
  return amd;
}

inline int wide_mode_override() // HSRANS_CPU_WIDE_MODE=0|2: force kCompact / kBySlot for bits >= 13 (A/B runs)
{
  static const int v = [] { const char *e = getenv("HSRANS_CPU_WIDE_MODE"); return e ? atoi(e) : -1; }();
  return v;
}

struct Table
{
  uint32_t bits = 0;
  int mode = kPacked;
  std::vector<uint32_t> slot;      // kPacked / kPackedM1 / kBySlot
  std::vector<uint16_t> slot_freq; // kBySlot
  std::vector<uint8_t> sym8;       // bits >= 13 (the scalar loop uses the compact pair at every such width)
  alignas(64) uint32_t fc[256];    // freq | cumul << 16
  bool build(const uint8_t *counts_le16, uint32_t b, uint32_t S)
  {
    bits = b;
    // (kBySlot pays on Intel cores only: Zen 5 runs the compact pair faster at every width, 1,822 against 1,452 MiB/s at 14 bits)
    mode = b <= 11 ? kPacked : b == 12 ? kPackedM1 : (S == 32 && b <= 14 && !cpu_is_amd()) ? kBySlot : kCompact;
    if (b >= 13 && (wide_mode_override() == kCompact || wide_mode_override() == kBySlot))
      mode = wide_mode_override();
    const uint32_t total = 1u << b;
    // (+16: gathers never leave the allocation even with junk indices in dead lanes)
    if (mode != kCompact)
      slot.resize(total + 16);
    if (mode == kBySlot)
      slot_freq.resize(total + 16);
    if (b >= 13)
      sym8.resize(total + 16);
    uint32_t cum = 0;
    for (uint32_t s = 0; s < 256; s++)
    {
      const uint32_t f = rd16(counts_le16 + 2 * s);
      fc[s] = f | (cum << 16);
      if (cum + f > total)
        return false;
      if (mode == kPacked)
        for (uint32_t k = 0; k < f; k++)
          slot[cum + k] = s | (k << 8) | (f << 20);
      else if (mode == kPackedM1)
        for (uint32_t k = 0; k < f; k++)
          slot[cum + k] = s | (k << 8) | ((f - 1) << 20);
      else if (mode == kBySlot)
        for (uint32_t k = 0; k < f; k++)
        {
          slot[cum + k] = s | (k << 8);
          slot_freq[cum + k] = (uint16_t)(f - 1); // (f - 1: 2^15 must fit; the dword gather at scale 2 reads two entries, the low one counts)
        }
      if (b >= 13)
        memset(sym8.data() + cum, (int)s, f);
      cum += f;
    }
    return cum == total; // inplace_complete_hist (hist.cpp:308-324): the decoder returns 0 otherwise
  }
};

struct Cursor
{
  const uint8_t *p;   // next word
  const uint8_t *end; // first byte that must not be read
  inline uint32_t next()
  {
    uint32_t w = 0;
    if (p + 2 <= end) // a truncated stream decodes zeros (as the GPU's bounds-checked stream window does), never reads outside
      w = rd16(p);
    p += 2;
    return w;
  }
};

// scalar: one group (lanes whose byte exists: `limit` = symbols of this group, S for a whole group)
inline void group_scalar(uint32_t *x, const Table &t, Cursor &c, uint8_t *out, uint32_t S, uint32_t limit)
{
  const uint32_t mask = (1u << t.bits) - 1;
  for (uint32_t j = 0; j < S; j++)
  {
    const uint32_t p = lane_to_byte(j);
    if (p >= limit)
      continue;
    const uint32_t v = x[j], sl = v & mask;
    uint32_t nx, e;
    if (t.bits <= 12)
    {
      e = t.slot[sl];
      nx = (v >> t.bits) * ((e >> 20) + (t.bits == 12 ? 1 : 0)) + ((e >> 8) & 0xFFF);
    }
    else
    {
      e = t.sym8[sl];
      nx = (v >> t.bits) * (t.fc[e] & 0xFFFF) + sl - (t.fc[e] >> 16);
    }
    out[p] = (uint8_t)e;
    if (nx < kConsumePoint16)
      nx = (nx << 16) | c.next();
    x[j] = nx;
  }
}

void groups_scalar(uint32_t *x, const Table &t, Cursor &c, uint8_t *out, uint64_t steps, uint32_t S)
{
  for (uint64_t g = 0; g < steps; g++, out += S)
    group_scalar(x, t, c, out, S, S);
}

// ---------------------------------------------------------------------------------------------------------------
// AVX2
// ---------------------------------------------------------------------------------------------------------------
alignas(16) uint8_t g_compact8[256][16]; // [mask]: byte shuffle that puts word rank(l) of a load into 16-bit lane l for every set bit l, zero elsewhere
alignas(64) uint32_t g_out_perm16[16];  // AVX-512: dword order after the two packs -> output order
alignas(64) uint32_t g_out_perm16_s32[16]; // 32 states: the same (the low 8 dwords are stored)
std::once_flag g_luts_once;

void init_luts()
{
  for (uint32_t m = 0; m < 256; m++)
  {
    uint32_t r = 0;
    for (uint32_t l = 0; l < 8; l++)
    {
      const bool take = (m >> l) & 1;
      g_compact8[m][2 * l] = take ? (uint8_t)(2 * r) : 0x80;     // (0x80: pshufb writes zero)
      g_compact8[m][2 * l + 1] = take ? (uint8_t)(2 * r + 1) : 0x80;
      r += take;
    }
  }
  // packed dword d = 4q + v holds the symbols of vector v, lanes 4q..4q+3 -> output dword 4*(q&1) + (q>>1) + 2*(v&1) + 8*(v>>1)
  for (uint32_t q = 0; q < 4; q++)
    for (uint32_t v = 0; v < 4; v++)
      g_out_perm16[4 * (q & 1) + (q >> 1) + 2 * (v & 1) + 8 * (v >> 1)] = 4 * q + v;
  // 32 states: state j = 16 v + 4 q + r -> byte r | (q >> 1) << 2 | v << 3 | (q & 1) << 4: output dword (q >> 1) | v << 1 | (q & 1) << 2
  // comes from packed dword 4 q + v (lane q, dword v); output dwords 8..15 are not stored
  for (uint32_t d = 0; d < 16; d++)
    g_out_perm16_s32[d] = 0;
  for (uint32_t q = 0; q < 4; q++)
    for (uint32_t v = 0; v < 2; v++)
      g_out_perm16_s32[(q >> 1) | (v << 1) | ((q & 1) << 2)] = 4 * q + v;
}

// One group = V vectors of 8 states, written stage by stage over all V vectors (V is a template parameter: every stage unrolls, the
// V gathers of a group are in flight together).  Renormalisation without a blend: the lanes below 2^15 shift left by 16, the others
// by 0 (one variable shift), and take their words from the next 8 stream words compacted into lane order by ONE byte shuffle —
// a 16-byte pattern per 8-bit compare mask that zeroes the lanes that take nothing — then widened to dwords and OR-ed in.
template <int PACKED, uint32_t V> // PACKED = Table::mode
__attribute__((target("avx2,bmi2,popcnt"))) void groups_avx2(uint32_t *xs, const Table &t, Cursor &c, uint8_t *out, uint64_t steps)
{
  constexpr uint32_t S = 8 * V;
  __m256i x[V];
  for (uint32_t v = 0; v < V; v++)
    x[v] = _mm256_loadu_si256((const __m256i *)(xs + 8 * v));
  const __m256i vmask = _mm256_set1_epi32((int)((1u << t.bits) - 1));
  const __m128i vbits = _mm_cvtsi32_si128((int)t.bits);
  const __m256i lim_m1 = _mm256_set1_epi32((int)kConsumePoint16 - 1), c16 = _mm256_set1_epi32(16);
  const __m256i m12 = _mm256_set1_epi32(0xFFF), mff = _mm256_set1_epi32(0xFF), one = _mm256_set1_epi32(1), mffff = _mm256_set1_epi32(0xFFFF);
  const int *tab = (const int *)t.slot.data();
  uint64_t g = 0;
  // a group reads at most S words + one 16-byte load at the last position: stay that far from the end, the rest goes scalar
  while (g < steps && c.p + 2 * S + 16 <= c.end)
  {
    __m256i slot[V], e[V], nx[V], low[V];
    for (uint32_t v = 0; v < V; v++)
      slot[v] = _mm256_and_si256(x[v], vmask);
    if (PACKED == kCompact)
      for (uint32_t v = 0; v < V; v++)
        e[v] = _mm256_and_si256(_mm256_i32gather_epi32((const int *)t.sym8.data(), slot[v], 1), mff);
    else
      for (uint32_t v = 0; v < V; v++)
        e[v] = _mm256_i32gather_epi32(tab, slot[v], 4);
    for (uint32_t v = 0; v < V; v++)
    {
      const __m256i q = _mm256_srl_epi32(x[v], vbits);
      if (PACKED == kPacked || PACKED == kPackedM1)
      {
        const __m256i f = PACKED == kPacked ? _mm256_srli_epi32(e[v], 20) : _mm256_add_epi32(_mm256_srli_epi32(e[v], 20), one);
        nx[v] = _mm256_add_epi32(_mm256_mullo_epi32(q, f), _mm256_and_si256(_mm256_srli_epi32(e[v], 8), m12));
      }
      else if (PACKED == kBySlot)
      {
        const __m256i f = _mm256_add_epi32(_mm256_and_si256(_mm256_i32gather_epi32((const int *)t.slot_freq.data(), slot[v], 2), mffff), one);
        nx[v] = _mm256_add_epi32(_mm256_mullo_epi32(q, f), _mm256_srli_epi32(e[v], 8));
      }
      else
      {
        const __m256i fc = _mm256_i32gather_epi32((const int *)t.fc, e[v], 4);
        nx[v] = _mm256_add_epi32(_mm256_mullo_epi32(q, _mm256_and_si256(fc, mffff)), _mm256_sub_epi32(slot[v], _mm256_srli_epi32(fc, 16)));
      }
    }
    // nx < 2^15, UNSIGNED like every other level and the GPU (start states come from plan blobs: a state >= 2^31 must not
    // renormalise here and nowhere else): min(nx, 2^15 - 1) == nx
    uint32_t m[V];
    for (uint32_t v = 0; v < V; v++)
    {
      low[v] = _mm256_cmpeq_epi32(_mm256_min_epu32(nx[v], lim_m1), nx[v]);
      m[v] = (uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(low[v]));
    }
    const uint8_t *p = c.p;
    for (uint32_t v = 0; v < V; v++)
    {
      const __m128i words = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)p), _mm_load_si128((const __m128i *)g_compact8[m[v]]));
      p += 2 * (uint32_t)_mm_popcnt_u32(m[v]);
      x[v] = _mm256_or_si256(_mm256_sllv_epi32(nx[v], _mm256_and_si256(low[v], c16)), _mm256_cvtepu16_epi32(words));
    }
    c.p = p;
    // symbols -> output order: packus(v0,v1) etc. put lanes 0..3 of consecutive vectors side by side, which is exactly idx2idx
    // (kCompact: e is the symbol already; the other layouts carry it in the low byte)
    __m256i sym[V];
    for (uint32_t v = 0; v < V; v++)
      sym[v] = PACKED == kCompact ? e[v] : _mm256_and_si256(e[v], mff);
    _mm256_storeu_si256((__m256i *)out, _mm256_packus_epi16(_mm256_packus_epi32(sym[0], sym[1]), _mm256_packus_epi32(sym[2], sym[3])));
    if constexpr (V == 8)
      _mm256_storeu_si256((__m256i *)(out + 32), _mm256_packus_epi16(_mm256_packus_epi32(sym[4], sym[5]), _mm256_packus_epi32(sym[6], sym[7])));
    out += S;
    g++;
  }
  for (uint32_t v = 0; v < V; v++)
    _mm256_storeu_si256((__m256i *)(xs + 8 * v), x[v]);
  groups_scalar(xs, t, c, out, steps - g, S);
}

// ---------------------------------------------------------------------------------------------------------------
// AVX-512 (F + BW + DQ + VL): 64 states = 4 vectors of 16, 32 states = 2 (the reference has AVX-512 decoders for both state
// counts too: rANS32x64_16w.cpp:2108-4187, rANS32x32_16w.cpp:3032,3387)
// ---------------------------------------------------------------------------------------------------------------

// STAGED: the group written stage by stage over its V vectors (all gathers of a group in flight together) instead of vector by
// vector.  Measured on the GPU box's Zen 5 core and on this container's Xeon: the two-gather layouts gain 33 % staged on Zen 5
// (1,910 -> 2,534 MiB/s at 64 states) and nothing on the Xeon; the one-gather layouts LOSE 11 % staged on Zen 5 (4,420 -> 3,950)
// and gain 3 % on the Xeon.  So: staged for the two-gather layouts, vector by vector for the packed ones.
template <int PACKED, uint32_t V, bool STAGED>
__attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,popcnt"))) void groups_avx512(uint32_t *xs, const Table &t, Cursor &c, uint8_t *out, uint64_t steps)
{
  constexpr uint32_t S = 16 * V;
  __m512i x[V];
  for (uint32_t v = 0; v < V; v++)
    x[v] = _mm512_loadu_si512(xs + 16 * v);
  const __m512i vmask = _mm512_set1_epi32((int)((1u << t.bits) - 1));
  const __m128i vbits = _mm_cvtsi32_si128((int)t.bits);
  const __m512i lim = _mm512_set1_epi32((int)kConsumePoint16);
  const __m512i m12 = _mm512_set1_epi32(0xFFF), mff = _mm512_set1_epi32(0xFF), one = _mm512_set1_epi32(1), mffff = _mm512_set1_epi32(0xFFFF);
  const __m512i order = _mm512_load_si512(V == 4 ? g_out_perm16 : g_out_perm16_s32);
  const int *tab = (const int *)t.slot.data();
  uint64_t g = 0;
  while (g < steps && c.p + 2 * S + 32 <= c.end)
  {
    __m512i sym[V];
    if constexpr (!STAGED)
    {
      for (uint32_t v = 0; v < V; v++)
      {
        const __m512i slot = _mm512_and_si512(x[v], vmask);
        const __m512i q = _mm512_srl_epi32(x[v], vbits);
        __m512i nx, e;
        if (PACKED == kPacked || PACKED == kPackedM1)
        {
          e = _mm512_i32gather_epi32(slot, tab, 4);
          const __m512i f = PACKED == kPacked ? _mm512_srli_epi32(e, 20) : _mm512_add_epi32(_mm512_srli_epi32(e, 20), one);
          nx = _mm512_add_epi32(_mm512_mullo_epi32(q, f), _mm512_and_si512(_mm512_srli_epi32(e, 8), m12));
        }
        else if (PACKED == kBySlot)
        {
          e = _mm512_i32gather_epi32(slot, tab, 4);
          const __m512i f = _mm512_add_epi32(_mm512_and_si512(_mm512_i32gather_epi32(slot, t.slot_freq.data(), 2), mffff), one);
          nx = _mm512_add_epi32(_mm512_mullo_epi32(q, f), _mm512_srli_epi32(e, 8));
        }
        else
        {
          e = _mm512_and_si512(_mm512_i32gather_epi32(slot, t.sym8.data(), 1), mff);
          const __m512i fc = _mm512_i32gather_epi32(e, t.fc, 4);
          nx = _mm512_add_epi32(_mm512_mullo_epi32(q, _mm512_and_si512(fc, mffff)), _mm512_sub_epi32(slot, _mm512_srli_epi32(fc, 16)));
        }
        sym[v] = _mm512_and_si512(e, mff);
        const __mmask16 low = _mm512_cmplt_epu32_mask(nx, lim);
        const __m512i words = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i *)c.p));
        const __m512i mine = _mm512_maskz_expand_epi32(low, words); // lane with the k-th set bit takes word k
        x[v] = _mm512_mask_or_epi32(nx, low, _mm512_slli_epi32(nx, 16), mine);
        c.p += 2 * (uint32_t)_mm_popcnt_u32((uint32_t)low);
      }
    }
    else
    {
    __m512i slot[V], e[V], nx[V];
    __mmask16 low[V];
    for (uint32_t v = 0; v < V; v++)
      slot[v] = _mm512_and_si512(x[v], vmask);
    if (PACKED == kCompact)
      for (uint32_t v = 0; v < V; v++)
        e[v] = _mm512_and_si512(_mm512_i32gather_epi32(slot[v], t.sym8.data(), 1), mff);
    else
      for (uint32_t v = 0; v < V; v++)
        e[v] = _mm512_i32gather_epi32(slot[v], tab, 4);
    for (uint32_t v = 0; v < V; v++)
    {
      const __m512i q = _mm512_srl_epi32(x[v], vbits);
      if (PACKED == kPacked || PACKED == kPackedM1)
      {
        const __m512i f = PACKED == kPacked ? _mm512_srli_epi32(e[v], 20) : _mm512_add_epi32(_mm512_srli_epi32(e[v], 20), one);
        nx[v] = _mm512_add_epi32(_mm512_mullo_epi32(q, f), _mm512_and_si512(_mm512_srli_epi32(e[v], 8), m12));
      }
      else if (PACKED == kBySlot)
      {
        const __m512i f = _mm512_add_epi32(_mm512_and_si512(_mm512_i32gather_epi32(slot[v], t.slot_freq.data(), 2), mffff), one);
        nx[v] = _mm512_add_epi32(_mm512_mullo_epi32(q, f), _mm512_srli_epi32(e[v], 8));
      }
      else
      {
        const __m512i fc = _mm512_i32gather_epi32(e[v], t.fc, 4);
        nx[v] = _mm512_add_epi32(_mm512_mullo_epi32(q, _mm512_and_si512(fc, mffff)), _mm512_sub_epi32(slot[v], _mm512_srli_epi32(fc, 16)));
      }
      sym[v] = PACKED == kCompact ? e[v] : _mm512_and_si512(e[v], mff);
      low[v] = _mm512_cmplt_epu32_mask(nx[v], lim);
    }
    const uint8_t *p = c.p;
    for (uint32_t v = 0; v < V; v++)
    {
      const __m512i words = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i *)p));
      p += 2 * (uint32_t)_mm_popcnt_u32((uint32_t)low[v]);
      const __m512i mine = _mm512_maskz_expand_epi32(low[v], words); // lane with the k-th set bit takes word k
      x[v] = _mm512_mask_or_epi32(nx[v], low[v], _mm512_slli_epi32(nx[v], 16), mine);
    }
    c.p = p;
    }
    if constexpr (V == 4)
    {
      const __m512i packed = _mm512_packus_epi16(_mm512_packus_epi32(sym[0], sym[1]), _mm512_packus_epi32(sym[2], sym[3]));
      _mm512_storeu_si512(out, _mm512_permutexvar_epi32(order, packed));
    }
    else
    {
      // per 128-bit lane q (states 4q..4q+3 of both vectors): bytes 0..3 = vector 0, bytes 4..7 = vector 1; the rest is padding
      const __m512i pk = _mm512_packus_epi32(sym[0], sym[V - 1]);
      const __m512i packed = _mm512_packus_epi16(pk, pk);
      _mm256_storeu_si256((__m256i *)out, _mm512_castsi512_si256(_mm512_permutexvar_epi32(order, packed)));
    }
    out += S;
    g++;
  }
  for (uint32_t v = 0; v < V; v++)
    _mm512_storeu_si512(xs + 16 * v, x[v]);
  groups_scalar(xs, t, c, out, steps - g, S);
}

int detect_level()
{
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl"))
    return kLevelAvx512;
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt"))
    return kLevelAvx2;
  return kLevelScalar;
}

} // namespace

int best_level()
{
  static const int level = detect_level();
  return level;
}

const char *level_name(int level) { return level == kLevelAvx512 ? "avx512" : level == kLevelAvx2 ? "avx2" : "scalar"; }

// `steps` whole groups from `x` (S states), advancing the cursor; level = the instruction set to use (clamped to the host's)
static void decode_groups(int level, uint32_t *x, const Table &t, Cursor &c, uint8_t *out, uint64_t steps, uint32_t S)
{
  std::call_once(g_luts_once, init_luts);
  if (level > best_level())
    level = best_level();
  const int mode = t.mode;
#define HSRANS_CPU_BY_MODE(fn, ...) \
  (mode == kPacked ? fn<kPacked __VA_ARGS__> : mode == kPackedM1 ? fn<kPackedM1 __VA_ARGS__> : mode == kBySlot ? fn<kBySlot __VA_ARGS__> : fn<kCompact __VA_ARGS__>)
  // 32-state chains on AMD cores: the 256-bit loop beats the 512-bit one at every width (Zen 5: 2,655 against 2,535 MiB/s at 11
  // bits, 1,822 against 1,662 at 14; on this container's Xeon it is the other way round: 1,620 against 1,820) — the reference's
  // dispatcher keeps AVX-512 away from Zen as well (block_rANS32x64_16w_decode.cpp:135)
  if (level == kLevelAvx512 && S == 32 && cpu_is_amd())
    level = kLevelAvx2;
  const bool packed = mode == kPacked || mode == kPackedM1;
  if (level == kLevelAvx512 && S == 64 && packed)
    HSRANS_CPU_BY_MODE(groups_avx512, , 4, false)(x, t, c, out, steps);
  else if (level == kLevelAvx512 && S == 64)
    HSRANS_CPU_BY_MODE(groups_avx512, , 4, true)(x, t, c, out, steps);
  else if (level == kLevelAvx512 && S == 32 && packed)
    HSRANS_CPU_BY_MODE(groups_avx512, , 2, false)(x, t, c, out, steps);
  else if (level == kLevelAvx512 && S == 32)
    HSRANS_CPU_BY_MODE(groups_avx512, , 2, true)(x, t, c, out, steps);
  else if (level >= kLevelAvx2 && S == 64)
    HSRANS_CPU_BY_MODE(groups_avx2, , 8)(x, t, c, out, steps);
  else if (level >= kLevelAvx2)
    HSRANS_CPU_BY_MODE(groups_avx2, , 4)(x, t, c, out, steps);
#undef HSRANS_CPU_BY_MODE
  else
    groups_scalar(x, t, c, out, steps, S);
}

// ---------------------------------------------------------------------------------------------------------------
// plan execution: the chains of a decode plan (hsrans_plan.h), on the host.  Chains are independent, so they are spread
// over `threads` std::threads (the reference's thread-pool fan-out, mt_rANS32x64_16w_decode.cpp:217-220).
// ---------------------------------------------------------------------------------------------------------------
namespace
{
struct CheckpointSink
{
  const uint64_t *groups = nullptr; // ascending absolute group indices to record at (index build), or null
  size_t n = 0;
  uint32_t *states = nullptr; // [n * S]
  uint64_t *words = nullptr;  // [n] absolute stream offset of the read cursor
  // index builds want the states, not the bytes: the decoded symbols then go to a 32 KiB scratch that is written over and over
  // (`out` may be null), so the pass needs no buffer of the stream's CLAIMED decoded length — a header is untrusted input
  bool discard = false;
};
constexpr uint64_t kDiscardGroups = 512;

bool run_chain(int level, const PlanHeader &h, const uint32_t *cf, const Piece *pc, const uint32_t *st, uint32_t chain, const uint8_t *stream, uint64_t stream_len,
               uint8_t *out, const CheckpointSink *sink)
{
  const uint32_t S = h.states;
  uint32_t x[64] = {};
  Table t;
  uint64_t have_hist = ~(uint64_t)0;
  const bool discard = sink != nullptr && sink->discard;
  alignas(64) uint8_t scratch[kDiscardGroups * 64 + 64];
  for (uint32_t pi = cf[chain]; pi < cf[chain + 1]; pi++)
  {
    const Piece &p = pc[pi];
    if (p.flags & kPieceChainStart)
      memcpy(x, st + (size_t)p.state_idx * S, 4 * (size_t)S);
    if (p.flags & kPieceFill)
    {
      if (!discard)
        memset(out + p.out_off, (int)(p.hist_off & 0xFF), (size_t)p.fill_len);
      continue;
    }
    if (p.hist_off != have_hist)
    {
      if (!t.build(stream + p.hist_off, h.bits, S))
        return false;
      have_hist = p.hist_off;
    }
    Cursor c{stream + p.words_off, stream + stream_len};
    uint8_t *o = discard ? scratch : out + p.out_off;
    uint64_t steps = p.steps;
    if (sink != nullptr && sink->n != 0)
    {
      uint64_t g_abs = p.out_off / S;
      size_t k = (size_t)(std::upper_bound(sink->groups, sink->groups + sink->n, g_abs) - sink->groups);
      while (steps > 0)
      {
        const uint64_t next = k < sink->n ? sink->groups[k] : ~(uint64_t)0;
        uint64_t n = std::min<uint64_t>(steps, next - g_abs);
        if (discard)
        {
          n = std::min<uint64_t>(n, kDiscardGroups); // the scratch holds this many groups
          decode_groups(level, x, t, c, scratch, n, S);
          if (c.p > c.end) // the stream ran out long before its header's claim: no point in "decoding" zeros for hours
            return false;
          steps -= n;
          g_abs += n;
          if (steps > 0 && g_abs == next)
          {
            memcpy(sink->states + k * S, x, 4 * (size_t)S);
            sink->words[k] = (uint64_t)(c.p - stream);
            k++;
          }
          continue;
        }
        decode_groups(level, x, t, c, o, n, S);
        o += n * S;
        steps -= n;
        g_abs += n;
        if (steps > 0)
        {
          memcpy(sink->states + k * S, x, 4 * (size_t)S);
          sink->words[k] = (uint64_t)(c.p - stream);
          k++;
        }
      }
    }
    else
    {
      decode_groups(level, x, t, c, o, steps, S);
      o += steps * S;
    }
    if (p.tail)
      group_scalar(x, t, c, discard ? scratch : o, S, p.tail);
  }
  return true;
}

// block_ container without an index: follow the inline headers (block_rANS32x64_16w_decode.cpp:47-123) — the host twin of
// run_block_walk in hsrans_kernels.hip, same checks, same order
bool run_block_walk(int level, const PlanHeader &h, const uint32_t *st, const uint8_t *stream, uint64_t stream_len, uint8_t *out, uint64_t out_cap)
{
  const uint32_t S = h.states;
  const uint64_t out_len = h.decoded_len;
  const uint64_t whole = out_len - S + 1;
  uint32_t x[64] = {};
  memcpy(x, st, 4 * (size_t)S);
  uint64_t pos = h.aux_off, i = 0;
  Table t;
  bool have_table = false;
  do
  {
    if (pos + 8 > stream_len)
      return false;
    const uint64_t hdr = rd64(stream + pos);
    pos += 8;
    if (hdr >> 63)
    {
      const uint64_t len = hdr & (((uint64_t)1 << 54) - 1);
      if (len == 0 || len > out_cap - i)
        return false;
      memset(out + i, (int)((hdr >> 54) & 0xFF), (size_t)len);
      i += len;
    }
    else
    {
      if (hdr == 0 || pos + 512 > stream_len || !t.build(stream + pos, h.bits, h.states))
        return false;
      have_table = true;
      pos += 512;
      uint64_t end = i + hdr;
      if (end > whole)
        end = whole;
      else if (end & (S - 1))
        return false;
      const uint64_t steps = end > i ? (end - i + S - 1) / S : 0;
      Cursor c{stream + pos, stream + stream_len};
      decode_groups(level, x, t, c, out + i, steps, S);
      i += steps * S;
      pos = (uint64_t)(c.p - stream);
    }
    if (i > whole)
    {
      if (i >= out_len)
        return true;
      break;
    }
  } while (i < whole);
  if (i < out_len)
  {
    if (!have_table)
      return false;
    Cursor c{stream + pos, stream + stream_len};
    group_scalar(x, t, c, out + i, S, (uint32_t)(out_len - i));
  }
  return true;
}

size_t exec_plan_impl(int level, uint32_t threads, const uint8_t *plan, size_t plan_size, const uint8_t *stream, size_t stream_len, uint8_t *out, size_t out_cap,
                      const CheckpointSink *sink)
{
  if (!plan_validate(plan, plan_size, stream_len, out_cap))
    return 0;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  const uint32_t *st = (const uint32_t *)(plan + plan_states_off(h.n_chains, h.n_pieces));
  if (h.flags & kPlanWalk)
    return run_block_walk(level, h, st, stream, stream_len, out, out_cap) ? (size_t)h.decoded_len : 0;
  if (threads <= 1 || h.n_chains == 1)
  {
    for (uint32_t ch = 0; ch < h.n_chains; ch++)
      if (!run_chain(level, h, cf, pc, st, ch, stream, stream_len, out, sink))
        return 0;
    return (size_t)h.decoded_len;
  }
  std::atomic<uint32_t> next{0};
  std::atomic<bool> good{true};
  auto worker = [&]() {
    // chains in batches (neighbouring chains share cache lines of the output and, in mt_ plans with checkpoints, a table)
    const uint32_t batch = std::max<uint32_t>(1, std::min<uint32_t>(64, h.n_chains / (8 * threads)));
    while (good.load(std::memory_order_relaxed))
    {
      const uint32_t first = next.fetch_add(batch, std::memory_order_relaxed);
      if (first >= h.n_chains)
        break;
      for (uint32_t ch = first; ch < std::min(h.n_chains, first + batch); ch++)
        if (!run_chain(level, h, cf, pc, st, ch, stream, stream_len, out, sink))
          good = false;
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t k = 1; k < threads; k++)
    pool.emplace_back(worker);
  worker();
  for (auto &th : pool)
    th.join();
  return good ? (size_t)h.decoded_len : 0;
}
} // namespace

size_t exec_plan(int level, uint32_t threads, const uint8_t *plan, size_t plan_size, const uint8_t *stream, size_t stream_len, uint8_t *out, size_t out_cap)
{
  return exec_plan_impl(level, threads, plan, plan_size, stream, stream_len, out, out_cap, nullptr);
}

size_t decode(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap)
{
  if (in == nullptr || out == nullptr || in_len < 16 || !valid_codec(container, states, bits))
    return 0;
  const uint64_t out_len = rd64(in);
  if (out_len > out_cap)
    return 0;
  std::vector<uint8_t> plan;
  if (!plan_build_vec(container, states, bits, in, in_len, out_cap, &plan))
    return 0;
  return exec_plan(level, threads, plan.data(), plan.size(), in, in_len, out, out_cap);
}

// Index of an existing raw / mt_ stream with checkpoints at the given groups: one sequential host decode that records
// {states, cursor} there — what hsrans_index_build_at does with one wavefront on the GPU
size_t index_build(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, const uint64_t *groups, size_t n_groups,
                   uint8_t *plan_out, size_t plan_cap, uint32_t uniform_interval)
{
  if (in == nullptr || plan_out == nullptr || in_len < 16 || !valid_codec(container, states, bits) || container == HSRANS_BLOCK || groups == nullptr || n_groups == 0)
    return 0;
  for (size_t k = 0; k < n_groups; k++)
    if (groups[k] == 0 || (groups[k] % 4) != 0 || (k > 0 && groups[k] <= groups[k - 1]))
      return 0;
  const uint64_t out_len = rd64(in);
  const uint32_t S = (uint32_t)states;
  // The decoded length in the header is untrusted: nothing here is sized by it alone.  The base plan (one chain per block) is
  // sized by the chains the stream really holds (a block header is >= 8 bytes, a chain of the plan ~308: at most ~40x the
  // stream); the decode pass keeps no output (sink.discard).
  std::vector<uint8_t> base;
  if (!plan_build_vec(container, states, bits, in, in_len, (size_t)out_len, &base))
    return 0;
  const size_t base_size = base.size();
  PlanHeader h;
  memcpy(&h, base.data(), sizeof(h));
  if (h.n_pieces != h.n_chains)
    return 0;
  std::vector<uint32_t> ck_states(n_groups * S);
  std::vector<uint64_t> ck_words(n_groups, 0);
  CheckpointSink sink;
  sink.groups = groups;
  sink.n = n_groups;
  sink.states = ck_states.data();
  sink.words = ck_words.data();
  sink.discard = true;
  if (exec_plan_impl(level, threads, base.data(), base_size, in, in_len, nullptr, (size_t)out_len, &sink) == 0)
    return 0;
  const uint32_t *cf0 = (const uint32_t *)(base.data() + plan_chain_first_off());
  const Piece *pc0 = (const Piece *)(base.data() + plan_pieces_off(h.n_chains));
  const uint32_t *st0 = (const uint32_t *)(base.data() + plan_states_off(h.n_chains, h.n_pieces));
  PlanBuilder pb;
  pb.begin(container, states, bits, out_len, in_len);
  pb.hdr.interval = uniform_interval; // != 0: the caller's groups are the multiples of it (the plan then equals the encoder's for that interval)
  if (container == HSRANS_RAW)
  {
    uint16_t counts[256];
    memcpy(counts, in + pc0[0].hist_off, 512);
    pb.set_hist(counts);
  }
  size_t k = 0;
  for (uint32_t ch = 0; ch < h.n_chains; ch++)
  {
    const Piece &bp = pc0[cf0[ch]];
    if (bp.flags & kPieceFill)
    {
      pb.add_chain(bp, nullptr);
      continue;
    }
    const uint64_t T = bp.steps, g0 = bp.out_off / S;
    while (k < n_groups && groups[k] <= g0)
      k++;
    uint64_t g = 0;
    const uint32_t *st = st0 + (size_t)bp.state_idx * S;
    uint64_t words = bp.words_off;
    while (true)
    {
      const bool more = k < n_groups && groups[k] < g0 + T;
      const uint64_t g_next = more ? groups[k] - g0 : T;
      Piece p{};
      p.hist_off = bp.hist_off;
      p.out_off = bp.out_off + g * S;
      p.words_off = words;
      p.steps = (uint32_t)(g_next - g);
      p.tail = (uint16_t)(more ? 0 : bp.tail);
      pb.add_chain(p, st);
      if (!more)
        break;
      st = &ck_states[k * S];
      words = ck_words[k];
      g = g_next;
      k++;
    }
  }
  return pb.serialize(plan_out, plan_cap);
}

} // namespace cpu
} // namespace hsrans
