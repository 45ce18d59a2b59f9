// gfx950 (MI355X, CDNA4) kernels for the hypersonic-rANS 32-bit-state / 16-bit-word decode path.
//
// What is computed (reference: /root/reference/src, scalar form rANS32x64_16w.cpp:223-250 = block_codec64.h:182-213):
//   for every group of S symbols, for state j = 0..S-1 in order:
//       slot = x_j & (2^bits - 1);  sym = cumulInv[slot];  out[i + idx2idx[j]] = sym;
//       x_j  = (x_j >> bits) * freq[sym] + slot - cumul[sym];
//       if (x_j < 2^15) x_j = (x_j << 16) | *readHead++;
//
// How it is mapped to a wave64 (one wavefront = one chain of the decode plan, hsrans_plan.h):
//   * lane j owns state j.  All S table steps happen at once; the only cross-lane dependency is the read cursor:
//     `readHead++` in ascending j  ==  lane j reads word[cur + popcount(renorm_mask & lanes_below_j)], i.e.
//     v_cmp -> SGPR-pair mask, v_mbcnt_lo/hi for the lane's rank, s_bcnt1 to advance the (scalar) cursor.
//   * decode table lives in LDS.  bits <= 12: one uint32 per slot = sym | (freq-1) << 8 | (slot-cumul) << 20
//     (freq-1 so that freq == 4096 fits — the reference's own packed table cannot, hist.cpp:304).
//     bits >= 13: uint8 sym[2^bits] + uint32 {freq | cumul << 16}[256] (two dependent LDS reads, as the reference's
//     hist_dec2_t path, hist.h:42-47).  The table is built in-kernel from the 256 uint16 counts in the stream.
//   * the uint16 stream is staged through a 2 KiB LDS ring per wave, refilled 1 KiB at a time by one coalesced,
//     bounds-checked buffer_load_dwordx4 per lane that is issued ~25 groups before it is needed.
//   * idx2idx maps 4 consecutive lanes to 4 consecutive output bytes (it is the bit permutation
//     j -> (j&0x23)|((j&4)<<2)|((j&0x18)>>1)), so a quad assembles one output dword with two DPP quad_perm ORs;
//     four groups are accumulated so that every lane stores one dword and the wave writes 4*S contiguous bytes.
//   * no MFMA: this is integer gather work; the roofline that bounds it is HBM (compressed bytes in + decoded bytes out).
//
// ONE device translation unit, in parts by kernel family (included below in dependency order):
//   kernels_common.h    ring, table build, group step + hand-scheduled groups, output path, generic chain runner
//   kernels_persist.h   k_decode_persist   uniform-interval raw plans (static runs + ticket queues)
//   kernels_direct.h    k_decode_direct    one chain (run of chains) per wave: the headline; k_calibrate
//   kernels_batch.h     k_decode_batch     K independent streams in one launch of the one-chain-per-wave form
//   kernels_grouped.h   k_decode_grouped   block_/mt_ plans with checkpoints (BASELINE config 4)
//   kernels_spread.h    k_decode_spread    the same plans with few, large blocks: the chains dealt out evenly, two tables per workgroup
//   kernels_dealt.h     k_decode_dealt     the same in one round with HOST-dealt shares (<= 2 blocks each), two dependent trips before the first group
//   kernels_generic.h   k_decode           mt_ without index, block_ header walk, index-build passes
//   kernels_dual.h      k_decode_dual      two chains per wave (13-15 bits)
//   kernels_single.h    k_decode_single    one dependent chain (raw stream without index)
//   kernels_walk.h      k_mt_chase / k_mt_fill   K2: the mt_ header chain on the device
// This file: the host side — tuning constants, launch shapes, hsrans_index_boundaries' chain lengths, launch_decode.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "hsrans_kernels.h"
#include "hsrans_plan.h"

#include "kernels_common.h"
#include "kernels_persist.h"
#include "kernels_direct.h"
#include "kernels_grouped.h"
#include "kernels_spread.h"
#include "kernels_dealt.h"
#include "kernels_generic.h"
#include "kernels_dual.h"
#include "kernels_batch.h"
#include "kernels_single.h"
#include "kernels_walk.h"

namespace hsrans
{

hipError_t launch_mt_chase(const uint8_t *d_stream, uint64_t stream_len, uint64_t out_cap, uint32_t S, uint64_t *d_blocks, uint32_t max_blocks, WalkResult *d_result,
                           hipStream_t stream)
{
  (void)hipGetLastError(); // (the runtime's last error is sticky per thread: an earlier failed call must not be reported as this launch's)
  hipLaunchKernelGGL(k_mt_chase, dim3(1), dim3(64), 0, stream, d_stream, stream_len, out_cap, S, d_blocks, max_blocks, d_result);
  return hipGetLastError();
}

hipError_t launch_index_assemble(const IndexArgs &a, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_index_count, dim3(1), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(k_index_fill, dim3(a.n_base), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_stream_checksum(const uint8_t *d_stream, uint64_t stream_len, uint64_t *d_sum, hipStream_t stream)
{
  hipError_t e = hipMemsetAsync(d_sum, 0, 8, stream);
  if (e != hipSuccess)
    return e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_stream_checksum, dim3(kChecksumGrid), dim3(kChecksumBlock), 0, stream, d_stream, stream_len, (unsigned long long *)d_sum);
  return hipGetLastError();
}

hipError_t launch_mt_fill(const uint8_t *d_stream, uint64_t stream_len, uint32_t S, uint32_t bits, const uint64_t *d_blocks, uint8_t *d_plan, uint32_t n_chains,
                          uint64_t out_len, WalkResult *d_result, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_mt_fill, dim3(n_chains), dim3(64), 0, stream, d_stream, stream_len, S, bits, d_blocks, d_plan, n_chains, out_len, d_result);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------------------------
// tuning knobs (environment, read once per process; the defaults are the measured best)
static constexpr uint32_t g_pack64_max_bits = 14; // widest histogram decoded with the 8-byte-per-slot shared table (knob removed in round 5: settled)
static uint32_t g_waves_per_wg = 16;    // HSRANS_WAVES_PER_WG: waves per workgroup of the shared-table launches
static uint32_t g_spread = 1;           // HSRANS_SPREAD=0: grouped plans with few large blocks keep the one-block-per-workgroup launch
static constexpr uint32_t g_static_percent = 100; // share of the chains handed out statically (uniform persistent launches; knob removed: settled)
// HSRANS_SLOT_WEIGHTS: per-mille run length of the 8 wave classes, see PersistentArgs::run_len.  Measured on
// MI355X at 8 waves per SIMD (bits <= 12): with equal runs the four age classes of a workgroup finish at 33/36/39/42 us,
// with these weights all at 39 us (tools/stamps.py), 3-7 % less kernel time; at 4 waves per SIMD (bits >= 13) equal
// runs are better and are kept.
static uint32_t g_slot_weights[8] = {1328, 1268, 1211, 1145, 1018, 875, 665, 490}; // (re-fitted after the wait fix; runs are whole chains, so this fit is coarse)
// HSRANS_SLOT_WEIGHTS4: the same for launches with one 16-wave workgroup per CU (4 waves per SIMD: 13-bit tables)
static uint32_t g_slot_weights4[8] = {1150, 1050, 950, 850, 1150, 1050, 950, 850};
// HSRANS_DIRECT_WEIGHTS / HSRANS_DIRECT_WEIGHTS4: the chain lengths of the one-chain-per-wave index (hsrans_index_boundaries),
// per mille of the mean, by the same 8 classes.  Here nothing evens out a wrong weight afterwards (the queues above only hold the
// short tail chains), so these are fitted until all classes finish together (tools/tune_weights.py: the spread of the classes'
// mean finish times goes from 19.8 us with the weights above to 0.1 us): the waves of a CU's first workgroup run ahead of
// the second one's on every SIMD, and inside a workgroup the older waves a little ahead of the younger.
// One set per occupancy (waves per SIMD): 8 = two 16-wave workgroups per CU (bits <= 12), 6 = two 12-wave workgroups (15 bits,
// rank table), 4 = one 16-wave workgroup (13 bits), 3 = one 12-wave workgroup (14 bits).
// (Re-fitted after the decode loops stopped draining the memory queue every iteration: the oldest class now runs three times as
// many groups as the youngest in the same time.  The same fit with the buffers rotated through HBM lands within 2 % of these.)
static uint32_t g_direct_weights[8] = {1396, 1332, 1244, 1131, 960, 809, 643, 484}; // (round 3, after the loop's scalar bookkeeping was trimmed: between the fits of two boxes; hsrans_ctx_calibrate fits them to the device at hand)
static uint32_t g_direct_weights6[8] = {1192, 1159, 1120, 1072, 976, 907, 829, 745};
static uint32_t g_direct_weights4[8] = {1097, 1053, 977, 873, 1098, 1053, 977, 873};
static uint32_t g_direct_weights3[8] = {1052, 1025, 986, 936, 1052, 1025, 986, 936};
// 32-state plans (two chains per wave, one per half: run_direct_pair, hand-scheduled pair loop; HSRANS_DIRECT_WEIGHTS_PAIR): with 7
// scalar instructions per group the CU's scalar unit is contended and the oldest waves get nearly all of it
static uint32_t g_direct_weights_pair[8] = {1662, 1550, 1365, 1142, 887, 656, 450, 289}; // (re-fitted twice in round 3 as the pair loop lost scalar instructions: 1847 ... 204 before)
// HSRANS_PRIVATE_PAIR: 0 = never, 1 = when there are more chains than wave slots (default), 2 = always pair the
// chains of 32-state plans in private-table launches.  Measured: 2^30 B in 16,384 blocks 1.40 -> 1.33 ms, but 100 MB in 1,526
// blocks 0.25 -> 0.30 ms (everything is latency-bound there and half as many waves are in flight)
static uint32_t g_private_pair = 1;
static constexpr bool g_weights_two_level = false; // (the class weights are not applied to the two-level table mode: measured, knob removed)
// HSRANS_TABLE_SPILL=1 (host-built tables stay in global memory: kModeSpill, BASELINE config 3's comparison side) is read whenever a table is
// chosen — at device-plan creation — so that one process can time both sides on the same buffers (bench.py's config-3 leg)
static bool table_spill_now() { const char *e = getenv("HSRANS_TABLE_SPILL"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }
// one-chain-per-wave plans (hsrans_index_boundaries): share of the stream (per mille) left to short chains that the ticket
// queues hand to waves that are done early, and the length of those chains in groups

static void read_tuning_once();
static constexpr uint32_t g_persist_kernel = 1; // uniform-interval plans run k_decode_persist (the A/B against k_decode<3, true> is settled: knob removed)
static uint32_t g_single_fast = 1; // HSRANS_SINGLE_FAST: 0 = un-indexed raw streams on the general kernel (one wave, two LDS round trips per group)
static constexpr bool g_rank_table = true; // wide histograms use the rank table (the comparison knob is gone: 52 against 57-63 us, CHANGELOG round 2-3)
static constexpr uint32_t g_dual_waves = 16; // waves per workgroup of k_decode_dual (12 — two workgroups per CU beside a 16 KiB table — measured slower again in round 5: profiles/r05_table_at_zero_ab.txt; knob removed)
static uint32_t g_dual = 1; // HSRANS_DUAL: 0 = never run two chains per wave (k_decode_dual), 1 = where it pays (default), 2 = for every width (experiment)
// the one-chain-per-wave weights of the dual kernel's launches (one 16-wave workgroup per CU, two chains per wave)
static uint32_t g_dual_weights[8] = {1232, 1112, 934, 722, 1232, 1112, 934, 722};        // 13 bits (8-byte table)
static uint32_t g_dual_weights_wide[8] = {1160, 1077, 955, 810, 1160, 1077, 955, 810}; // 14 / 15 bits (rank table; fitted with 4 pairs rotated: spread of the classes' finish 5.4 -> 0.2 us)

typedef void (*KernelFn)(KParams);
static KernelFn kernel_for(int mode, bool shared)
{
  switch (mode * 2 + (shared ? 1 : 0))
  {
  case 0: return k_decode<kModePack, false>;
  case 1: return k_decode<kModePack, true>;
  case 2: return k_decode<kModePackM1, false>;
  case 3: return k_decode<kModePackM1, true>;
  case 4: return k_decode<kModeTwoLevel, false>;
  case 5: return k_decode<kModeTwoLevel, true>;
  case 8: case 9: return k_decode<kModeRank, true>;
  case 10: case 11: return k_decode<kModeSpill, true>;
  default: return k_decode<kModePack64, true>;
  }
}

uint32_t pack64_max_bits() { return g_pack64_max_bits; }
bool table_spill() { return table_spill_now(); }

// Which host-built decode table a plan that carries its histogram gets, and whether its launch runs two chains per wave.
// `direct`: the plan has one chain per wave (PlanHeader::interval == 0), which is what the dual kernel is written for.
TableChoice choose_table(uint32_t bits, uint32_t states, bool direct)
{
  read_tuning_once();
  TableChoice t{0, false};
  if (table_spill_now())
    t.mode = kModeSpill;
  else if (direct && g_dual == 2 && states == 64 && bits <= 12) // experiment: the dual kernel below 13 bits too
  {
    t.mode = kModePack64;
    t.dual = true;
  }
  else if (direct && g_dual && states == 64 && bits >= 13)
  {
    // one workgroup of 16 waves per CU, two chains per wave: 13 bits keeps the 8-byte-per-slot table (64 KiB + 32 rings = 144 KiB);
    // at 14 / 15 bits that table does not fit beside 32 rings, so the rank table (18 / 34 KiB).  Measured (100 MB, last wave
    // done, fitted weights, hand-scheduled loops): 13 bits 42.7 us against 49.7 us one chain per wave; 14 / 15 bits 52.3 / 52.7 us
    // (round 2's coarse + fine tables: 57.8 / 57.7; one chain per wave beside the 128 KiB one-lookup table at 14 bits: 63.3)
    t.mode = bits == 13 ? kModePack64 : kModeRank;
    t.dual = true;
  }
  else if (bits <= g_pack64_max_bits && !(bits == 14 && g_rank_table))
    t.mode = kModePack64;
  else if (bits >= 13 && g_rank_table)
    t.mode = kModeRank; // (14 bits too: beside the 128 KiB one-lookup table only 12 waves fit a CU — 0.31 against 0.37 with a checkpoint every 32 groups)
  return t;
}

size_t rank_table_entries(uint32_t bits) { return table_bytes_for(kModeRank, bits) / 8; }

size_t build_rank_table(const uint16_t counts[256], uint32_t bits, uint2 *out, size_t capacity_entries)
{
  if (bits < 13 || bits > 15 || capacity_entries < rank_table_entries(bits))
    return 0;
  // the 32 most frequent symbols — nearly every lane of a group — get 32 different bank pairs of the entry table (indexed by
  // symbol value, symbols 32 apart would share banks: 8.0 instead of 7.3 bank-conflict cycles per group)
  uint8_t *rank_of_slot = (uint8_t *)out;
  uint2 *ent = (uint2 *)(rank_of_slot + (1u << bits));
  uint32_t order[256];
  for (uint32_t s = 0; s < 256; s++)
    order[s] = s;
  std::stable_sort(order, order + 256, [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
  uint8_t rank[256];
  for (uint32_t r = 0; r < 256; r++)
    rank[order[r]] = (uint8_t)r;
  uint32_t cumul = 0;
  for (uint32_t s = 0; s < 256; s++)
  {
    ent[rank[s]] = make_uint2((uint32_t)counts[s] | (s << 24), 0u - cumul);
    for (uint32_t k = 0; k < counts[s] && cumul + k < (1u << bits); k++)
      rank_of_slot[cumul + k] = rank[s];
    cumul += counts[s];
  }
  return cumul == (1u << bits) ? rank_table_entries(bits) : 0;
}

static void read_tuning_impl();
static void read_tuning_once() // contexts may be created from several threads
{
  static std::once_flag once;
  std::call_once(once, read_tuning_impl);
}
static void read_tuning_impl()
{
  if (const char *e = getenv("HSRANS_SPREAD"))
    g_spread = atoi(e) != 0;
  if (const char *e = getenv("HSRANS_WAVES_PER_WG"))
    if (atoi(e) >= 4 && atoi(e) <= 16)
      g_waves_per_wg = (uint32_t)atoi(e);
  auto read_weights = [](const char *name, uint32_t *w) { // 8 comma-separated per-mille values, rescaled to mean 1000
    const char *e = getenv(name);
    if (e == nullptr)
      return;
    uint32_t v[8], n = 0;
    uint64_t sum = 0;
    for (const char *p = e; n < 8 && *p; n++)
    {
      v[n] = (uint32_t)strtoul(p, (char **)&p, 10);
      sum += v[n];
      if (*p == ',')
        p++;
    }
    if (n == 8 && sum > 0)
      for (uint32_t k = 0; k < 8; k++)
        w[k] = (uint32_t)((uint64_t)v[k] * 8000 / sum);
  };
  read_weights("HSRANS_SLOT_WEIGHTS", g_slot_weights);
  read_weights("HSRANS_SLOT_WEIGHTS4", g_slot_weights4);
  read_weights("HSRANS_DIRECT_WEIGHTS", g_direct_weights);
  read_weights("HSRANS_DIRECT_WEIGHTS4", g_direct_weights4);
  read_weights("HSRANS_DIRECT_WEIGHTS6", g_direct_weights6);
  read_weights("HSRANS_DIRECT_WEIGHTS3", g_direct_weights3);
  read_weights("HSRANS_DIRECT_WEIGHTS_PAIR", g_direct_weights_pair);
  if (const char *e = getenv("HSRANS_PRIVATE_PAIR"))
    g_private_pair = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_DUAL"))
    g_dual = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_SINGLE_FAST"))
    g_single_fast = (uint32_t)atoi(e);
  read_weights("HSRANS_DUAL_WEIGHTS", g_dual_weights);               // 13 bits (8-byte table)
  read_weights("HSRANS_DUAL_WEIGHTS_WIDE", g_dual_weights_wide); // 14 / 15 bits (rank table)
}

// per device (the CURRENT device): dynamic-LDS limit of every kernel variant, CU count
hipError_t prepare_kernels(DeviceGeom *geom)
{
  read_tuning_once();
  geom->max_lds = 160 * 1024;
  geom->num_cus = 256;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
    geom->num_cus = (uint32_t)cus;
  for (int mode = 0; mode < 6; mode++)
    for (int shared = 0; shared < 2; shared++)
    {
      const hipError_t e = hipFuncSetAttribute((const void *)kernel_for(mode, shared != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
      if (e != hipSuccess)
        return e;
    }
  for (KernelFn fn : {(KernelFn)k_decode_single, (KernelFn)k_decode_persist<kModePack64>, (KernelFn)k_decode_persist<kModeRank>, (KernelFn)k_calibrate, (KernelFn)k_decode_dual<kModePack64>, (KernelFn)k_decode_dual<kModeRank>, (KernelFn)k_decode_direct<kModePack>, (KernelFn)k_decode_direct<kModePackM1>,
                      (KernelFn)k_decode_direct<kModeTwoLevel>, (KernelFn)k_decode_direct<kModePack64>, (KernelFn)k_decode_direct<kModeRank>, (KernelFn)k_decode_direct<kModeSpill>,
                      (KernelFn)k_decode_grouped<kModePack, false>, (KernelFn)k_decode_grouped<kModePackM1, false>, (KernelFn)k_decode_grouped<kModeTwoLevel, false>,
                      (KernelFn)k_decode_grouped<kModePack64, false>, (KernelFn)k_decode_grouped<kModeTwoLevel, true>, (KernelFn)k_decode_grouped<kModePack64, true>,
                      (KernelFn)k_decode_grouped<kModeRank, false>, (KernelFn)k_decode_grouped<kModeRank, true>, (KernelFn)k_decode_spread<kModePack64>,
                      (KernelFn)k_decode_grouped<kModePack64, true, true>, (KernelFn)k_decode_grouped<kModeRank, true, true>, (KernelFn)k_decode_spread<kModePack64, true>})
  {
    const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e != hipSuccess)
      return e;
  }
  for (const void *fn : {(const void *)k_decode_dealt<true, false>, (const void *)k_decode_dealt<false, false>, (const void *)k_decode_dealt<true, true>,
                         (const void *)k_decode_dealt_rank<13, false>, (const void *)k_decode_dealt_rank<13, true>, (const void *)k_decode_dealt_rank<14, false>,
                         (const void *)k_decode_dealt_rank<14, true>})
  {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e != hipSuccess)
      return e;
  }
  {
    hipError_t e = hipFuncSetAttribute((const void *)k_decode_batch<kModePack64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_decode_grouped_batch<kModePack64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_decode_batch_pair<kModePack64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_decode_batch_dual<kModePack64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_decode_batch_dual<kModeRank>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_calibrate_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

DeviceGeom default_geom()
{
  read_tuning_once();
  DeviceGeom g{};
  g.max_lds = 160 * 1024;
  g.num_cus = 256; // MI355X
  return g;
}

// k_decode_spread's wave weights: the age-class weights of the one-chain-per-wave launch (the device's own once calibrated), as
// the cumulative table run_grouped's shares use: cum[half of the grid][wave]
constexpr uint32_t kSpreadWaves = 16;
static void spread_weights(const DeviceGeom &dg, uint16_t (*cum_out)[17])
{
  read_tuning_once();
  const uint32_t *w8 = dg.have_direct_weights ? dg.direct_weights : g_direct_weights;
  for (uint32_t hf = 0; hf < 2; hf++)
  {
    uint32_t cum = 0;
    for (uint32_t k = 0; k <= 16; k++)
    {
      cum_out[hf][k] = (uint16_t)cum;
      cum += k < kSpreadWaves ? w8[hf * 4 + k / (kSpreadWaves / 4)] / 10 : 0;
    }
  }
}

// (`cum`: the weights the launch will hand the kernel — computed ONCE per launch, so that the check and the kernel cannot see two
// different sets when hsrans_ctx_calibrate rewrites the context's geometry on another thread)
static uint32_t spread_longest_share_of(const DeviceGeom &dg, uint64_t n_chains, const uint16_t (*cum)[17])
{
  const uint32_t grid = spread_grid(dg);
  if (n_chains < (uint64_t)grid * kSpreadWaves || n_chains > (uint64_t)grid * kSpreadMaxShare) // (a chain per wave at least)
    return 0;
  uint32_t longest = 0;
  for (uint32_t b : {0u, (grid + 1) / 2 - 1, (grid + 1) / 2, grid - 1}) // (shares differ by rounding only within a half)
  {
    const uint32_t len = spread_share_begin((uint32_t)n_chains, b + 1, grid, cum[0][kSpreadWaves], cum[1][kSpreadWaves]) -
                         spread_share_begin((uint32_t)n_chains, b, grid, cum[0][kSpreadWaves], cum[1][kSpreadWaves]);
    longest = len > longest ? len : longest;
  }
  longest += 1; // (rounding)
  return longest > kSpreadMaxShare ? 0 : longest;
}

uint32_t spread_longest_share(const DeviceGeom &dg, uint64_t n_chains)
{
  uint16_t cum[2][17];
  spread_weights(dg, cum);
  return spread_longest_share_of(dg, n_chains, cum);
}

// the class weights of a one-round launch whose waves decode `run_groups` groups each on average: the one-chain-per-wave launch's, this
// device's own once calibrated — the set fitted nearest to that run length where several are (hsrans_ctx_calibrate_runs)
void dealt_weights_now(const DeviceGeom &dg, uint64_t run_groups, uint32_t bits, uint32_t weights_out[8])
{
  direct_weights_for(dg, run_groups, weights_out);
  // 13 / 14 bits (k_decode_dealt_rank): the rank loop's three dependent LDS reads per group make a wave's age count for less — the flatter
  // class lengths fitted for the rank-table one-chain-per-wave launch (g_direct_weights6); nothing calibrates this loop per device.
  // 100 MB in 256 KiB blocks, G = 16, rotated: 14 bits 54.1 us with the 11-bit loop's lengths, 51.3-52.1 with these; 13 bits 52.1 -> 50.0
  // (tools/dealt_weights_probe.py)
  if (bits >= 13)
    for (uint32_t k = 0; k < 8; k++)
      weights_out[k] = g_direct_weights6[k];
  if (const char *e = getenv("HSRANS_DEALT_WEIGHTS")) // tuning: 8 per-mille class lengths for the dealt launch (read at every dealing)
  {
    uint32_t v[8], n = 0;
    for (const char *p = e; n < 8 && *p; n++)
    {
      v[n] = (uint32_t)strtoul(p, (char **)&p, 10);
      if (*p == ',')
        p++;
    }
    if (n == 8)
      for (uint32_t k = 0; k < 8; k++)
        weights_out[k] = v[k] >= 10 ? v[k] : 10;
  }
}
static void cum_from_weights(const uint32_t w8[8], uint16_t (*cum_out)[17])
{
  for (uint32_t hf = 0; hf < 2; hf++)
  {
    uint32_t cum = 0;
    for (uint32_t k = 0; k <= 16; k++)
    {
      cum_out[hf][k] = (uint16_t)cum;
      cum += k < kSpreadWaves ? w8[hf * 4 + k / (kSpreadWaves / 4)] / 10 : 0;
    }
  }
}

static uint32_t g_dealt = 1;          // HSRANS_DEALT=0: grouped plans keep k_decode_spread / k_decode_grouped (comparison)
static uint32_t g_dealt_min_chains = 1400; // HSRANS_DEALT_MIN_CHAINS: chains per 1,000 waves of the launch below which a plan is not dealt (tuning; 100 MB in 256 KiB blocks, G = 128 — 1.49 chains a wave — 58.5 us grouped, 47.4 dealt; one chain a wave leaves the class weights nothing to work with)
static uint32_t g_dealt_wt = 1;       // HSRANS_DEALT_WT=0: k_decode_dealt's stores as `nt` instead of written through (comparison)

bool deal_shares(const DeviceGeom &dg, const uint32_t *block_begin, uint32_t n_blocks, uint32_t n_chains, uint64_t total_groups, uint32_t bits, DealtTable *out,
                 uint32_t weights_out[8])
{
  dealt_weights_now(dg, total_groups / ((uint64_t)spread_grid(dg) * kSpreadWaves), bits, weights_out);
  // (read at every dealing, not once: tools/ab_probe.py and the tests switch sides inside one process)
  const char *e_on = getenv("HSRANS_DEALT"), *e_min = getenv("HSRANS_DEALT_MIN_CHAINS");
  g_dealt = e_on ? (uint32_t)atoi(e_on) : 1;
  g_dealt_min_chains = e_min && atoi(e_min) > 0 ? (uint32_t)atoi(e_min) : 1400;
  const uint32_t grid = spread_grid(dg);
  if (!g_dealt || !g_spread || grid > kDealtGridMax || n_blocks == 0 || block_begin[0] != 0 || block_begin[n_blocks] != n_chains)
    return false;
  if ((uint64_t)n_chains * 1000 < (uint64_t)grid * kSpreadWaves * g_dealt_min_chains)
    return false;
  // a block and more for every 8-wave workgroup slot of the grouped launch (four per CU): that launch's home ground — one block per slot and
  // round, blocks behind the first round by ticket (256 MiB in 256 KiB blocks: 107-112 us grouped, 119-121 dealt with shares cut at two blocks)
  if (n_blocks >= 2 * grid)
    return false;
  // a workgroup's weight = the sum of its waves' (four waves per class; as the kernel's cumulative table has them); the first half of the grid is resident first
  uint64_t w1 = 0, w2 = 0;
  for (uint32_t k = 0; k < 4; k++)
    w1 += weights_out[k] / 10, w2 += weights_out[4 + k] / 10;
  const uint32_t fh = (grid + 1) / 2;
  uint64_t weight_left = (uint64_t)fh * w1 + (uint64_t)(grid - fh) * w2;
  uint32_t blk = 0; // block of the next share's first chain
  out->begin[0] = 0;
  for (uint32_t b = 0; b < grid; b++)
  {
    // what is left, by the weights that are left: a share that was cut back at a third block leaves its chains to ALL the workgroups behind
    // it (given to the next one alone — targets as fixed fractions of the whole — shares of one grid half ranged from 1.0 to 1.5 blocks
    // at 128 MiB in 256 KiB blocks, and the launch ended 10 us after its median wave: tools/stamps_dealt.py)
    const uint32_t c0 = out->begin[b];
    const uint64_t wb = b < fh ? w1 : w2;
    uint32_t e = b + 1 == grid ? n_chains : c0 + (uint32_t)(((uint64_t)(n_chains - c0) * wb + weight_left / 2) / weight_left);
    weight_left -= wb;
    e = e > n_chains ? n_chains : e;
    while (blk + 1 < n_blocks && block_begin[blk + 1] <= c0) // the block chain c0 is in
      blk++;
    uint32_t split = 0xFFFF;
    if (c0 < n_chains)
    {
      const uint32_t end_a = block_begin[blk + 1];                                   // first block ends here
      const uint32_t end_b = blk + 2 <= n_blocks ? block_begin[blk + 2] : n_chains; // the second one here: the share may not go on
      e = e > end_b ? end_b : e;
      if (e > end_a)
      {
        if (end_a - c0 > 0xFFFE)
          return false;
        split = end_a - c0;
      }
    }
    if (e - c0 > 0xFFFE)
      return false;
    out->begin[b + 1] = e;
    out->split[b] = (uint16_t)split;
  }
  for (uint32_t b = grid; b < kDealtGridMax; b++)
    out->begin[b + 1] = n_chains, out->split[b] = 0xFFFF;
  if (out->begin[grid] != n_chains) // the cuts at third blocks left chains over — blocks much shorter than a share: the grouped launch's case
    return false;
  // ... and where the cuts bent the shares too far from the weights (shares of about two blocks and more), the launch would end with its
  // most loaded workgroup: the grouped launch's case as well
  const double mean = (double)n_chains / (double)((uint64_t)fh * w1 + (uint64_t)(grid - fh) * w2);
  for (uint32_t b = 0; b < grid; b++)
    if ((double)(out->begin[b + 1] - out->begin[b]) / (double)(b < fh ? w1 : w2) > 1.25 * mean + 1.0 / (double)w2)
      return false;
  return true;
}

// Everything about a launch that follows from the plan header and the device alone (no pointers): the table layout, the
// workgroup shape and the grid.  launch_decode uses it; direct_boundaries uses it to size one chain per resident wave.
LaunchShape launch_shape(const PlanHeader &h, const DeviceGeom &dg, bool persistent, uint32_t table_mode, uint32_t n_groups, bool index_pass, bool direct, bool dual)
{
  read_tuning_once();
  LaunchShape L{};
  const bool walk = (h.flags & kPlanWalk) != 0;
  const bool grouped = n_groups != 0 && !index_pass;
  L.walk = walk;
  L.shared = !walk && (grouped || (h.shared_hist != 0 && h.n_chains > 1));
  // 64-bit entries only where one table serves a whole workgroup (LDS: 16 KiB table + 16 x 2.25 KiB rings, two per CU)
  const bool rank_table = L.shared && persistent && table_mode == kModeRank;
  const bool spill = L.shared && persistent && table_mode == kModeSpill;
  // (grouped launches from 13 bits on: the rank table, built per block on the device — beside the 64 / 128 KiB one-lookup tables
  // only one workgroup fits a CU: 100 MB in 256 KiB blocks + G=32: 0.269 / 0.206 / 0.190 at 13 / 14 / 15 bits)
  const bool grouped_rank = grouped && g_rank_table && h.bits >= 13;
  L.mode = spill ? kModeSpill : rank_table || grouped_rank ? kModeRank : L.shared && h.bits <= pack64_max_bits() ? kModePack64 : h.bits >= 13 ? kModeTwoLevel : h.bits == 12 ? kModePackM1 : kModePack;
  const bool two_level = L.mode == kModeTwoLevel;
  const uint32_t table_bytes = table_bytes_for(L.mode, h.bits);
  const uint32_t wave_bytes = kWaveRingBytes + ((table_bytes + 15) & ~15u); // private rings + table
  uint32_t waves, lds, grid;
  const uint32_t dual_ring = kFastRingBytes;
  L.dual = dual && L.shared && persistent && (L.mode == kModePack64 || L.mode == kModeRank) && g_dual_waves * 2 * dual_ring + table_bytes <= dg.max_lds;
  if (L.dual)
  {
    // k_decode_dual: workgroups of 16 (or HSRANS_DUAL_WAVES) waves, two rings per wave, wave w decodes chains 2w and 2w + 1
    waves = g_dual_waves;
    lds = waves * 2 * dual_ring + table_bytes;
    const uint32_t per_cu = dg.max_lds / lds ? dg.max_lds / lds : 1;
    L.resident = dg.num_cus * (per_cu * waves > 32 ? 32 / waves : per_cu);
    grid = (h.n_chains + 2 * waves - 1) / (2 * waves);
    if (grid > L.resident)
      grid = L.resident;
  }
  else if (L.shared)
  {
    // (k_decode_direct's hand-scheduled loop wants a whole-chunk mirror behind every ring)
    const uint32_t ring = fast_ring_mode(L.mode) ? kFastRingBytes : kWaveRingBytes;
    waves = g_waves_per_wg;
    // Grouped launches (one workgroup per block, round after round): FOUR workgroups of 8 waves per CU instead of two of 16
    // wherever four fit the LDS.  A round is table build + records + first chunks (~12 us in which the workgroup decodes
    // nothing) and then the decode; with two workgroups per CU the other one is alone on the CU meanwhile, 4 waves per SIMD,
    // which is too few to hide the loop's LDS latency.  With four, three are decoding while one is between rounds, and a round
    // is twice as long for the same fixed cost.  Measured at 2^30 bytes: 256 KiB blocks 0.461 -> 0.492 of 8 TB/s, 64 KiB blocks
    // 0.358 -> 0.427 (1 MiB blocks: unchanged).  HSRANS_WAVES_PER_WG still overrides.
    // (only with at least four groups per CU: fewer, e.g. a 100 MB stream in 256 KiB blocks, fill more wave slots as 16-wave workgroups)
    if (grouped && getenv("HSRANS_WAVES_PER_WG") == nullptr && 4 * (8 * ring + table_bytes + 64 + 1024) <= dg.max_lds && n_groups >= 4 * dg.num_cus)
      waves = 8;
    if (L.mode == kModeRank && waves * ring + table_bytes > dg.max_lds / 2)
      waves = 12; // 15 bits: 48 KiB of tables + 12 rings = 75 KiB, two workgroups per CU
    if (waves * ring + table_bytes > dg.max_lds && table_bytes + 4 * ring <= dg.max_lds)
      waves = (dg.max_lds - table_bytes) / ring / 4 * 4; // a big table: as many waves as still fit (multiple of 4: one per SIMD)
    while (waves > 1 && (waves / 2 >= h.n_chains || waves * ring + table_bytes > dg.max_lds))
      waves /= 2;
    lds = waves * ring + table_bytes + (grouped ? 64 + 1024 : 0); // (run_grouped's two next-group words and its table-build scratch)
    grid = (h.n_chains + waves - 1) / waves;
    if (grouped)
      grid = n_groups;
    const uint32_t per_cu = dg.max_lds / lds ? dg.max_lds / lds : 1; // workgroups one CU can hold (LDS-limited; 32 waves max)
    L.resident = dg.num_cus * (per_cu * waves > 32 ? 32 / waves : per_cu);
    if ((persistent || grouped) && grid > L.resident)
      grid = L.resident;
  }
  else
  {
    // 32-state plans: two chains per wave, one per half, when two tables fit (run_private_pair); not for the index-build pass
    const bool pair = (g_private_pair == 2 || (g_private_pair == 1 && h.n_chains >= 32 * dg.num_cus)) && !walk && h.states == 32 && h.n_chains > 1 &&
                      table_bytes <= 16384 && !index_pass;
    L.private_pair = pair ? 1 : 0;
    const uint32_t wave_lds = pair ? wave_bytes + ((table_bytes + 15) & ~15u) : wave_bytes;
    const uint32_t work = pair ? (h.n_chains + 1) / 2 : h.n_chains;
    waves = walk ? 1 : 4;
    while (waves > 1 && (waves * wave_lds > dg.max_lds / 2 || waves / 2 >= work))
      waves /= 2;
    lds = waves * wave_lds;
    grid = walk ? 1 : (work + waves - 1) / waves;
    L.resident = dg.num_cus * (dg.max_lds / (lds ? lds : 1));
  }
  L.waves = waves;
  L.lds = lds;
  L.grid = grid ? grid : 1;
  // run-length weights of the 8 wave classes (per mille of the mean): class = (workgroup in the grid's second half) * 4 + wave / (waves / 4)
  const bool weighted = (waves == 16 || waves == 12) && (!two_level || g_weights_two_level);
  for (uint32_t k = 0; k < 8; k++)
    L.weights[k] = !weighted ? 1000
                   : L.dual   ? (L.mode == kModeRank ? g_dual_weights_wide : g_dual_weights)[k]
                   : direct   ? (L.grid > dg.num_cus ? (waves == 16 ? (h.states == 32 ? g_direct_weights_pair : dg.have_direct_weights ? dg.direct_weights : g_direct_weights) : g_direct_weights6) : (waves == 16 ? g_direct_weights4 : g_direct_weights3))[k]
                              : (L.grid > dg.num_cus ? (persistent && !grouped && h.states == 32 ? g_direct_weights_pair : g_slot_weights) : g_slot_weights4)[k];
  return L;
}

void direct_weights_for(const DeviceGeom &dg, uint64_t run_groups, uint32_t out[8])
{
  read_tuning_once();
  const uint32_t *base = dg.have_direct_weights ? dg.direct_weights : g_direct_weights;
  if (dg.n_weight_sets == 0 || run_groups == 0 || getenv("HSRANS_DIRECT_WEIGHTS") != nullptr) // (an explicit override rules)
  {
    for (uint32_t k = 0; k < 8; k++)
      out[k] = base[k];
    return;
  }
  const uint32_t n = dg.n_weight_sets < 4 ? dg.n_weight_sets : 4;
  uint32_t hi = 0;
  while (hi < n && dg.set_run[hi] < run_groups)
    hi++;
  if (hi == 0 || hi == n) // outside the fitted lengths: the nearest set
  {
    const uint32_t k0 = hi == 0 ? 0 : n - 1;
    for (uint32_t k = 0; k < 8; k++)
      out[k] = dg.set_weights[k0][k];
    return;
  }
  const double a = log((double)dg.set_run[hi - 1]), b = log((double)dg.set_run[hi]), x = log((double)run_groups);
  const double t = b > a ? (x - a) / (b - a) : 0.0;
  for (uint32_t k = 0; k < 8; k++)
    out[k] = (uint32_t)((1.0 - t) * dg.set_weights[hi - 1][k] + t * dg.set_weights[hi][k] + 0.5);
}

// Group boundaries that cut `total_groups` whole groups into one chain per wave of the direct launch this device would use
// for a mergeable plan of (states, bits): chain w belongs to wave w, its length follows the wave's class weight.
// Boundaries are multiples of 4 groups (the decode loop stores 4 groups at a time).  Returns the number of chains;
// out[k] = first group of chain k + 1 (k < chains - 1).
size_t direct_boundaries(const DeviceGeom &dg, uint32_t states, uint32_t bits, uint64_t total_groups, uint64_t *out, size_t cap)
{
  PlanHeader h{};
  h.states = states;
  h.bits = bits;
  h.shared_hist = 1;
  h.n_chains = 1u << 30; // "many": the full machine
  const TableChoice tc = choose_table(bits, states, true);
  const LaunchShape L = launch_shape(h, dg, true, tc.mode, 0, false, true, tc.dual);
  const uint32_t runs_per_wave = (states == 32 || L.dual) ? 2 : 1;
  const uint64_t W = (uint64_t)L.grid * L.waves;
  uint64_t chains = W * runs_per_wave;
  const uint64_t all_units = total_groups / 4; // boundaries in units of 4 groups
  if (chains > all_units / 8) // a chain is worth its 308 bytes of index and its prologue from about 32 groups on
    chains = all_units / 8 ? all_units / 8 : 1;
  if (chains <= 1)
    return 1;
  if (chains - 1 > cap)
    return 0;
  // cumulative weight up to chain k, then boundaries at units * cum / all
  const uint32_t first_half = (L.grid + 1) / 2;
  const uint32_t per_class = L.waves >= 4 ? L.waves / 4 : 1;
  // the headline's launch shape (64 states, 8-byte table, two 16-wave workgroups per CU): the lengths fitted for runs of this length
  uint32_t w8[8];
  for (uint32_t k = 0; k < 8; k++)
    w8[k] = L.weights[k];
  if (states == 64 && !L.dual && L.waves == 16 && L.grid > dg.num_cus && L.mode == kModePack64)
    direct_weights_for(dg, total_groups / chains, w8);
  auto weight_of = [&](uint64_t chain) {
    const uint64_t w = chain / runs_per_wave;
    const uint32_t blk = (uint32_t)(w / L.waves), wave_in_wg = (uint32_t)(w % L.waves);
    const uint32_t cls = (blk >= first_half ? 4 : 0) + (wave_in_wg / per_class < 4 ? wave_in_wg / per_class : 3);
    return (uint64_t)w8[cls];
  };
  uint64_t all = 0;
  for (uint64_t k = 0; k < chains; k++)
    all += weight_of(k);
  uint64_t cum = 0, prev = 0;
  size_t n = 0;
  for (uint64_t k = 0; k + 1 < chains; k++)
  {
    cum += weight_of(k);
    uint64_t b = (uint64_t)((unsigned __int128)all_units * cum / all);
    if (b <= prev)
      b = prev + 1; // every chain gets at least one unit
    if (b >= all_units)
      break;
    out[n++] = b * 4;
    prev = b;
  }
  return n + 1;
}

// The batch launch of 64-state plans with 8-byte tables: the one-chain-per-wave launch's own shape (two 16-wave workgroups per CU,
// LDS = 16 rings + the widest member's table) and its age-class weights (the device's own once calibrated).
BatchShape batch_direct_shape(const DeviceGeom &dg, uint32_t max_bits, uint64_t total_groups, uint32_t states)
{
  PlanHeader h{};
  h.states = states;
  h.bits = max_bits;
  h.shared_hist = 1;
  h.n_chains = 1u << 30; // "many": the full machine
  // 13-15 bits (64 states): k_decode_dual's shape — one 16-wave workgroup per CU, two chains per wave, the 8-byte table at 13 bits,
  // the rank table at 14 / 15 — and its class lengths
  const bool wide = states == 64 && max_bits >= 13;
  const LaunchShape L = launch_shape(h, dg, true, wide && max_bits >= 14 ? kModeRank : kModePack64, 0, false, true, wide);
  BatchShape b{};
  b.kind = states == 32 ? kBatchPair : !wide ? kBatchDirect : max_bits >= 14 ? kBatchDualRank : kBatchDualPack;
  b.grid = L.grid;
  b.waves = L.waves;
  b.lds = L.lds;
  b.states = states;
  for (uint32_t k = 0; k < 8; k++)
    b.weights[k] = L.weights[k]; // (32 states: the pair loop's own class lengths, g_direct_weights_pair)
  if (states == 64 && total_groups != 0 && L.waves == 16 && L.grid > dg.num_cus)
    direct_weights_for(dg, total_groups / ((uint64_t)L.grid * L.waves), b.weights);
  // HSRANS_BATCH_WEIGHTS (tuning; read at every batch creation so that one process can try several): 8 per-mille run lengths
  if (const char *e = getenv("HSRANS_BATCH_WEIGHTS"))
  {
    uint32_t v[8], n = 0;
    for (const char *p = e; n < 8 && *p; n++)
    {
      v[n] = (uint32_t)strtoul(p, (char **)&p, 10);
      if (*p == ',')
        p++;
    }
    if (n == 8)
      for (uint32_t k = 0; k < 8; k++)
        b.weights[k] = v[k] ? v[k] : 1;
  }
  return b;
}

hipError_t launch_batch_direct(const BatchParams &bp, const BatchShape &shape, hipStream_t stream)
{
  (void)hipGetLastError();
  if (shape.kind == kBatchPair)
    hipLaunchKernelGGL(k_decode_batch_pair<kModePack64>, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  else if (shape.kind == kBatchDualPack)
    hipLaunchKernelGGL(k_decode_batch_dual<kModePack64>, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  else if (shape.kind == kBatchDualRank)
    hipLaunchKernelGGL(k_decode_batch_dual<kModeRank>, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  else if (bp.finish != nullptr)
    hipLaunchKernelGGL(k_calibrate_batch, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  else
    hipLaunchKernelGGL(k_decode_batch<kModePack64>, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  return hipGetLastError();
}

// the per-wave shares of a group's chains as cumulative class weights (run_grouped: kp.group_cum)
static void group_cum_of(const LaunchShape &L, uint16_t (*cum_out)[17])
{
  const uint32_t per_class = L.waves >= 4 ? L.waves / 4 : 1;
  for (uint32_t hf = 0; hf < 2; hf++)
  {
    uint32_t cum = 0;
    for (uint32_t k = 0; k <= 16; k++)
    {
      cum_out[hf][k] = (uint16_t)cum;
      const uint32_t cls = k / per_class < 4 ? k / per_class : 3;
      cum += k < L.waves ? L.weights[hf * 4 + cls] / 10 : 0;
    }
  }
}

// The grouped batch launch (64-state members, bits <= 12: the 8-byte table built per group): the shape k_decode_grouped would get for a
// plan with all the members' groups and chains
BatchGroupShape batch_grouped_shape(const DeviceGeom &dg, uint32_t bits, uint32_t n_groups, uint64_t n_chains)
{
  PlanHeader h{};
  h.states = 64;
  h.bits = bits;
  h.n_chains = n_chains > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_chains;
  const LaunchShape L = launch_shape(h, dg, false, 0, n_groups, false, false, false);
  BatchGroupShape b{};
  b.grid = L.grid;
  b.waves = L.waves;
  b.lds = L.lds;
  group_cum_of(L, b.group_cum);
  return b;
}

hipError_t launch_batch_grouped(const BatchGroupParams &bp, const BatchGroupShape &shape, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_decode_grouped_batch<kModePack64>, dim3(shape.grid), dim3(shape.waves * 64), shape.lds, stream, bp);
  return hipGetLastError();
}

static hipError_t launch_dealt(const KParams &kp, const PlanHeader &h, const DeviceGeom &dg, hipStream_t stream, LaunchInfo *info, const PartPlan *parts, const DealtTable &dt,
                               const uint32_t *w8)
{
  const char *e_wt = getenv("HSRANS_DEALT_WT");
  g_dealt_wt = e_wt ? (uint32_t)atoi(e_wt) : 1;
  DealtParams dp{};
  dp.stream = kp.stream;
  dp.stream_len = kp.stream_len;
  dp.stream_lo = kp.stream_lo;
  dp.out = kp.out;
  dp.out_cap = kp.out_cap;
  dp.pieces = (const Piece *)(kp.plan + plan_pieces_off(h.n_chains));
  dp.states = (const uint32_t *)(kp.plan + plan_states_off(h.n_chains, h.n_chains));
  dp.status = kp.status;
  dp.stamps = kp.stamps;
  dp.n_chains = h.n_chains;
  dp.bits = h.bits;
  const uint32_t grid = spread_grid(dg);
  cum_from_weights(w8, dp.cum); // (the weights the shares were dealt with)
  {
    // a second prologue is ~2.5 us = ~17 groups of decoding at 8 waves per SIMD (HSRANS_DEALT_GAP_GROUPS: tuning)
    const char *e_gap = getenv("HSRANS_DEALT_GAP_GROUPS");
    const uint64_t gap_groups = e_gap ? (uint64_t)atoi(e_gap) : 17;
    const uint64_t per_chain = h.interval ? h.interval : h.n_chains ? (uint64_t)h.decoded_len / 64 / h.n_chains : 0; // groups per chain: the index interval (a sliced plan keeps the stream's header)
    dp.gap_chains = per_chain ? (uint32_t)((gap_groups + per_chain / 2) / per_chain) : 0;
  }
  const bool rank = h.bits >= 13; // (13 / 14 bits: two rank tables, k_decode_dealt_rank)
  const uint32_t lds = kSpreadWaves * kFastRingBytes + 2 * table_bytes_for(rank ? kModeRank : kModePack64, h.bits) + 2048;
  if (parts != nullptr)
  {
    dp.parts = kp.parts;
    dp.parts.n = parts->n;
    uint32_t begin = 0;
    for (uint32_t k = 0; k < parts->n; k++)
    {
      const uint32_t end = parts->chain_end[k];
      uint32_t units = 0; // workgroups whose share overlaps part k (run_dealt counts itself by the same rule)
      for (uint32_t b = 0; b < grid && end > begin; b++)
        units += dt.begin[b + 1] > dt.begin[b] && dt.begin[b] < end && dt.begin[b + 1] > begin ? 1 : 0;
      dp.parts.chain_end[k] = end;
      dp.parts.target[k] = (parts->cum[k] += units);
      begin = end > begin ? end : begin;
    }
  }
  if (info)
  {
    *info = LaunchInfo{};
    info->grid = grid;
    info->block = kSpreadWaves * 64;
    info->lds_bytes = lds;
    info->waves_per_block = kSpreadWaves;
    info->chains = h.n_chains;
    info->shared_table = 1;
    info->table_mode = (uint32_t)(rank ? kModeRank : kModePack64);
    info->chains_per_wave = 1;
    for (uint32_t k = 0; k < 8; k++)
      info->class_weights[k] = w8[k];
    info->spread = 2;
  }
  (void)hipGetLastError();
  if (rank && h.bits == 13 && parts != nullptr)
    hipLaunchKernelGGL((k_decode_dealt_rank<13, true>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else if (rank && h.bits == 13)
    hipLaunchKernelGGL((k_decode_dealt_rank<13, false>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else if (rank && parts != nullptr)
    hipLaunchKernelGGL((k_decode_dealt_rank<14, true>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else if (rank)
    hipLaunchKernelGGL((k_decode_dealt_rank<14, false>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else if (parts != nullptr)
    hipLaunchKernelGGL((k_decode_dealt<true, true>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else if (g_dealt_wt)
    hipLaunchKernelGGL((k_decode_dealt<true, false>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  else
    hipLaunchKernelGGL((k_decode_dealt<false, false>), dim3(grid), dim3(kSpreadWaves * 64), lds, stream, dp, dt);
  return hipGetLastError();
}

hipError_t launch_decode(const KParams &kp_in, const PlanHeader &h, const DeviceGeom &dg, hipStream_t stream, LaunchInfo *info, const PartPlan *parts, const DealtTable *dealt,
                         const uint32_t *dealt_weights)
{
  KParams kp = kp_in;
  if (parts != nullptr && (parts->n == 0 || parts->n > kMaxLaunchParts))
    return hipErrorInvalidValue;
  if (dealt != nullptr && dealt_weights != nullptr && kp.groups != nullptr && kp.groups_lean && kp.ckpt_interval == 0 && kp.ckpt_groups == nullptr && h.states == 64 &&
      (h.bits <= 11 || h.bits == 13 || h.bits == 14) && h.n_pieces == h.n_chains &&
      2 * (kSpreadWaves * kFastRingBytes + 2 * table_bytes_for(h.bits >= 13 ? kModeRank : kModePack64, h.bits) + 2048) <= dg.max_lds)
    return launch_dealt(kp, h, dg, stream, info, parts, *dealt, dealt_weights);
  const bool persistent = kp.pa.pieces != nullptr;
  const bool index_pass = kp.ckpt_interval != 0 || kp.ckpt_groups != nullptr;
  const LaunchShape L = launch_shape(h, dg, persistent, persistent && kp.pa.table != nullptr ? kp.pa.table_mode : 0, kp.groups != nullptr ? kp.n_groups : 0, index_pass, persistent && kp.pa.interval == 0, persistent && kp.pa.dual != 0);
  const bool grouped = kp.groups != nullptr && !index_pass;
  const uint32_t waves = L.waves, grid = L.grid;
  kp.private_pair = L.private_pair;

  if (grouped)
    group_cum_of(L, kp.group_cum);
  if (persistent && kp.pa.interval != 0)
  {
    // static share: a fixed fraction of the chains, split over the waves by class weight; the rest goes through the queues
    const uint64_t W = (uint64_t)grid * waves;
    // 32-state streams: two runs per wave (run_persistent_pair)
    kp.pa.static_per_wave = (uint32_t)((uint64_t)h.n_chains * g_static_percent / 100 / (h.states == 32 ? 2 * W : W));
    const uint32_t first_half = (grid + 1) / 2, second_half = grid - first_half;
    const uint32_t per_class = waves >= 4 ? waves / 4 : 1, classes = waves / per_class;
    const uint32_t runs_per_wave = h.states == 32 ? 2 : 1; // run_persistent_pair decodes two runs side by side
    uint32_t longest = 0;
    for (uint32_t hf = 0; hf < 2; hf++)
    {
      uint32_t off = 0;
      for (uint32_t k = 0; k < 4; k++)
      {
        kp.pa.run_len[hf * 4 + k] = k < classes ? (uint32_t)((uint64_t)h.n_chains * g_static_percent / 100 * L.weights[hf * 4 + k] / (1000 * W * runs_per_wave)) : 0;
        kp.pa.class_off[hf * 4 + k] = off;
        off += kp.pa.run_len[hf * 4 + k] * per_class * runs_per_wave;
        longest = kp.pa.run_len[hf * 4 + k] > longest ? kp.pa.run_len[hf * 4 + k] : longest;
      }
      kp.pa.wg_chains[hf] = off;
    }
    kp.pa.half_base[0] = 0;
    kp.pa.half_base[1] = first_half * kp.pa.wg_chains[0];
    kp.pa.static_total = first_half * kp.pa.wg_chains[0] + second_half * kp.pa.wg_chains[1];
    if (kp.pa.static_total > h.n_chains) // weights sum to <= 8000 by construction; belt and braces
      return hipErrorInvalidValue;
    // a merged run is read through one 32-bit window of the stream (at most one 16-bit word per symbol)
    if ((uint64_t)(longest ? longest : 1) * kp.pa.interval * h.states * 2 >= 0xFFFF0000ull)
      return hipErrorInvalidValue;
  }
  if (persistent && kp.pa.interval == 0)
  {
    // one-chain-per-wave launches: the plan's chains as W runs of consecutive chains (run_direct); 32-state and two-chain launches keep their own dealing
    const uint64_t W = (uint64_t)grid * waves;
    kp.pa.run_chains = h.states == 64 && !L.dual ? (uint32_t)((h.n_chains + W - 1) / W) : 1;
  }
  if (kp.single.valid && !index_pass && g_single_fast)
  {
    // one chain of one piece (a raw stream without an index): the two-wave latency kernel
    const uint32_t lds = (8u << h.bits) + (kp.single.ring_entries + kSingleMirror) * 16 + 64;
    if (info)
    {
      *info = LaunchInfo{};
      info->grid = 1;
      info->block = 128;
      info->lds_bytes = lds;
      info->waves_per_block = 2;
      info->chains = 1;
      info->table_mode = kModePack64;
      info->chains_per_wave = 1;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_decode_single, dim3(1), dim3(128), lds, stream, kp);
    return hipGetLastError();
  }
  KernelFn fn = kernel_for(L.mode, L.shared);
  if (L.dual)
    fn = L.mode == kModeRank ? (KernelFn)k_decode_dual<kModeRank> : (KernelFn)k_decode_dual<kModePack64>;
  else if (persistent && kp.pa.interval != 0 && L.shared && (L.mode == kModePack64 || L.mode == kModeRank) && kp.pa.table != nullptr && !index_pass && g_persist_kernel)
    fn = L.mode == kModeRank ? (KernelFn)k_decode_persist<kModeRank> : (KernelFn)k_decode_persist<kModePack64>;

  else if (grouped && L.shared)
    switch (L.mode)
    {
    case kModePack: fn = k_decode_grouped<kModePack, false>; break;
    case kModePackM1: fn = k_decode_grouped<kModePackM1, false>; break;
    case kModeTwoLevel: fn = kp.groups_lean ? k_decode_grouped<kModeTwoLevel, true> : k_decode_grouped<kModeTwoLevel, false>; break;
    case kModeRank: fn = kp.groups_lean ? k_decode_grouped<kModeRank, true> : k_decode_grouped<kModeRank, false>; break;
    default: fn = kp.groups_lean ? k_decode_grouped<kModePack64, true> : k_decode_grouped<kModePack64, false>; break;
    }
  else if (persistent && kp.pa.interval == 0 && L.shared)
    switch (L.mode)
    {
    case kModePack: fn = k_decode_direct<kModePack>; break;
    case kModePackM1: fn = k_decode_direct<kModePackM1>; break;
    case kModeTwoLevel: fn = k_decode_direct<kModeTwoLevel>; break;
    case kModeRank: fn = k_decode_direct<kModeRank>; break;
    case kModeSpill: fn = k_decode_direct<kModeSpill>; break;
    default: fn = kp.finish != nullptr && h.states == 64 ? (KernelFn)k_calibrate : (KernelFn)k_decode_direct<kModePack64>; break;
    }
  // grouped plans with few, large blocks (kp.spread: the fewest chains of a block, where the host found the plan eligible): every
  // resident workgroup takes its share of ALL chains and builds the one or two tables it needs (kernels_spread.h)
  uint32_t launch_grid = grid, launch_waves = waves, launch_lds = L.lds;
  bool spread = false;
  if (grouped && L.shared && kp.spread != 0 && kp.groups_lean && L.mode == kModePack64 && g_spread)
  {
    uint16_t spread_cum[2][17];
    spread_weights(dg, spread_cum); // one snapshot: the same weights for the check below and for the kernel
    const uint32_t longest = spread_longest_share_of(dg, h.n_chains, spread_cum);
    const uint32_t slds = kSpreadWaves * kFastRingBytes + 2 * table_bytes_for(kModePack64, h.bits) + (kSpreadMaxShare + 1) * (uint32_t)sizeof(Piece);
    if (longest != 0 && longest < kp.spread && 2 * slds <= dg.max_lds) // (a share shorter than every block touches at most two)
    {
      spread = true;
      fn = k_decode_spread<kModePack64>;
      launch_grid = spread_grid(dg);
      launch_waves = kSpreadWaves;
      launch_lds = slds;
      memcpy(kp.group_cum, spread_cum, sizeof(kp.group_cum));
      kp.pa.n_chains = h.n_chains; // (single-piece chains: n_pieces == n_chains)
      kp.pa.S = h.states;
      kp.pa.bits = h.bits;
    }
  }
  if (parts != nullptr)
  {
    // a rank's sub-runs in one launch: only the kernels that count their units into the sub-runs (PartArgs)
    if (spread)
      fn = k_decode_spread<kModePack64, true>;
    else if (grouped && L.shared && kp.groups_lean && L.mode == kModePack64)
      fn = k_decode_grouped<kModePack64, true, true>;
    else if (grouped && L.shared && kp.groups_lean && L.mode == kModeRank)
      fn = k_decode_grouped<kModeRank, true, true>;
    else
      return hipErrorNotSupported;
    kp.parts.n = parts->n;
    uint32_t begin = 0;
    for (uint32_t k = 0; k < parts->n; k++)
    {
      const uint32_t end = parts->chain_end[k];
      uint32_t units = parts->group_units[k];
      if (spread) // workgroups whose share of the chains overlaps part k (run_spread counts itself by the same rule)
      {
        units = 0;
        for (uint32_t b = 0; b < launch_grid && end > begin; b++)
        {
          const uint32_t c0 = spread_share_begin(h.n_chains, b, launch_grid, kp.group_cum[0][launch_waves], kp.group_cum[1][launch_waves]);
          const uint32_t c1 = spread_share_begin(h.n_chains, b + 1, launch_grid, kp.group_cum[0][launch_waves], kp.group_cum[1][launch_waves]);
          units += c1 > c0 && c0 < end && c1 > begin ? 1 : 0;
        }
      }
      kp.parts.chain_end[k] = end;
      kp.parts.target[k] = (parts->cum[k] += units);
      begin = end > begin ? end : begin;
    }
  }
  if (info)
  {
    info->grid = launch_grid;
    info->block = launch_waves * 64;
    info->lds_bytes = launch_lds;
    info->waves_per_block = launch_waves;
    info->chains = h.n_chains;
    info->shared_table = L.shared;
    info->walk = L.walk;
    info->two_level = L.mode == kModeTwoLevel;
    info->table_mode = (uint32_t)L.mode;
    info->chains_per_wave = L.dual ? 2 : 1;
    for (uint32_t k = 0; k < 8; k++)
      info->class_weights[k] = L.weights[k];
    info->dynamic_groups = grouped && !spread && kp.group_tickets != nullptr && kp.n_groups > grid ? 1 : 0;
    info->spread = spread ? 1 : 0;
  }
  (void)hipGetLastError(); // (sticky per thread: an earlier failed call — e.g. an allocation a hostile stream asked for — is not this launch's error)
  hipLaunchKernelGGL(fn, dim3(launch_grid), dim3(launch_waves * 64), launch_lds, stream, kp);
  return hipGetLastError();
}

} // namespace hsrans
