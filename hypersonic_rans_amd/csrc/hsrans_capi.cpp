// C ABI of libhsrans_hip.so (declared in include/hsrans_hip.h).  Nothing here decodes on the CPU: every decode entry
// ends in a launch of the gfx950 kernels in hsrans_kernels.hip and fails when no usable device exists.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/hsrans_hip.h"
#include "hsrans_host.h"
#include "hsrans_cpu.h"
#include "hsrans_encode.h"
#include "hsrans_kernels.h"

using namespace hsrans;

#include "hsrans_internal.h"
#include "hsrans_batch.h"


extern "C"
{

const char *hsrans_version(void) { return "hsrans-hip 0.1 (gfx950)"; }

// ---- host-side format functions ---------------------------------------------------------------------------------
size_t hsrans_capacity(int container, int states, size_t input_size)
{
  if (!valid_codec(container, states, 10))
    return 0;
  return capacity(container, states, input_size);
}

void hsrans_make_hist(hsrans_hist *hist, const uint8_t *data, size_t size, uint32_t bits)
{
  if (hist && data && size && bits >= 10 && bits <= 15)
    make_hist(hist, data, size, bits);
}

size_t hsrans_encode(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t out_capacity, const hsrans_hist *hist)
try
{
  return encode(container, states, bits, in, length, out, out_capacity, hist, nullptr);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

size_t hsrans_encode_ex(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t out_capacity,
                        const hsrans_hist *hist, hsrans_encode_opts *opts)
try
{
  if (opts)
    opts->plan_size = 0;
  return encode(container, states, bits, in, length, out, out_capacity, hist, opts);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

size_t hsrans_plan_capacity(int container, int states, size_t decoded_size, uint32_t index_interval, uint32_t block_size)
{
  if (!valid_codec(container, states, 10))
    return 0;
  return plan_capacity(container, states, decoded_size, index_interval, block_size);
}

size_t hsrans_plan_build(int container, int states, uint32_t bits, const uint8_t *stream, size_t stream_length, size_t out_capacity, uint8_t *plan_out,
                         size_t plan_capacity)
try
{
  if (plan_out == nullptr)
    return 0;
  return plan_build(container, states, bits, stream, stream_length, out_capacity, plan_out, plan_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

uint32_t hsrans_plan_chain_count(const uint8_t *plan, size_t plan_size)
{
  PlanHeader h;
  return read_header(plan, plan_size, &h) ? h.n_chains : 0;
}

uint64_t hsrans_plan_decoded_length(const uint8_t *plan, size_t plan_size)
{
  PlanHeader h;
  return read_header(plan, plan_size, &h) ? h.decoded_len : 0;
}

size_t hsrans_plan_slice(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint8_t *out, size_t out_capacity)
try
{
  if (out == nullptr)
    return 0;
  return plan_slice(plan, plan_size, first_chain, chain_count, out, out_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

int hsrans_plan_chain_range(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint64_t *begin, uint64_t *end)
{
  if (begin == nullptr || end == nullptr)
    return HSRANS_E_ARG;
  return plan_chain_range(plan, plan_size, first_chain, chain_count, begin, end) ? HSRANS_OK : HSRANS_E_FORMAT;
}

int hsrans_plan_stream_ranges(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint64_t ranges[4])
{
  if (ranges == nullptr)
    return HSRANS_E_ARG;
  return plan_stream_ranges(plan, plan_size, first_chain, chain_count, ranges) ? HSRANS_OK : HSRANS_E_FORMAT;
}

// ---- host SIMD decoders (hsrans_cpu.cpp): never reached from the GPU entries below --------------------------------------
int hsrans_cpu_level(void) { return cpu::best_level(); }

size_t hsrans_decode_cpu(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out,
                         size_t out_capacity, const uint8_t *plan, size_t plan_size)
try
{
  if (in == nullptr || out == nullptr || !valid_codec(container, states, bits))
    return 0;
  if (level < 0)
    level = cpu::best_level();
  if (threads == 0)
    threads = 1;
  if (plan == nullptr)
    return cpu::decode(level, threads, container, states, bits, in, in_length, out, out_capacity);
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || (int)h.container != container || (int)h.states != states || h.bits != bits)
    return 0;
  return cpu::exec_plan(level, threads, plan, plan_size, in, in_length, out, out_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

size_t hsrans_index_build_host(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length,
                               const uint64_t *groups, size_t n_groups, uint8_t *plan_out, size_t plan_capacity)
try
{
  if (level < 0)
    level = cpu::best_level();
  return cpu::index_build(level, threads ? threads : 1, container, states, bits, in, in_length, groups, n_groups, plan_out, plan_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

// ---- GPU side ---------------------------------------------------------------------------------------------------
int hsrans_ctx_create(int device, hsrans_ctx **out_ctx)
{
  if (out_ctx == nullptr)
    return HSRANS_E_ARG;
  *out_ctx = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return HSRANS_E_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess)
    return HSRANS_E_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return HSRANS_E_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) // the code object holds gfx950 ISA only
    return HSRANS_E_NO_DEVICE;
  hsrans_ctx *ctx = new (std::nothrow) hsrans_ctx;
  if (ctx == nullptr)
    return HSRANS_E_HIP;
  ctx->device = device;
  // the marketing name needs the amdgpu.ids table, which minimal images lack: fall back to the ISA name
  snprintf(ctx->name, sizeof(ctx->name), "%s%s%s (%d CUs)", prop.name, prop.name[0] ? " " : "", prop.gcnArchName, prop.multiProcessorCount);
  if (prepare_kernels(&ctx->geom) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&ctx->d_status, 64) != hipSuccess)
  {
    hsrans_ctx_destroy(ctx);
    return HSRANS_E_HIP;
  }
  // HSRANS_CALIBRATE=1: fit the one-chain-per-wave index's class lengths to this device right away (~0.5 s; for callers that cannot
  // call hsrans_ctx_calibrate themselves, e.g. the drop-in entries, which create their one context on first use)
  if (const char *e = getenv("HSRANS_CALIBRATE"))
    if (atoi(e) != 0)
    {
      const int crc = hsrans_ctx_calibrate(ctx, 11, 0, nullptr); // (a failed fit leaves the compiled-in lengths in place: the context is still usable)
      if (crc != HSRANS_OK)
        fprintf(stderr, "hsrans: HSRANS_CALIBRATE=1: hsrans_ctx_calibrate failed (code %d); the compiled-in class lengths stay in use\n", crc);
    }
  *out_ctx = ctx;
  return HSRANS_OK;
}

void hsrans_ctx_destroy(hsrans_ctx *ctx)
{
  if (ctx == nullptr)
    return;
  (void)hipSetDevice(ctx->device);
  if (ctx->cached_pipe)
    hsrans_hpipe_destroy(ctx->cached_pipe);
  if (ctx->host_dplan)
    hsrans_dplan_destroy(ctx->host_dplan);
  if (ctx->host_index)
    hsrans_dplan_destroy(ctx->host_index);
  if (ctx->stream)
    (void)hipStreamDestroy(ctx->stream);
  if (ctx->d_in)
    (void)hipFree(ctx->d_in);
  if (ctx->d_out)
    (void)hipFree(ctx->d_out);
  if (ctx->d_plan)
    (void)hipFree(ctx->d_plan);
  if (ctx->d_enc_scratch)
    (void)hipFree(ctx->d_enc_scratch);
  if (ctx->d_enc_meta)
    (void)hipFree(ctx->d_enc_meta);
  if (ctx->d_enc_ck)
    (void)hipFree(ctx->d_enc_ck);
  if (ctx->d_status)
    (void)hipFree(ctx->d_status);
  if (ctx->h_pin)
    (void)hipHostFree(ctx->h_pin);
  if (ctx->h_enc_result)
    (void)hipHostFree(ctx->h_enc_result);
  for (hipStream_t st : ctx->pipe_streams)
    if (st)
      (void)hipStreamDestroy(st);
  delete ctx;
}

const char *hsrans_ctx_device_name(const hsrans_ctx *ctx) { return ctx ? ctx->name : ""; }

uint32_t hsrans_ctx_host_index_chains(hsrans_ctx *ctx)
{
  if (ctx == nullptr)
    return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return ctx->host_index ? ctx->host_index->hdr.n_chains : 0;
}

// (Re)fills a device plan from a validated host plan blob: uploads it and prepares whatever the launch of this plan's
// kind needs (persistent arguments + host-built table, or the group list).  Device buffers are kept and grown, so a plan
// object can be refilled per call without allocations (the host-pointer entries do that).  The device must be current.

extern "C++" int dplan_fill(hsrans_dplan *d, const uint8_t *plan, size_t plan_size, const PlanHeader &h, hipStream_t s)
{
  hsrans_ctx *ctx = d->ctx;
  d->hdr = h;
  d->plan_bytes = plan_size;
  d->pa = PersistentArgs{};
  d->single = SingleArgs{};
  d->n_groups = 0;
  d->groups_lean = false;
  d->spread_min_block = 0;
  d->uid = next_dplan_uid(); // (a refill is another plan)
  d->deal_sig = 0;
  d->block_begin.clear();
  d->dealt_state = 0;
  {
    d->body_lo = 0;
    d->out_lo = 0;
    d->out_hi = h.decoded_len;
    if (!(h.flags & kPlanWalk)) // (a walk plan starts from the states at stream + 16: body_lo stays 0)
    {
      const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
      uint64_t lo = h.stream_len, olo = h.decoded_len, ohi = 0;
      for (uint32_t i = 0; i < h.n_pieces; i++)
      {
        const Piece &p = pc[i];
        const uint64_t len = (p.flags & kPieceFill) ? p.fill_len : (uint64_t)p.steps * h.states + p.tail;
        olo = std::min(olo, p.out_off);
        ohi = std::max(ohi, p.out_off + len);
        if (p.flags & kPieceFill)
          continue;
        lo = std::min(lo, p.words_off);
        if (!h.shared_hist || !(h.flags & kPlanHasHist)) // the kernel reads this histogram from the stream
          lo = std::min(lo, p.hist_off);
      }
      d->body_lo = lo;
      d->out_lo = std::min(olo, ohi);
      d->out_hi = ohi;
    }
  }
  // one allocation for everything this function uploads (sizes: upper bounds known from the header alone)
  const bool mergeable_raw = (h.flags & kPlanMergeable) && h.container == HSRANS_RAW;
  const bool may_group = !(h.flags & (kPlanWalk | kPlanMergeable)) && h.n_chains > 1;
  const size_t counter_bytes = (size_t)kCounterSets * kDynQueues * kDynQueueStride * 8;
  const bool need_counters = (mergeable_raw && h.interval != 0) || may_group; // (one-chain-per-wave plans draw nothing)
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t table_bound = mergeable_raw && (h.flags & kPlanHasHist) ? up256(std::max<size_t>((size_t)8 << h.bits, rank_table_entries(h.bits >= 13 ? h.bits : 13) * 8)) : 0;
  const size_t group_bound = may_group ? up256(((size_t)h.n_chains + 16) * sizeof(Group)) : 0;
  const size_t arena_need = 256 + (need_counters ? up256(counter_bytes) : 0) + up256(plan_size) + table_bound + group_bound;
  if (!grow(&d->d_arena, &d->d_arena_cap, arena_need))
    return HSRANS_E_HIP;
  d->arena_used = 0;
  auto carve = [&](size_t bytes) -> uint8_t * {
    uint8_t *ptr = d->d_arena + d->arena_used;
    d->arena_used += up256(bytes);
    return d->arena_used <= d->d_arena_cap ? ptr : nullptr;
  };
  d->d_status = (uint32_t *)carve(64);
  d->d_counters = need_counters ? (unsigned long long *)carve(counter_bytes) : nullptr;
  // status word and ticket counters start from zero on every (re)fill: "ticket mod draws-per-launch" only works while every
  // launch on a set of heads draws the same number of tickets, i.e. for ONE plan
  if (hipMemsetAsync(d->d_arena, 0, d->arena_used, s) != hipSuccess)
    return HSRANS_E_HIP;
  d->epoch.store(0, std::memory_order_relaxed);
  d->d_plan = carve(plan_size);
  d->d_plan_cap = plan_size;
  d->d_table = nullptr, d->d_table_cap = 0;
  d->d_groups = nullptr, d->d_groups_cap = 0;
  if (d->d_plan == nullptr || hipMemcpyAsync(d->d_plan, plan, plan_size, hipMemcpyHostToDevice, s) != hipSuccess)
    return HSRANS_E_HIP;
  std::vector<uint2> tab;     // (host copies of what is uploaded asynchronously: alive until the one synchronisation at the end)
  std::vector<Group> groups;
  auto fail = [&](int rc) { // (nothing queued above may still be reading `plan`, `tab` or `groups` when the caller sees the failure)
    (void)hipStreamSynchronize(s);
    return rc;
  };
  if (mergeable_raw)
  {
    // persistent launch arguments, taken from the plan once (hsrans_kernels.h PersistentArgs).  plan_validate has
    // re-derived what the flag promises: single-piece chains, back to back in output and stream, tail on the last only,
    // uniform `interval` (or interval == 0: chains of any length, decoded one per wave by the direct launch)
    const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
    const Piece &first = pc[0], &last = pc[h.n_pieces - 1];
    const uint64_t steps_total = (last.out_off - first.out_off) / h.states + last.steps;
    if (first.out_off > h.decoded_len || steps_total * h.states + last.tail > h.decoded_len - first.out_off)
      return fail(HSRANS_E_FORMAT);
    {
      uint64_t sig = 0x9E3779B97F4A7C15ull ^ ((uint64_t)h.states << 48) ^ ((uint64_t)h.bits << 40) ^ h.n_chains;
      for (uint32_t i = 0; i < h.n_pieces; i++)
        sig = (sig ^ pc[i].steps) * 0x100000001B3ull + (sig >> 29);
      d->deal_sig = sig ? sig : 1;
    }
    d->pa.pieces = (const Piece *)(d->d_plan + plan_pieces_off(h.n_chains));
    d->pa.states = (const uint32_t *)(d->d_plan + plan_states_off(h.n_chains, h.n_pieces));
    d->pa.n_chains = h.n_chains;
    d->pa.interval = h.interval;
    d->pa.S = h.states;
    d->pa.bits = h.bits;
    d->pa.out_base = first.out_off;
    d->pa.steps_total = steps_total;
    d->pa.hist_off = h.aux_off;
    d->pa.tail = last.tail;
    d->pa.counters = d->d_counters;
    if (h.flags & kPlanHasHist)
    {
      const uint16_t *counts = (const uint16_t *)(plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
      TableChoice tc = choose_table(h.bits, h.states, h.interval == 0);
      if (tc.dual)
      {
        // k_decode_dual reads the two neighbouring chains of a wave through ONE 32-bit window of the stream: a pair whose words
        // could span 4 GiB (a multi-GiB stream indexed for very few chains, e.g. by hsrans_plan_thin) goes one chain per wave
        for (uint32_t a = 0; a < h.n_chains && tc.dual; a += 2)
          if ((a + 2 < h.n_chains ? pc[a + 2].words_off : h.stream_len) - pc[a].words_off >= 0xFFFF0000ull)
            tc = choose_table(h.bits, h.states, false);
      }
      uint32_t mode = tc.mode;
      if (mode == 3 || mode == 5)
      {
        // decode table for the shared-table kernel (MODE 3): {freq | sym << 24, slot - cumul} per slot, the same
        // entries build_table<kModePack64> produces (hist.cpp:291-306 / :308-324 for the sum check)
        const uint32_t total = 1u << h.bits;
        tab.resize(total);
        uint32_t cum = 0;
        for (uint32_t sy = 0; sy < 256; sy++)
        {
          for (uint32_t k = 0; k < counts[sy] && cum + k < total; k++)
            tab[cum + k] = make_uint2((uint32_t)counts[sy] | (sy << 24), k);
          cum += counts[sy];
        }
        if (cum != total)
          return fail(HSRANS_E_FORMAT);
      }
      else if (mode == 4)
      {
        // wider histograms: the rank table (kModeRank: a byte per slot + 256 entries), 18 / 34 KiB at 14 / 15 bits instead of 128 / 256 KiB
        tab.resize(rank_table_entries(h.bits));
        if (build_rank_table(counts, h.bits, tab.data(), tab.size()) == 0)
          return fail(HSRANS_E_FORMAT);
      }
      d->pa.dual = tc.dual ? 1 : 0;
      if (mode != 0)
      {
        d->d_table = carve(tab.size() * sizeof(uint2));
        d->d_table_cap = tab.size() * sizeof(uint2);
        if (d->d_table == nullptr || hipMemcpyAsync(d->d_table, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice, s) != hipSuccess)
          return fail(HSRANS_E_HIP);
        d->pa.table = (const uint2 *)d->d_table;
        d->pa.table_mode = mode;
        d->pa.hist_copy = (const uint16_t *)(d->d_plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
      }
    }
  }
  if (h.shared_hist && d->pa.table == nullptr) // no host-built table: the kernel builds its own from the histogram in the stream
    d->body_lo = std::min(d->body_lo, h.aux_off);
  if (h.container == HSRANS_RAW && h.n_chains == 1 && h.n_pieces == 1 && !(h.flags & kPlanWalk) && h.bits <= 14)
  {
    // a raw stream without an index: one chain — the two-wave latency kernel (k_decode_single) instead of one wave of k_decode
    const Piece &p = *(const Piece *)(plan + plan_pieces_off(1));
    if (!(p.flags & kPieceFill))
    {
      d->single.valid = 1;
      d->single.steps = p.steps;
      d->single.tail = p.tail;
      d->single.S = h.states;
      d->single.bits = h.bits;
      d->single.ring_entries = h.bits <= 13 ? 2048 : 1024; // 14 bits: 128 KiB of table leave room for 1,088 ring entries
      d->single.hist_off = p.hist_off;
      d->single.words_off = p.words_off;
      d->single.out_off = p.out_off;
    }
  }
  if (!(h.flags & (kPlanWalk | kPlanMergeable)) && h.n_chains > 1)
  {
    // group consecutive chains that decode with the same histogram (= the chains of one block_/mt_ block)
    const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
    const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
    for (uint32_t ch = 0; ch < h.n_chains; ch++)
    {
      const Piece &p = pc[cf[ch]];
      const bool single = cf[ch + 1] - cf[ch] == 1;
      const bool fill = (p.flags & kPieceFill) != 0;
      bool joins = false;
      if (!groups.empty() && single)
      {
        Group &g = groups.back();
        const Piece &q = pc[cf[ch - 1]];
        if (fill && (g.flags & kGroupFill))
          joins = true;
        else if (!fill && !(g.flags & kGroupFill) && g.hist_off == p.hist_off)
        {
          joins = true;
          if (!(cf[ch] - cf[ch - 1] == 1 && q.tail == 0 && q.out_off + (uint64_t)q.steps * h.states == p.out_off && q.words_off <= p.words_off && p.state_idx == ch))
            g.flags &= ~kGroupMergeable;
        }
      }
      if (joins)
        groups.back().count++;
      else
      {
        Group g{};
        g.begin = ch;
        g.count = 1;
        g.flags = fill ? kGroupFill : (single && p.state_idx == ch ? kGroupMergeable : 0);
        g.piece0 = cf[ch];
        g.hist_off = fill ? 0 : p.hist_off;
        g.words_end = h.stream_len;
        // the previous rANS group's words end no later than this group's histogram / header
        if (!fill && !groups.empty())
          for (size_t k = groups.size(); k-- > 0 && groups[k].words_end == h.stream_len;)
            groups[k].words_end = p.hist_off;
        groups.push_back(g);
      }
    }
    // Few, large blocks would leave workgroup slots empty (one workgroup per group): mergeable groups are cut into parts of
    // >= kGroupPartChains chains while there are fewer groups than kGroupPartsPerCU per CU (hsrans_kernels.h: the rule and what was
    // measured).  A part is a group of its own: same histogram, a sub-range of the chains, and its words end where the next part's
    // first chain starts reading.
    // k_decode_spread (all chains dealt out over every resident wave, kernels_spread.h) wants single-piece chains — chain c is piece
    // c with states c — and no share of the chains touching three blocks: the launcher compares its longest share with the fewest
    // chains of a coded block that is not the last
    {
      bool ok = h.states == 64 && h.n_pieces == h.n_chains && groups.size() < h.n_chains;
      size_t last_coded = groups.size(), first_coded = groups.size();
      for (size_t k = groups.size(); k-- > 0 && last_coded == groups.size();)
        if (!(groups[k].flags & kGroupFill))
          last_coded = k;
      for (size_t k = 0; k < groups.size() && first_coded == groups.size(); k++)
        if (!(groups[k].flags & kGroupFill))
          first_coded = k;
      // (a share touches three coded blocks only when one lies wholly INSIDE it: neither the plan's first nor its last coded block can —
      // a slice of a plan, e.g. a rank's run of a sharded decode, usually begins and ends with part of a block)
      uint32_t fewest = 0xFFFFFFFFu;
      for (size_t k = 0; k < groups.size() && ok; k++)
      {
        const Group &g = groups[k];
        if (g.flags & kGroupFill)
          continue;
        ok = (g.flags & kGroupMergeable) && g.piece0 == g.begin;
        if (k != last_coded && k != first_coded)
          fewest = std::min(fewest, g.count);
      }
      d->spread_min_block = ok ? fewest : 0;
      // k_decode_dealt: the blocks as chain ranges, where every one of them is a coded block of such chains
      bool plain = ok && !groups.empty();
      for (size_t k = 0; k < groups.size() && plain; k++)
        plain = !(groups[k].flags & kGroupFill) && groups[k].begin == (k ? groups[k - 1].begin + groups[k - 1].count : 0);
      if (plain)
      {
        d->block_begin.reserve(groups.size() + 1);
        for (const Group &g : groups)
          d->block_begin.push_back(g.begin);
        d->block_begin.push_back(h.n_chains);
      }
    }
    const size_t want = (size_t)kGroupPartsPerCU * ctx->geom.num_cus;
    if (groups.size() < h.n_chains && groups.size() < want)
    {
      const uint32_t k_max = (uint32_t)((want + groups.size() - 1) / groups.size());
      std::vector<Group> parts;
      for (const Group &g : groups)
      {
        const uint32_t k = (g.flags & kGroupMergeable) ? group_parts_of(g.count, k_max) : 1;
        if (k < 2)
        {
          parts.push_back(g);
          continue;
        }
        for (uint32_t part = 0; part < k; part++)
        {
          Group q = g;
          const uint32_t lo = (uint32_t)((uint64_t)g.count * part / k), hi = (uint32_t)((uint64_t)g.count * (part + 1) / k);
          q.begin = g.begin + lo;
          q.piece0 = g.piece0 + lo;
          q.count = hi - lo;
          if (part + 1 < k)
            q.words_end = pc[cf[g.begin + hi]].words_off;
          parts.push_back(q);
        }
      }
      groups.swap(parts);
    }
    // (Dynamic group order, run_grouped: the END of the list decides how evenly the launch finishes.  Cutting the last eighth /
    // quarter of the list into half-blocks was built and measured in round 3 at 2^30 bytes — 0.454-0.458 ms against 0.451-0.456
    // without: the extra table builds cost what the evener finish gains — and is gone.)
    if (!d->part_ends.empty() && d->part_ends.size() <= kMaxLaunchParts)
    {
      // the sub-runs of a sharded decode (hsrans_comm.cpp): which of them a group overlaps, and how many groups each will be counted by
      const std::vector<uint32_t> &ends = d->part_ends;
      auto part_of = [&](uint32_t chain) { return (uint32_t)(std::upper_bound(ends.begin(), ends.end(), chain) - ends.begin()); };
      d->part_units.assign(ends.size(), 0);
      d->part_cum.assign(ends.size(), 0);
      for (Group &g : groups)
      {
        const uint32_t lo = std::min<uint32_t>(part_of(g.begin), (uint32_t)ends.size() - 1), hi = std::min<uint32_t>(part_of(g.begin + g.count - 1), (uint32_t)ends.size() - 1);
        g.flags = (g.flags & ((1u << kGroupPartShift) - 1)) | (lo << kGroupPartShift) | (hi << (kGroupPartShift + 8));
        for (uint32_t p = lo; p <= hi; p++)
          d->part_units[p]++;
      }
    }
    if (groups.size() < h.n_chains)
    {
      d->d_groups = carve(groups.size() * sizeof(Group)); // (the dynamic group order's ticket counters: d_counters, zeroed above)
      d->d_groups_cap = groups.size() * sizeof(Group);
      if (d->d_groups == nullptr || hipMemcpyAsync(d->d_groups, groups.data(), groups.size() * sizeof(Group), hipMemcpyHostToDevice, s) != hipSuccess)
        return fail(HSRANS_E_HIP);
      d->n_groups = (uint32_t)groups.size();
      d->groups_lean = h.states == 64;
      for (const Group &g : groups)
        if (!(g.flags & (kGroupMergeable | kGroupFill)))
          d->groups_lean = false;
    }
  }
  // `tab` and `groups` are about to go away: everything queued above has to have left them
  if (hipStreamSynchronize(s) != hipSuccess)
    return HSRANS_E_HIP;
  return HSRANS_OK;
}

// Plans written on the device (the GPU encoder's, an indexing decode's) have their group list there: a block's parts are consecutive groups of one
// histogram.  One small copy (32 bytes a group) at plan creation gives k_decode_dealt's dealing the blocks as chain ranges.
extern "C++" void dplan_blocks_from_device_groups(hsrans_dplan *d, hipStream_t s)
{
  d->block_begin.clear();
  d->dealt_state = 0;
  if (d->n_groups == 0 || d->d_groups == nullptr || !d->groups_lean || d->hdr.n_pieces != d->hdr.n_chains)
    return;
  try
  {
    std::vector<Group> groups(d->n_groups);
    if (hipMemcpyAsync(groups.data(), d->d_groups, groups.size() * sizeof(Group), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    {
      (void)hipGetLastError();
      return;
    }
    std::vector<uint32_t> begins;
    uint32_t next = 0;
    for (size_t k = 0; k < groups.size(); k++)
    {
      const Group &g = groups[k];
      if (g.count == 0) // (device builders leave unused part slots empty)
        continue;
      if ((g.flags & kGroupFill) || !(g.flags & kGroupMergeable) || g.begin != next || g.piece0 != g.begin)
      {
        if (getenv("HSRANS_DEALT_TRACE"))
          fprintf(stderr, "hsrans dealt: group %zu of %zu: flags %x begin %u count %u piece0 %u, expected begin %u\n", k, groups.size(), g.flags, g.begin, g.count, g.piece0, next);
        return;
      }
      if (begins.empty() || g.hist_off != groups[k - 1].hist_off || groups[k - 1].count == 0)
      {
        // (a part continues its block when the previous non-empty group has the same histogram)
        bool cont = false;
        for (size_t j = k; j-- > 0;)
          if (groups[j].count != 0)
          {
            cont = groups[j].hist_off == g.hist_off;
            break;
          }
        if (!cont)
          begins.push_back(g.begin);
      }
      next = g.begin + g.count;
    }
    if (next != d->hdr.n_chains || begins.empty())
      return;
    begins.push_back(d->hdr.n_chains);
    d->block_begin.swap(begins);
  }
  catch (...)
  {
    d->block_begin.clear();
  }
}

// A page-locked, device-mapped host range (hipHostMalloc / hipHostRegister): the address the GPU reaches it at, else null.
extern "C++" uint8_t *device_view_of_host(const void *ptr, size_t bytes)
{
  if (ptr == nullptr || bytes == 0)
    return nullptr;
  hipPointerAttribute_t a{}, b{};
  if (hipPointerGetAttributes(&a, ptr) != hipSuccess || hipPointerGetAttributes(&b, (const uint8_t *)ptr + bytes - 1) != hipSuccess)
  {
    (void)hipGetLastError(); // pageable memory: not an error of ours
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost || b.type != hipMemoryTypeHost || a.devicePointer == nullptr || b.devicePointer == nullptr)
    return nullptr;
  if ((const uint8_t *)b.devicePointer - (const uint8_t *)a.devicePointer != (ptrdiff_t)(bytes - 1)) // one mapping, end to end
    return nullptr;
  return (uint8_t *)a.devicePointer;
}

// one launch of a filled device plan (asynchronous on s; the device must be current)
extern "C++" int dplan_launch(hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, hipStream_t s, uint64_t stream_lo,
                              const PartArgs *part_words)
{
  KParams kp{};
  kp.stream = (const uint8_t *)d_stream;
  kp.stream_len = stream_length;
  kp.stream_lo = stream_lo;
  kp.out = (uint8_t *)d_out;
  kp.out_cap = out_capacity;
  kp.plan = d->d_plan;
  kp.status = d->d_status;
  kp.stamps = d->d_stamps;
  kp.finish = d->d_finish;
  kp.pa = d->pa;
  kp.single = d->single;
  kp.single_states = (const uint32_t *)(d->d_plan + plan_states_off(d->hdr.n_chains, d->hdr.n_pieces));
  if (kp.pa.counters != nullptr) // uniform persistent launch: its own set of queue heads
    kp.pa.counters += (size_t)(d->epoch.fetch_add(1, std::memory_order_relaxed) % kCounterSets) * kDynQueues * kDynQueueStride;
  if (d->n_groups)
  {
    kp.groups = (const Group *)d->d_groups;
    kp.n_groups = d->n_groups;
    kp.groups_lean = d->groups_lean ? 1 : 0;
    kp.spread = d->groups_lean ? d->spread_min_block : 0;
    // (measured at 2^30 bytes, two runs each on one box: 0 -> 0.447-0.450 ms, 300 -> 0.440, 500 -> 0.440-0.445, 700 -> 0.447-0.451, 1000 -> 0.452-0.455)
    kp.group_prio = getenv("HSRANS_GROUP_PRIO") != nullptr ? (uint32_t)atoi(getenv("HSRANS_GROUP_PRIO")) : 350;
    // HSRANS_GROUP_PRIO_CLASS: ten per-mille values, see KParams::group_prio_class (tuning; tools/group_prio_probe.py)
    kp.group_prio_class[9] = 0xFFFF;
    if (const char *e = getenv("HSRANS_GROUP_PRIO_CLASS"))
    {
      uint32_t v[10], n = 0;
      for (const char *p = e; n < 10 && *p; n++)
      {
        v[n] = (uint32_t)strtoul(p, (char **)&p, 10);
        if (*p == ',')
          p++;
      }
      if (n == 10)
        for (uint32_t k = 0; k < 10; k++)
          kp.group_prio_class[k] = (uint16_t)(v[k] > 1000 ? 1000 : v[k]);
    }
    // (requesting a round's records and first chunks before its table build: measured, no gain — the other workgroups of the CU
    // fill the gap either way — so off unless asked for)
    // dynamic group order: this launch's own ticket counter (the counter sets of the persistent launches, one head of each used)
    if (d->d_counters != nullptr && getenv("HSRANS_GROUP_STATIC") == nullptr)
      kp.group_tickets = d->d_counters + (size_t)(d->epoch.fetch_add(1, std::memory_order_relaxed) % kCounterSets) * kDynQueues * kDynQueueStride;
  }
  // lean grouped plans of coded blocks: the host-dealt one-round launch where the plan suits it (dealt once per weight set)
  const DealtTable *dealt = nullptr;
  if (d->n_groups && d->groups_lean && d->block_begin.size() >= 2 && (d->hdr.bits <= 11 || d->hdr.bits == 13 || d->hdr.bits == 14) && d->hdr.states == 64 &&
      !(d->hdr.bits >= 13 && getenv("HSRANS_DEALT_WIDE") != nullptr && atoi(getenv("HSRANS_DEALT_WIDE")) == 0)) // (HSRANS_DEALT_WIDE=0: 13 / 14 bits keep the grouped launch: comparison)
  {
    uint32_t w8[8];
    const uint64_t total_groups = (d->out_hi - d->out_lo) / 64; // (what THIS plan's chains decode: a rank's slice of a sharded stream, not the stream)
    dealt_weights_now(d->ctx->geom, total_groups / ((uint64_t)spread_grid(d->ctx->geom) * 16), d->hdr.bits, w8);
    if (d->dealt_state == 0 || memcmp(w8, d->dealt_weights, sizeof(w8)) != 0)
      d->dealt_state = deal_shares(d->ctx->geom, d->block_begin.data(), (uint32_t)d->block_begin.size() - 1, d->hdr.n_chains, total_groups, d->hdr.bits, &d->dealt, d->dealt_weights) ? 1 : -1;
    if (d->dealt_state == 1)
      dealt = &d->dealt;
  }
  if (getenv("HSRANS_DEALT_TRACE"))
    fprintf(stderr, "hsrans dealt: groups %u lean %d blocks %zu bits %u chains %u state %d\n", d->n_groups, (int)d->groups_lean, d->block_begin.size(), d->hdr.bits, d->hdr.n_chains, d->dealt_state);
  if (part_words != nullptr)
  {
    // a rank's sub-runs in one launch: the caller's completion words and sequence number, this plan's parts and running totals
    if (d->part_ends.empty() || d->part_units.size() != d->part_ends.size() || d->n_groups == 0)
      return HSRANS_E_ARG;
    kp.parts = *part_words;
    PartPlan pp{(uint32_t)d->part_ends.size(), d->part_ends.data(), d->part_units.data(), d->part_cum.data()};
    return launch_decode(kp, d->hdr, d->ctx->geom, s, &d->info, &pp, dealt, d->dealt_weights) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
  }
  return launch_decode(kp, d->hdr, d->ctx->geom, s, &d->info, nullptr, dealt, d->dealt_weights) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
}


size_t hsrans_decode_host(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out, size_t out_capacity,
                          const uint8_t *plan, size_t plan_size)
try
{
  if (ctx == nullptr || in == nullptr || out == nullptr || !valid_codec(container, states, bits))
    return 0;

  // A caller that loops the plain decodeFunc over one file (the reference's benchmark does: src/main.cpp:860-889) pays for the
  // missing index once: the first call's decode records checkpoints (hsrans_decode_device_indexing) and the plan it leaves is kept
  // in the context; later calls on the same bytes launch it.  "The same bytes" is checked on ALL of them: the stream is uploaded
  // anyway, a wide kernel fingerprints it there, and the decode that ran beside it only counts when the fingerprint matches.
  const bool cacheable = plan == nullptr && (container == HSRANS_MT || container == HSRANS_RAW) && in_length >= 16 && getenv("HSRANS_HOST_INDEX_CACHE_OFF") == nullptr;
  const uint64_t codec_key = (uint64_t)container | ((uint64_t)states << 8) | ((uint64_t)bits << 16) | (1ull << 32);
  if (cacheable)
  {
    std::lock_guard<std::mutex> guard(ctx->lock);
    hsrans_dplan *ix = ctx->host_index;
    // (the stream's first bytes — its header, the first block header or the histogram — are compared on the host before anything is
    // launched; the fingerprint of ALL bytes is what the result then rests on)
    if (ix != nullptr && ctx->host_index_key[0] == (uint64_t)(uintptr_t)in && ctx->host_index_key[1] == in_length && ctx->host_index_key[2] == codec_key &&
        ctx->host_index_head_len == std::min<size_t>(in_length, sizeof(ctx->host_index_head)) && memcmp(ctx->host_index_head, in, ctx->host_index_head_len) == 0 &&
        ix->hdr.decoded_len <= out_capacity && hipSetDevice(ctx->device) == hipSuccess)
    {
      const size_t n = (size_t)ix->hdr.decoded_len;
      const size_t in_pad = (in_length + 15) / 16 * 16;
      if (grow(&ctx->d_in, &ctx->d_in_cap, in_pad) && grow(&ctx->d_out, &ctx->d_out_cap, n + 16) && grow(&ctx->d_enc_meta, &ctx->d_enc_meta_cap, 64))
      {
        hipStream_t s = ctx->stream;
        uint32_t status = 0xFFFFFFFF;
        uint64_t sum = 0;
        bool ok = hipMemcpyAsync(ctx->d_in, in, in_length, hipMemcpyHostToDevice, s) == hipSuccess &&
                  launch_stream_checksum(ctx->d_in, in_length, (uint64_t *)ctx->d_enc_meta, s) == hipSuccess && hipMemsetAsync(ix->d_status, 0, 4, s) == hipSuccess &&
                  dplan_launch(ix, ctx->d_in, in_length, ctx->d_out, n, s) == HSRANS_OK && hipMemcpyAsync(out, ctx->d_out, n, hipMemcpyDeviceToHost, s) == hipSuccess &&
                  hipMemcpyAsync(&sum, ctx->d_enc_meta, 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                  hipMemcpyAsync(&status, ix->d_status, 4, hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
        if (ok && status == 0 && sum == ctx->host_index_key[3])
          return n;
        (void)hipGetLastError();
      }
      // other bytes at that address (or a failure): the index is dropped and the call starts over below (`out` is rewritten in full)
      hsrans_dplan_destroy(ctx->host_index);
      ctx->host_index = nullptr;
    }
  }

  std::vector<uint8_t> own_plan;
  if (plan == nullptr)
  {
    // header-only peek to size the plan, then the real planner (which repeats the reference's entry checks)
    if (in_length < 16)
      return 0;
    uint64_t out_len;
    memcpy(&out_len, in, 8);
    if (out_len > out_capacity)
      return 0;
    if (!plan_build_vec(container, states, bits, in, in_length, out_capacity, &own_plan)) // sized by the stream's own chain count
      return 0;
    plan = own_plan.data();
    plan_size = own_plan.size();
  }
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || (int)h.container != container || (int)h.states != states || h.bits != bits)
    return 0;
  if (!plan_validate(plan, plan_size, in_length, out_capacity))
    return 0;

  std::lock_guard<std::mutex> guard(ctx->lock);
  if (hipSetDevice(ctx->device) != hipSuccess)
    return 0;
  const size_t in_pad = (in_length + 15) / 16 * 16;
  // (Storing straight into a page-locked `out`, as hsrans_hpipe_decode does, was measured here too: with nothing to overlap it
  // only replaces a download copy at 55 GB/s by the kernel's own PCIe writes at 47 — 100 MB: 3.58 instead of 3.26 ms — so this
  // one-shot entry stages its output and copies it down (the knob that switched the direct stores on here is gone: settled).)
  uint8_t *out_view = nullptr;
  if (!grow(&ctx->d_in, &ctx->d_in_cap, in_pad) || (out_view == nullptr && !grow(&ctx->d_out, &ctx->d_out_cap, (size_t)h.decoded_len + 16)))
    return 0;
  hipStream_t s = ctx->stream;
  if (ctx->host_dplan == nullptr)
  {
    ctx->host_dplan = new (std::nothrow) hsrans_dplan;
    if (ctx->host_dplan == nullptr)
      return 0;
    ctx->host_dplan->ctx = ctx;
  }
  hsrans_dplan *d = ctx->host_dplan;
  // the same launch the device entry gets for this plan (persistent / direct / grouped), on the context's staging buffers
  uint32_t status = 0xFFFFFFFF;
  bool ok = hipMemcpyAsync(ctx->d_in, in, in_length, hipMemcpyHostToDevice, s) == hipSuccess && dplan_fill(d, plan, plan_size, h, s) == HSRANS_OK &&
            hipMemsetAsync(d->d_status, 0, 4, s) == hipSuccess;
  // the first decode of a stream without an index records one for the next call (see the top of the function); streams whose base
  // plan has a single short chain, or any failure of the recording path, take the plain launch
  bool indexed_now = false;
  uint64_t sum = 0;
  if (ok && cacheable && out_view == nullptr && h.interval == 0 && h.decoded_len >= (1u << 20) && grow(&ctx->d_enc_meta, &ctx->d_enc_meta_cap, 64))
  {
    if (ctx->host_index)
      hsrans_dplan_destroy(ctx->host_index);
    ctx->host_index = nullptr;
    d->hdr = h;
    d->plan_bytes = plan_size;
    hsrans_dplan *ix = nullptr;
    bool have_index = launch_stream_checksum(ctx->d_in, in_length, (uint64_t *)ctx->d_enc_meta, s) == hipSuccess &&
                      hipMemcpyAsync(&sum, ctx->d_enc_meta, 8, hipMemcpyDeviceToHost, s) == hipSuccess;
    // HSRANS_HIP_STRICT=1: nothing of a `*_decode_hip_N` call runs on a host core — the raw stream's checkpoints are recorded by the one
    // wavefront that decodes it (hsrans_decode_device_indexing: ~125 ms for 100 MB, once) instead of by the host SIMD decoder's pass (~35 ms)
    const char *e_strict = getenv("HSRANS_HIP_STRICT");
    const bool strict = e_strict != nullptr && e_strict[0] != '\0' && e_strict[0] != '0';
    if (have_index && container == HSRANS_RAW && !strict)
    try
    {
      // a raw stream is ONE chain: the pass that records its checkpoints is the host SIMD decoder's (2-4 GB/s on one core, while the
      // upload is on its way; one wavefront would need four times as long), at the one-chain-per-wavefront boundaries of this device;
      // the decode itself is the indexed GPU launch
      std::vector<uint64_t> groups(2 * 8192 + 64);
      const size_t ng = hsrans_index_boundaries(ctx, states, bits, (size_t)h.decoded_len, groups.data(), groups.size());
      std::vector<uint8_t> iplan(ng ? plan_capacity_chains(HSRANS_RAW, states, (size_t)h.decoded_len, ng, 0) : 0);
      const size_t plen = ng ? cpu::index_build(cpu::best_level(), 1, HSRANS_RAW, states, bits, in, in_length, groups.data(), ng, iplan.data(), iplan.size()) : 0;
      have_index = plen != 0 && hsrans_dplan_create(ctx, iplan.data(), plen, &ix) == HSRANS_OK;
      if (have_index)
      {
        have_index = hipMemsetAsync(ix->d_status, 0, 4, s) == hipSuccess && dplan_launch(ix, ctx->d_in, in_length, ctx->d_out, (size_t)h.decoded_len, s) == HSRANS_OK &&
                     hipMemcpyAsync(&status, ix->d_status, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && status == 0;
        status = 0xFFFFFFFF; // (the entry's own status word is read below)
        if (!have_index)
        {
          (void)hipStreamSynchronize(s);
          hsrans_dplan_destroy(ix);
          ix = nullptr;
        }
      }
    }
    catch (...) // (bad_alloc from the vectors above: the upload of `in` and the fingerprint's copy are already queued — nothing queued
    {           // may still read `in` or write this frame when the function returns)
      (void)hipStreamSynchronize(s);
      if (ix != nullptr)
        hsrans_dplan_destroy(ix);
      return 0;
    }
    else if (have_index)
      have_index = decode_device_indexing_impl(ctx, d, ctx->d_in, in_length, ctx->d_out, (size_t)h.decoded_len, 64, s, &ix, true) == HSRANS_OK;
    if (have_index)
    {
      indexed_now = true; // (the recording pass has decoded into d_out and was synchronised: `sum` has arrived too)
      ctx->host_index = ix;
      ctx->host_index_key[0] = (uint64_t)(uintptr_t)in;
      ctx->host_index_key[1] = in_length;
      ctx->host_index_key[2] = codec_key;
      ctx->host_index_key[3] = sum;
      ctx->host_index_head_len = (uint32_t)std::min<size_t>(in_length, sizeof(ctx->host_index_head));
      memcpy(ctx->host_index_head, in, ctx->host_index_head_len);
    }
    else
      (void)hipGetLastError();
  }
  if (!indexed_now)
    ok = ok && dplan_launch(d, ctx->d_in, in_length, out_view ? out_view : ctx->d_out, (size_t)h.decoded_len, s) == HSRANS_OK;
  if (ok && out_view == nullptr)
    ok = hipMemcpyAsync(out, ctx->d_out, (size_t)h.decoded_len, hipMemcpyDeviceToHost, s) == hipSuccess;
  ok = ok && hipMemcpyAsync(&status, d->d_status, 4, hipMemcpyDeviceToHost, s) == hipSuccess;
  // nothing queued may still read `in` / `own_plan` or write `out` when this returns, whether or not a call above failed
  ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
  return ok && status == 0 ? (size_t)h.decoded_len : 0;
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

int hsrans_dplan_create(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, hsrans_dplan **out_dplan)
try
{
  if (ctx == nullptr || out_dplan == nullptr)
    return HSRANS_E_ARG;
  *out_dplan = nullptr;
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || !plan_validate(plan, plan_size, h.stream_len, h.decoded_len))
    return HSRANS_E_FORMAT;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_dplan *d = new (std::nothrow) hsrans_dplan;
  if (d == nullptr)
    return HSRANS_E_HIP;
  d->ctx = ctx;
  int rc = dplan_fill(d, plan, plan_size, h, nullptr);
  if (rc == HSRANS_OK && hipStreamSynchronize(nullptr) != hipSuccess)
    rc = HSRANS_E_HIP;
  if (rc != HSRANS_OK)
  {
    hsrans_dplan_destroy(d);
    return rc;
  }
  if (getenv("HSRANS_DEBUG_STAMPS") && hipMalloc((void **)&d->d_stamps, kStampWaves * 8 * 8) == hipSuccess)
    (void)hipMemset(d->d_stamps, 0, kStampWaves * 8 * 8);
  *out_dplan = d;
  return HSRANS_OK;
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

// hsrans_dplan_create for a plan whose chains [part_ends[k - 1], part_ends[k]) are the sub-runs of a sharded decode (hsrans_comm.cpp): the
// group list is tagged with them.  *out_dplan's part_units is empty when the plan is of a kind no one-launch kernel takes.
extern "C++" int dplan_create_with_parts(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, const std::vector<uint32_t> &part_ends, hsrans_dplan **out_dplan)
{
  *out_dplan = nullptr;
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || !plan_validate(plan, plan_size, h.stream_len, h.decoded_len))
    return HSRANS_E_FORMAT;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_dplan *d = new (std::nothrow) hsrans_dplan;
  if (d == nullptr)
    return HSRANS_E_HIP;
  d->ctx = ctx;
  d->part_ends = part_ends;
  int rc = dplan_fill(d, plan, plan_size, h, nullptr);
  if (rc == HSRANS_OK && hipStreamSynchronize(nullptr) != hipSuccess)
    rc = HSRANS_E_HIP;
  if (rc != HSRANS_OK)
  {
    hsrans_dplan_destroy(d);
    return rc;
  }
  *out_dplan = d;
  return HSRANS_OK;
}

size_t hsrans_debug_read_stamps(hsrans_dplan *d, uint64_t *out, size_t capacity_u64)
{
  if (d == nullptr || d->d_stamps == nullptr || out == nullptr)
    return 0;
  const size_t n = capacity_u64 < kStampWaves * 8 ? capacity_u64 : kStampWaves * 8;
  return hipMemcpy(out, d->d_stamps, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? n : 0;
}

int hsrans_dplan_create_from_device_stream(hsrans_ctx *ctx, int container, int states, uint32_t bits, const void *d_stream, size_t stream_length,
                                           size_t out_capacity, void *hip_stream, hsrans_dplan **out_dplan)
{
  // K2: the mt_ header chain is followed on the device, so a stream that only exists in HBM can be
  // planned without a host copy.  Pass 1 is a pointer chase by one wavefront (one 16-byte read per block: about one memory
  // round trip each) that lists the blocks; pass 2 writes the plan, one wavefront per block.
  if (ctx == nullptr || out_dplan == nullptr || d_stream == nullptr)
    return HSRANS_E_ARG;
  *out_dplan = nullptr;
  if (container != HSRANS_MT || !valid_codec(container, states, bits) || ((uintptr_t)d_stream & 15) != 0)
    return HSRANS_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hipStream_t s = (hipStream_t)hip_stream;
  WalkResult *d_res = nullptr;
  uint64_t *d_blocks = nullptr;
  WalkResult res{};
  hsrans_dplan *d = nullptr;
  int rc = HSRANS_E_HIP;
  do
  {
    if (hipMalloc((void **)&d_res, sizeof(WalkResult)) != hipSuccess)
      break;
    // block list: sized for blocks of >= 4 KiB on average, enlarged (up to one entry per 8 stream bytes, the smallest
    // block there is) when the chase reports that it ran out
    uint64_t max_blocks = out_capacity / 4096 + 4096;
    const uint64_t hard_max = std::min<uint64_t>(stream_length / 8 + 1, 0xFFFFFFFFull);
    bool chased = false;
    while (true)
    {
      max_blocks = std::min(max_blocks, hard_max);
      if (d_blocks)
        (void)hipFree(d_blocks);
      d_blocks = nullptr;
      if (hipMalloc((void **)&d_blocks, max_blocks * 16) != hipSuccess)
        break;
      if (launch_mt_chase((const uint8_t *)d_stream, stream_length, out_capacity, (uint32_t)states, d_blocks, (uint32_t)max_blocks, d_res, s) != hipSuccess ||
          hipMemcpyAsync(&res, d_res, sizeof(res), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        break;
      if (res.error == 7 && max_blocks < hard_max)
      {
        max_blocks *= 8;
        continue;
      }
      chased = true;
      break;
    }
    if (!chased)
      break;
    if (res.error != 0 || res.n_chains == 0)
    {
      rc = HSRANS_E_FORMAT;
      break;
    }
    d = new (std::nothrow) hsrans_dplan;
    if (d == nullptr)
      break;
    d->ctx = ctx;
    PlanHeader h{};
    memcpy(h.magic, "HSRPLAN1", 8);
    h.container = HSRANS_MT;
    h.states = (uint32_t)states;
    h.bits = bits;
    h.decoded_len = res.decoded_len;
    h.stream_len = stream_length;
    h.n_chains = h.n_pieces = res.n_chains;
    const size_t bytes = (size_t)plan_size(h.n_chains, h.n_pieces, h.states, 0);
    WalkResult res2{};
    if (hipMalloc((void **)&d->d_plan, bytes) != hipSuccess || hipMalloc((void **)&d->d_status, 64) != hipSuccess ||
        hipMemsetAsync(d->d_plan, 0, bytes, s) != hipSuccess || hipMemsetAsync(d->d_status, 0, 64, s) != hipSuccess ||
        hipMemcpyAsync(d->d_plan, &h, sizeof(h), hipMemcpyHostToDevice, s) != hipSuccess ||
        launch_mt_fill((const uint8_t *)d_stream, stream_length, (uint32_t)states, bits, d_blocks, d->d_plan, h.n_chains, res.decoded_len, d_res, s) != hipSuccess ||
        hipMemcpyAsync(&res2, d_res, sizeof(res2), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      break;
    if (res2.error != 0)
    {
      rc = HSRANS_E_FORMAT;
      break;
    }
    d->hdr = h;
    d->plan_bytes = bytes;
    d->out_hi = h.decoded_len; // a plan written on the device covers the whole stream and the whole output
    rc = HSRANS_OK;
  } while (false);
  if (d_res)
    (void)hipFree(d_res);
  if (d_blocks)
    (void)hipFree(d_blocks);
  if (rc != HSRANS_OK)
  {
    hsrans_dplan_destroy(d);
    return rc;
  }
  *out_dplan = d;
  return HSRANS_OK;
}

size_t hsrans_dplan_read_plan(hsrans_dplan *d, uint8_t *out, size_t capacity)
{
  if (d == nullptr || out == nullptr || d->d_plan == nullptr || d->plan_bytes == 0 || capacity < d->plan_bytes)
    return 0;
  return hipMemcpy(out, d->d_plan, d->plan_bytes, hipMemcpyDeviceToHost) == hipSuccess ? d->plan_bytes : 0;
}

void hsrans_dplan_destroy(hsrans_dplan *d)
{
  if (d == nullptr)
    return;
  if (d->d_stamps)
    (void)hipFree(d->d_stamps);
  if (d->d_arena) // (status, counters, plan, table and groups live inside it)
    (void)hipFree(d->d_arena);
  else
  {
    if (d->d_counters)
      (void)hipFree(d->d_counters);
    if (d->d_table)
      (void)hipFree(d->d_table);
    if (d->d_groups)
      (void)hipFree(d->d_groups);
    if (d->d_plan)
      (void)hipFree(d->d_plan);
    if (d->d_status)
      (void)hipFree(d->d_status);
  }
  delete d;
}

int hsrans_decode_device(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, void *hip_stream)
{
  if (ctx == nullptr || d == nullptr || d_stream == nullptr || d_out == nullptr || d->ctx != ctx)
    return HSRANS_E_ARG;
  if (((uintptr_t)d_stream & 15) != 0 || ((uintptr_t)d_out & 3) != 0)
    return HSRANS_E_ARG;
  if (stream_length < d->hdr.stream_len || out_capacity < d->hdr.decoded_len)
    return HSRANS_E_FORMAT;
  if (hipSetDevice(ctx->device) != hipSuccess) // the launch goes to the context's device whatever the caller's current device is
    return HSRANS_E_HIP;
  // the status word is sticky: kernels only ever OR error bits into it and hsrans_dplan_status() clears it after
  // reporting, so the launch path is exactly one kernel node (no memset node in front of it)
  return dplan_launch(d, d_stream, stream_length, d_out, out_capacity, (hipStream_t)hip_stream);
}

// the general launch: stream bytes [window_offset, +window_length) at d_window, output bytes [out_offset, +out_length) at d_out
extern "C++" int dplan_launch_ranges(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out, size_t out_offset,
                                     size_t out_length, void *hip_stream, const PartArgs *part_words)
{
  if (ctx == nullptr || d == nullptr || d_window == nullptr || d_out == nullptr || d->ctx != ctx)
    return HSRANS_E_ARG;
  if (((uintptr_t)d_window & 15) != 0 || (window_offset & 15) != 0 || ((uintptr_t)d_out & 3) != 0 || (out_offset & 3) != 0 ||
      (uintptr_t)d_window < window_offset || (uintptr_t)d_out < out_offset)
    return HSRANS_E_ARG;
  if (window_offset > d->hdr.stream_len || out_offset > d->hdr.decoded_len)
    return HSRANS_E_FORMAT;
  // Every byte the plan's chains read or write must be inside what the caller holds.  Lower edges: from the plan (dplan_fill
  // recorded them).  Upper edge of the stream: the kernels' buffer descriptors end at min(window end, chain's last word), so a
  // request past the window is dropped by the hardware; upper edge of the output: checked here.
  if (window_offset > d->body_lo || d->out_lo < out_offset || d->out_hi - out_offset > out_length)
    return HSRANS_E_FORMAT;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  const uint64_t end = std::min<uint64_t>((uint64_t)window_offset + window_length, d->hdr.stream_len);
  // (the kernels' own bound on the output is the end of the caller's window, not of the whole output: a path that rounded a store
  // up past a chain's end must not reach past a rank's smaller buffer either)
  const uint64_t out_end = std::min<uint64_t>((uint64_t)out_offset + out_length, d->hdr.decoded_len);
  return dplan_launch(d, (const uint8_t *)d_window - window_offset, (size_t)end, (uint8_t *)d_out - out_offset, (size_t)out_end, (hipStream_t)hip_stream, window_offset, part_words);
}

int hsrans_decode_device_window(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out,
                                size_t out_capacity, void *hip_stream)
{
  if (d == nullptr || out_capacity < d->hdr.decoded_len)
    return d == nullptr ? HSRANS_E_ARG : HSRANS_E_FORMAT;
  return dplan_launch_ranges(ctx, d, d_window, window_offset, window_length, d_out, 0, out_capacity, hip_stream, nullptr);
}

int hsrans_decode_device_ranges(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out_window,
                                size_t out_offset, size_t out_length, void *hip_stream)
{
  return dplan_launch_ranges(ctx, d, d_window, window_offset, window_length, d_out_window, out_offset, out_length, hip_stream, nullptr);
}

int hsrans_dplan_status(hsrans_ctx *ctx, hsrans_dplan *d, void *hip_stream)
{
  if (ctx == nullptr || d == nullptr)
    return HSRANS_E_ARG;
  uint32_t status = 0xFFFFFFFF;
  hipStream_t s = (hipStream_t)hip_stream;
  if (hipMemcpyAsync(&status, d->d_status, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return HSRANS_E_HIP;
  if (status == 0)
    return HSRANS_OK;
  if (hipMemsetAsync(d->d_status, 0, 4, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return HSRANS_E_HIP;
  return HSRANS_E_DEVICE;
}

int hsrans_dplan_launch_info(const hsrans_dplan *d, hsrans_launch_info *info)
{
  if (d == nullptr || info == nullptr)
    return HSRANS_E_ARG;
  info->grid = d->info.grid;
  info->block = d->info.block;
  info->lds_bytes = d->info.lds_bytes;
  info->waves_per_block = d->info.waves_per_block;
  info->chains = d->info.chains;
  info->shared_table = d->info.shared_table;
  info->walk = d->info.walk;
  info->two_level = d->info.two_level;
  info->table_mode = d->info.table_mode;
  info->chains_per_wave = d->info.chains_per_wave;
  for (int k = 0; k < 8; k++)
    info->class_weights[k] = d->info.class_weights[k];
  info->dynamic_groups = d->info.dynamic_groups;
  info->spread = d->info.spread;
  return HSRANS_OK;
}

int hsrans_dealt_shares(const hsrans_ctx *ctx, uint32_t bits, const uint32_t *block_begin, uint32_t n_blocks, uint32_t n_chains, uint64_t total_groups,
                        uint32_t *begin_out, uint16_t *split_out)
try
{
  if (block_begin == nullptr || begin_out == nullptr || split_out == nullptr || n_blocks == 0 || bits < 10 || bits > 15)
    return -1;
  for (uint32_t k = 0; k < n_blocks; k++)
    if (block_begin[k] >= block_begin[k + 1])
      return -1;
  DealtTable dt{};
  uint32_t w8[8];
  const DeviceGeom dg = ctx ? ctx->geom : default_geom();
  const bool ok = deal_shares(dg, block_begin, n_blocks, n_chains, total_groups, bits, &dt, w8);
  memcpy(begin_out, dt.begin, sizeof(dt.begin));
  memcpy(split_out, dt.split, sizeof(dt.split));
  return ok ? 1 : 0;
}
catch (...)
{
  return -1;
}

int hsrans_host_register(hsrans_ctx *ctx, void *ptr, size_t bytes)
{
  if (ctx == nullptr || ptr == nullptr || bytes == 0)
    return HSRANS_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
}

int hsrans_host_unregister(hsrans_ctx *ctx, void *ptr)
{
  if (ctx == nullptr || ptr == nullptr)
    return HSRANS_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  return hipHostUnregister(ptr) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
}

size_t hsrans_index_boundaries(const hsrans_ctx *ctx, int states, uint32_t bits, size_t decoded_size, uint64_t *groups_out, size_t capacity)
{
  if ((states != 32 && states != 64) || bits < 10 || bits > 15 || groups_out == nullptr)
    return 0;
  const DeviceGeom dg = ctx ? ctx->geom : default_geom();
  const uint64_t S = (uint64_t)states;
  const uint64_t T = decoded_size + 1 >= S ? (decoded_size - S + 1 + S - 1) / S : 0; // whole groups (rANS32x64_16w.cpp:223)
  const size_t chains = direct_boundaries(dg, (uint32_t)states, bits, T, groups_out, capacity);
  return chains > 1 ? chains - 1 : 0;
}

size_t hsrans_plan_thin(const uint8_t *plan, size_t plan_size, const uint64_t *groups, size_t n_groups, uint8_t *out, size_t out_capacity)
try
{
  return plan_thin(plan, plan_size, groups, n_groups, out, out_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

size_t hsrans_plan_capacity_chains(int container, int states, size_t decoded_size, size_t extra_chains, uint32_t block_size)
{
  if (!valid_codec(container, states, 10))
    return 0;
  return plan_capacity_chains(container, states, decoded_size, extra_chains, block_size);
}

} // extern "C"
