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
      size_t last_coded = groups.size();
      for (size_t k = groups.size(); k-- > 0 && last_coded == groups.size();)
        if (!(groups[k].flags & kGroupFill))
          last_coded = k;
      uint32_t fewest = 0xFFFFFFFFu;
      for (size_t k = 0; k < groups.size() && ok; k++)
      {
        const Group &g = groups[k];
        if (g.flags & kGroupFill)
          continue;
        ok = (g.flags & kGroupMergeable) && g.piece0 == g.begin;
        if (k != last_coded)
          fewest = std::min(fewest, g.count);
      }
      d->spread_min_block = ok ? fewest : 0;
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

// A page-locked, device-mapped host range (hipHostMalloc / hipHostRegister): the address the GPU reaches it at, else null.
static uint8_t *device_view_of_host(const void *ptr, size_t bytes)
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
extern "C++" int dplan_launch(hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, hipStream_t s, uint64_t stream_lo)
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
    // (requesting a round's records and first chunks before its table build: measured, no gain — the other workgroups of the CU
    // fill the gap either way — so off unless asked for)
    // dynamic group order: this launch's own ticket counter (the counter sets of the persistent launches, one head of each used)
    if (d->d_counters != nullptr && getenv("HSRANS_GROUP_STATIC") == nullptr)
      kp.group_tickets = d->d_counters + (size_t)(d->epoch.fetch_add(1, std::memory_order_relaxed) % kCounterSets) * kDynQueues * kDynQueueStride;
  }
  return launch_decode(kp, d->hdr, d->ctx->geom, s, &d->info) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
}

static int decode_device_indexing_impl(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                                       uint32_t index_interval, void *hip_stream, hsrans_dplan **indexed, bool have_lock);

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
  // one-shot entry stages its output and copies it down; HSRANS_HOST_DIRECT=1 switches the direct stores on.)
  uint8_t *out_view = getenv("HSRANS_HOST_DIRECT") != nullptr && ((uintptr_t)out & 3) == 0 ? device_view_of_host(out, (size_t)h.decoded_len) : nullptr;
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
    if (have_index && container == HSRANS_RAW)
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
static int launch_ranges(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out, size_t out_offset,
                         size_t out_length, void *hip_stream)
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
  return dplan_launch(d, (const uint8_t *)d_window - window_offset, (size_t)end, (uint8_t *)d_out - out_offset, (size_t)out_end, (hipStream_t)hip_stream, window_offset);
}

int hsrans_decode_device_window(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out,
                                size_t out_capacity, void *hip_stream)
{
  if (d == nullptr || out_capacity < d->hdr.decoded_len)
    return d == nullptr ? HSRANS_E_ARG : HSRANS_E_FORMAT;
  return launch_ranges(ctx, d, d_window, window_offset, window_length, d_out, 0, out_capacity, hip_stream);
}

int hsrans_decode_device_ranges(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out_window,
                                size_t out_offset, size_t out_length, void *hip_stream)
{
  return launch_ranges(ctx, d, d_window, window_offset, window_length, d_out_window, out_offset, out_length, hip_stream);
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

size_t hsrans_encode_device_raw(hsrans_ctx *ctx, int states, uint32_t bits, const void *d_in, size_t length, void *d_out, size_t out_capacity, const hsrans_hist *hist,
                                uint32_t index_interval, const uint64_t *index_groups, size_t n_index_groups, uint8_t *plan_out, size_t plan_capacity,
                                size_t *plan_size, void *hip_stream, hsrans_dplan **out_dplan)
{
  // SURVEY.md §8(f) row 2, the raw half: rANS32x64_16w.cpp:34-166 carries every coder state from the file's last symbol to its
  // first, so the format has work for exactly ONE wavefront (lane j = state j).  What the GPU adds is that input and stream never
  // leave HBM: a wide kernel counts the bytes, the coding wavefront normalises them exactly as hist.cpp:16-215 does, codes the file
  // back to front through an LDS ring with its table entries fetched two sets ahead, records the checkpoints of the sidecar index
  // on its way, and a wide copy puts the finished image at the front of d_out.  Byte-identical to hsrans_encode_ex (tests).
  if (out_dplan)
    *out_dplan = nullptr;
  if (plan_size)
    *plan_size = 0;
  if (ctx == nullptr || !valid_codec(HSRANS_RAW, states, bits) || d_in == nullptr || d_out == nullptr || length == 0)
    return 0;
  if (length > 0x7FFF0000ull || index_interval % 4 != 0 || ((uintptr_t)d_in & 15) != 0 || ((uintptr_t)d_out & 15) != 0) // (byte offsets inside the slot are 32-bit)
    return 0;
  if (out_capacity < capacity(HSRANS_RAW, states, length))
    return 0;
  const uint32_t S = (uint32_t)states;
  const bool listed = index_groups != nullptr && n_index_groups != 0;
  const bool want_plan = (plan_out != nullptr || out_dplan != nullptr) && (listed || index_interval != 0);
  if (plan_out != nullptr && plan_size == nullptr)
    return 0;
  if (listed)
  {
    if (n_index_groups > 0x7FFFFFFFull)
      return 0;
    for (size_t k = 0; k < n_index_groups; k++) // (the host encoder's rule)
      if (index_groups[k] == 0 || (index_groups[k] % 4) != 0 || (k > 0 && index_groups[k] <= index_groups[k - 1]))
        return 0;
  }
  if (hist != nullptr)
  {
    uint32_t sum = 0;
    for (int k = 0; k < 256; k++)
      sum += hist->symbolCount[k];
    if (sum != (1u << bits))
      return 0;
  }
  const uint64_t whole_groups = length / S;
  // checkpoints the pass will record: interval -> group (k + 1) * interval; list -> the entries below the last whole group
  size_t n_ck = 0;
  std::vector<uint32_t> groups32;
  if (want_plan && listed)
  {
    while (n_ck < n_index_groups && index_groups[n_ck] < whole_groups)
      n_ck++;
    groups32.resize(n_ck);
    for (size_t k = 0; k < n_ck; k++)
      groups32[k] = (uint32_t)index_groups[k];
  }
  else if (want_plan)
    n_ck = whole_groups >= 1 ? (size_t)((whole_groups - 1) / index_interval) : 0;
  EncParams ep{};
  ep.S = S;
  ep.bits = bits;
  ep.n = length;
  ep.block = length;
  ep.n_blocks = 1;
  ep.slot_bytes = encode_slot_bytes(length, S);
  ep.interval = want_plan && !listed ? index_interval : 0;
  ep.max_ck = (uint32_t)n_ck;
  std::lock_guard<std::mutex> guard(ctx->lock);
  if (hipSetDevice(ctx->device) != hipSuccess)
    return 0;
  const size_t meta_bytes = (2 + kEncResultWords + 4) * 8 + 256 * 4 + 256 * 2 + n_ck * 4 + 64;
  const size_t ck_slots = n_ck ? n_ck : 1;
  if (!grow(&ctx->d_enc_scratch, &ctx->d_enc_scratch_cap, ep.slot_bytes) || !grow(&ctx->d_enc_meta, &ctx->d_enc_meta_cap, meta_bytes) ||
      !grow(&ctx->d_enc_ck, &ctx->d_enc_ck_cap, ck_slots * ((size_t)S * 4 + 4)))
    return 0;
  ep.in = (const uint8_t *)d_in;
  ep.out = (uint8_t *)d_out;
  ep.out_cap = out_capacity;
  ep.scratch = ctx->d_enc_scratch;
  ep.image_bytes = (uint64_t *)ctx->d_enc_meta;
  ep.image_off = ep.image_bytes + 1;
  ep.result = ep.image_off + 1;
  ep.stamps = ep.result + kEncResultWords;
  uint32_t *d_counts = (uint32_t *)(ep.stamps + 4);
  uint16_t *d_given = (uint16_t *)(d_counts + 256);
  uint32_t *d_groups = (uint32_t *)(d_given + 256);
  ep.raw_counts = d_counts;
  ep.given_counts = hist ? d_given : nullptr;
  ep.ck_groups = want_plan && listed && n_ck ? d_groups : nullptr;
  ep.n_ck_groups = ep.ck_groups ? (uint32_t)n_ck : 0;
  ep.ck_states = (uint32_t *)ctx->d_enc_ck;
  ep.ck_pos = ep.ck_states + ck_slots * S;
  hipStream_t s = (hipStream_t)hip_stream;
  uint64_t result[kEncResultWords] = {};
  bool ok = true;
  if (hist)
    ok = hipMemcpyAsync(d_given, hist->symbolCount, 512, hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok && ep.ck_groups)
    ok = hipMemcpyAsync(d_groups, groups32.data(), n_ck * 4, hipMemcpyHostToDevice, s) == hipSuccess;
  ok = ok && launch_encode_raw(ep, d_counts, s, &ctx->enc_raw_prepared) == hipSuccess &&
       hipMemcpyAsync(result, ep.result, sizeof(result), hipMemcpyDeviceToHost, s) == hipSuccess;
  if (hipStreamSynchronize(s) != hipSuccess || !ok) // (groups32 / *hist may be read until here)
  {
    (void)hipGetLastError();
    return 0;
  }
  if (getenv("HSRANS_DEBUG_STAMPS"))
  {
    uint64_t st[4] = {};
    if (hipMemcpy(st, ep.stamps, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[hsrans raw encode stamps] us: counts+normalise+table %.1f  rANS pass %.1f\n", (double)(st[2] - st[0]) / 100.0, (double)(st[3] - st[2]) / 100.0);
  }
  if (result[1] != 1 || result[2] != 0)
    return 0;
  const size_t total = (size_t)result[0];
  if (!want_plan)
    return total;

  // ---- the sidecar plan: checkpoints and the stream's header come down (2.1 MB for the one-chain-per-wavefront index), the host
  // assembles exactly what hsrans_encode_ex emits (raw_plan_from_checkpoints is that code) ----
  const size_t header_bytes = 16 + 512 + 4 * (size_t)S;
  std::vector<uint8_t> header(header_bytes);
  std::vector<uint32_t> ck_states(n_ck * S), ck_pos(n_ck);
  std::vector<uint64_t> ck_group(n_ck), ck_wfe(n_ck);
  if (hipMemcpy(header.data(), d_out, header_bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      (n_ck && (hipMemcpy(ck_states.data(), ep.ck_states, n_ck * S * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(ck_pos.data(), ep.ck_pos, n_ck * 4, hipMemcpyDeviceToHost) != hipSuccess)))
    return 0;
  for (size_t k = 0; k < n_ck; k++)
  {
    ck_group[k] = listed ? index_groups[k] : (uint64_t)(k + 1) * index_interval;
    ck_wfe[k] = ck_pos[k];
  }
  const size_t pcap = plan_capacity_chains(HSRANS_RAW, states, length, n_ck, 0);
  std::vector<uint8_t> own;
  uint8_t *blob = plan_out;
  size_t cap = plan_capacity;
  if (blob == nullptr)
  {
    own.resize(pcap);
    blob = own.data();
    cap = own.size();
  }
  const size_t psize = raw_plan_from_checkpoints(states, bits, length, total, (const uint16_t *)(header.data() + 16), (const uint32_t *)(header.data() + 16 + 512), n_ck,
                                                 ck_group.data(), ck_wfe.data(), ck_states.data(), listed ? 0 : index_interval, blob, cap);
  if (psize == 0)
    return 0;
  if (plan_size)
    *plan_size = psize;
  if (out_dplan != nullptr && hsrans_dplan_create(ctx, blob, psize, out_dplan) != HSRANS_OK)
    return 0;
  return total;
}

size_t hsrans_encode_device(hsrans_ctx *ctx, int container, int states, uint32_t bits, const void *d_in, size_t length, void *d_out, size_t out_capacity,
                            uint32_t block_size, uint32_t index_interval, void *hip_stream, hsrans_dplan **out_dplan)
{
  if (container == HSRANS_RAW) // one wavefront (the format's one dependent chain); block_size has no meaning
    return hsrans_encode_device_raw(ctx, states, bits, d_in, length, d_out, out_capacity, nullptr, index_interval, nullptr, 0, nullptr, 0, nullptr, hip_stream, out_dplan);
  if (out_dplan)
    *out_dplan = nullptr;
  if (ctx == nullptr || container != HSRANS_MT || !valid_codec(container, states, bits) || d_in == nullptr || d_out == nullptr || length == 0)
    return 0;
  if (block_size == 0 || block_size % 64 != 0 || block_size > (1u << 30) || index_interval % 4 != 0 || ((uintptr_t)d_in & 15) != 0 || ((uintptr_t)d_out & 15) != 0)
    return 0;
  if (out_capacity < capacity(container, states, length)) // same contract as the host encoders
    return 0;
  EncParams ep{};
  ep.S = (uint32_t)states;
  ep.bits = bits;
  ep.n = length;
  ep.block = block_size;
  ep.n_blocks = encode_block_count(length, block_size, ep.S);
  if (ep.n_blocks == 0)
    return 0;
  ep.slot_bytes = encode_slot_bytes(block_size, ep.S);
  ep.interval = out_dplan ? index_interval : 0; // checkpoints only serve the plan
  ep.max_ck = ep.interval ? (block_size / ep.S - 1) / ep.interval : 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  if (hipSetDevice(ctx->device) != hipSuccess)
    return 0;
  const bool stamps = getenv("HSRANS_DEBUG_STAMPS") != nullptr;
  const size_t nb = ep.n_blocks;
  const bool wide_hist = getenv("HSRANS_ENC_WAVE_HISTOGRAM") == nullptr; // (=1: the coding wavefront counts its own block, as in rounds 1-3)
  const size_t meta_bytes = (nb * 2 + kEncResultWords) * 8 + nb * 2 * 4 + (stamps ? nb * 4 * 8 : 0) + 64 + (wide_hist ? nb * 1024 + 16 : 0);
  const size_t ck_slots = nb * (ep.max_ck ? ep.max_ck : 1);
  if (!grow(&ctx->d_enc_scratch, &ctx->d_enc_scratch_cap, nb * ep.slot_bytes) || !grow(&ctx->d_enc_meta, &ctx->d_enc_meta_cap, meta_bytes) ||
      !grow(&ctx->d_enc_ck, &ctx->d_enc_ck_cap, ck_slots * ((size_t)ep.S * 4 + 4)))
    return 0;
  ep.in = (const uint8_t *)d_in;
  ep.out = (uint8_t *)d_out;
  ep.out_cap = out_capacity;
  ep.scratch = ctx->d_enc_scratch;
  ep.image_bytes = (uint64_t *)ctx->d_enc_meta;
  ep.image_off = ep.image_bytes + nb;
  ep.result = ep.image_off + nb;
  uint64_t *after = ep.result + kEncResultWords;
  ep.stamps = stamps ? after : nullptr;
  after += stamps ? nb * 4 : 0;
  ep.chain_count = (uint32_t *)after;
  ep.chain_off = ep.chain_count + nb;
  if (wide_hist)
    ep.raw_counts = (const uint32_t *)(((uintptr_t)(ep.chain_off + nb) + 15) & ~(uintptr_t)15);
  ep.ck_states = (uint32_t *)ctx->d_enc_ck;
  ep.ck_pos = ep.ck_states + ck_slots * ep.S;
  hipStream_t s = (hipStream_t)hip_stream;
  uint64_t result[kEncResultWords] = {};
  if (launch_encode(ep, s, &ctx->enc_prepared) != hipSuccess)
    return 0;
  if (hipMemcpyAsync(result, ep.result, sizeof(result), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return 0;
  if (stamps) // printed, not returned: a tuning aid only
  {
    std::vector<uint64_t> st(nb * 4);
    if (hipMemcpy(st.data(), ep.stamps, st.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
    {
      double ph[3] = {0, 0, 0};
      uint64_t lo = ~0ull, hi = 0;
      for (uint32_t b = 0; b < ep.n_blocks; b++)
      {
        for (int k = 0; k < 3; k++)
          ph[k] += (double)(st[b * 4 + k + 1] - st[b * 4 + k]);
        lo = st[b * 4] < lo ? st[b * 4] : lo;
        hi = st[b * 4 + 3] > hi ? st[b * 4 + 3] : hi;
      }
      fprintf(stderr, "[hsrans encode stamps] blocks %u  mean us: histogram %.1f  normalise+table %.1f  rANS pass %.1f   first start -> last end %.1f us\n", ep.n_blocks,
              ph[0] / ep.n_blocks / 100.0, ph[1] / ep.n_blocks / 100.0, ph[2] / ep.n_blocks / 100.0, (double)(hi - lo) / 100.0);
    }
  }
  if (result[1] != 1)
    return 0;
  const size_t total = (size_t)result[0];
  if (out_dplan == nullptr)
    return total;

  // ---- the stream's plan, written on the device (K_plan), wrapped into a device plan ready for hsrans_decode_device ----
  if (result[2] == 0 || result[2] > 0xFFFFFFFFull)
    return 0;
  hsrans_dplan *d = new (std::nothrow) hsrans_dplan;
  if (d == nullptr)
    return 0;
  d->ctx = ctx;
  PlanHeader h{};
  memcpy(h.magic, "HSRPLAN1", 8);
  h.container = HSRANS_MT;
  h.states = ep.S;
  h.bits = bits;
  h.decoded_len = length;
  h.stream_len = total;
  h.n_chains = h.n_pieces = (uint32_t)result[2];
  h.shared_hist = result[3] == 1 ? 1 : 0; // exactly one block with a histogram (hsrans_host.cpp PlanBuilder::serialize)
  h.aux_off = h.shared_hist ? result[4] : 0;
  h.interval = ep.interval;
  const size_t bytes = (size_t)plan_size(h.n_chains, h.n_pieces, h.states, 0);
  const bool grouped = ep.interval != 0 && ep.n_blocks < h.n_chains;
  // few large blocks: cut every block's chains into parts so that there are about two workgroup tasks per resident workgroup
  // (parts of >= 128 chains, only while there are fewer blocks than resident workgroups: see hsrans_dplan_create)
  const size_t want = (size_t)kGroupPartsPerCU * ctx->geom.num_cus;
  ep.group_split = 1;
  if (grouped && nb < want)
    ep.group_split = (uint32_t)std::max<size_t>(1, std::min<size_t>({(want + nb - 1) / nb, (size_t)(ep.max_ck + 1) / kGroupPartChains, (size_t)64}));
  bool ok = hipMalloc((void **)&d->d_plan, bytes) == hipSuccess && hipMalloc((void **)&d->d_status, 64) == hipSuccess &&
            (!grouped || grow(&d->d_groups, &d->d_groups_cap, nb * ep.group_split * sizeof(Group))) && hipMemsetAsync(d->d_plan, 0, bytes, s) == hipSuccess &&
            hipMemsetAsync(d->d_status, 0, 64, s) == hipSuccess && hipMemcpyAsync(d->d_plan, &h, sizeof(h), hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok)
  {
    ep.plan = d->d_plan;
    ep.groups = grouped ? d->d_groups : nullptr;
    ep.n_chains = h.n_chains;
    ok = launch_encode_plan(ep, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (!ok)
  {
    hsrans_dplan_destroy(d);
    return 0;
  }
  d->hdr = h;
  d->plan_bytes = bytes;
  d->out_hi = h.decoded_len;
  d->n_groups = grouped ? ep.n_blocks * ep.group_split : 0;
  d->groups_lean = grouped && h.states == 64; // k_plan_blocks writes mergeable runs and fill groups only
  d->spread_min_block = d->groups_lean ? ep.max_ck + 1 : 0; // (every coded block but the last has max_ck + 1 chains)
  if (grouped)
  {
    // ticket counters of the dynamic group order (as dplan_fill); without them the launch falls back to the static order
    const size_t cbytes = (size_t)kCounterSets * kDynQueues * kDynQueueStride * 8;
    if (hipMalloc((void **)&d->d_counters, cbytes) == hipSuccess && (hipMemsetAsync(d->d_counters, 0, cbytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
    {
      (void)hipFree(d->d_counters);
      d->d_counters = nullptr;
    }
  }
  *out_dplan = d;
  return total;
}

// The chains of a plan with a checkpoint every `index_interval` groups (absolute group numbers: slot = group / interval), given
// the base plan (one single-piece chain per block) and what a recording decode pass left at the checkpoints
static void add_interval_chains(PlanBuilder &pb, const PlanHeader &h, const uint32_t *cf0, const Piece *pc0, const uint32_t *st0, uint32_t index_interval,
                                const uint32_t *ck_states, const uint64_t *ck_words)
{
  const uint32_t S = h.states;
  for (uint32_t ch = 0; ch < h.n_chains; ch++)
  {
    const Piece &bp = pc0[cf0[ch]];
    if (bp.flags & kPieceFill)
    {
      pb.add_chain(bp, nullptr);
      continue;
    }
    const uint64_t T = bp.steps, g_abs0 = bp.out_off / S;
    for (uint64_t g = 0; g < T || g == 0; g += index_interval)
    {
      Piece p{};
      p.hist_off = bp.hist_off;
      p.out_off = bp.out_off + g * S;
      const uint64_t slot = (g_abs0 + g) / index_interval;
      p.words_off = g == 0 ? bp.words_off : ck_words[slot];
      const uint64_t steps = T - g < index_interval ? T - g : index_interval;
      p.steps = (uint32_t)steps;
      p.tail = (uint16_t)(g + steps == T ? bp.tail : 0);
      pb.add_chain(p, g == 0 ? st0 + (size_t)bp.state_idx * S : &ck_states[slot * S]);
    }
  }
}

static size_t index_build_impl(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint32_t index_interval,
                               const uint64_t *groups, size_t n_groups, uint8_t *plan_out, size_t plan_capacity)
{
  // One pass over an existing stream that records {states, read cursor} every `index_interval` groups inside every rANS
  // piece of the stream's own plan (raw: one sequential wavefront; mt_: one wavefront per block, in parallel); the
  // checkpoints then become additional chains.  A block_ stream is one chain with inline headers (the position of a block's
  // header is only known once the block before it is decoded): the single wavefront that walks it also reports every block
  // header it meets and the states it enters the block with, and the plan gets one chain per block plus the checkpoints.
  if (ctx == nullptr || in == nullptr || plan_out == nullptr || !valid_codec(container, states, bits))
    return 0;
  // checkpoints every index_interval groups, or (groups != nullptr) at explicit ascending group indices
  if (groups == nullptr && (index_interval == 0 || index_interval % 4 != 0))
    return 0;
  if (in_length < 16)
    return 0;
  if (groups != nullptr)
  {
    index_interval = 0;
    if (n_groups == 0 || n_groups > 0xFFFFFFFFull || container == HSRANS_BLOCK)
      return 0;
    for (size_t k = 0; k < n_groups; k++)
      if (groups[k] == 0 || (groups[k] % 4) != 0 || (k > 0 && groups[k] <= groups[k - 1]))
        return 0;
  }
  uint64_t out_len;
  memcpy(&out_len, in, 8);
  // (the header's decoded length is untrusted: the base plan is sized by the chains the stream really holds, at most ~40x the stream)
  std::vector<uint8_t> base;
  if (!plan_build_vec(container, states, bits, in, in_length, (size_t)out_len, &base))
    return 0;
  const size_t base_size = base.size();
  PlanHeader h;
  memcpy(&h, base.data(), sizeof(h));
  const uint32_t *cf0 = (const uint32_t *)(base.data() + plan_chain_first_off());
  const Piece *pc0 = (const Piece *)(base.data() + plan_pieces_off(h.n_chains));
  const uint32_t *st0 = (const uint32_t *)(base.data() + plan_states_off(h.n_chains, h.n_pieces));
  const uint32_t S = (uint32_t)states;
  const bool walk = (h.flags & kPlanWalk) != 0;
  if (!walk && h.n_pieces != h.n_chains) // the planner only produces single-piece chains for raw and mt_
    return 0;
  // A raw stream is one dependent chain: one wavefront records its checkpoints at ~0.65 GB/s, one host core with this
  // library's SIMD decoder at 2-3 GB/s and without the upload — so raw streams are indexed on the host (same plan, byte for
  // byte; HSRANS_INDEX_ON_GPU=1 keeps the wavefront pass).  mt_ blocks (one wavefront each, in parallel) and block_ streams
  // (the walk that also reports the inline headers) stay on the GPU.
  if (container == HSRANS_RAW && getenv("HSRANS_INDEX_ON_GPU") == nullptr)
  {
    std::vector<uint64_t> own;
    if (groups == nullptr)
    {
      const uint64_t T = h.n_pieces == 1 ? pc0[0].steps : 0;
      for (uint64_t g = index_interval; g < T; g += index_interval)
        own.push_back(g);
      if (own.empty())
        return plan_capacity >= base_size ? (memcpy(plan_out, base.data(), base_size), base_size) : 0;
    }
    return cpu::index_build(cpu::best_level(), 1, container, states, bits, in, in_length, groups ? groups : own.data(), groups ? n_groups : own.size(), plan_out,
                            plan_capacity, groups ? 0 : index_interval);
  }
  const uint64_t n_ck = groups ? n_groups : out_len / S / index_interval + 2;
  // block_: room for blocks of >= 4 KiB on average (the reference's smallest block is 32 KiB, block_rANS32x64_16w_encode.cpp:21-39)
  const uint64_t max_blocks = walk ? out_len / 4096 + 16 : 0;

  std::lock_guard<std::mutex> guard(ctx->lock);
  if (hipSetDevice(ctx->device) != hipSuccess)
    return 0;
  const size_t in_pad = (in_length + 15) / 16 * 16;
  if (!grow(&ctx->d_in, &ctx->d_in_cap, in_pad) || !grow(&ctx->d_out, &ctx->d_out_cap, (size_t)out_len + 16) || !grow(&ctx->d_plan, &ctx->d_plan_cap, base_size))
    return 0;
  uint32_t *d_ck_states = nullptr;
  uint64_t *d_ck_words = nullptr, *d_groups = nullptr;
  uint64_t *d_walk_blocks = nullptr;
  uint32_t *d_walk_states = nullptr, *d_walk_count = nullptr;
  size_t result = 0;
  hipStream_t s = ctx->stream;
  std::vector<uint32_t> ck_states(n_ck * S);
  std::vector<uint64_t> ck_words(n_ck);
  uint32_t status = 0xFFFFFFFF;
  do
  {
    if (hipMalloc((void **)&d_ck_states, n_ck * S * 4) != hipSuccess || hipMalloc((void **)&d_ck_words, n_ck * 8) != hipSuccess)
      break;
    if (groups && (hipMalloc((void **)&d_groups, n_groups * 8) != hipSuccess || hipMemcpyAsync(d_groups, groups, n_groups * 8, hipMemcpyHostToDevice, s) != hipSuccess))
      break;
    if (walk && (hipMalloc((void **)&d_walk_blocks, max_blocks * 24) != hipSuccess || hipMalloc((void **)&d_walk_states, max_blocks * S * 4) != hipSuccess ||
                 hipMalloc((void **)&d_walk_count, 4) != hipSuccess || hipMemsetAsync(d_walk_count, 0, 4, s) != hipSuccess))
      break;
    if (hipMemcpyAsync(ctx->d_in, in, in_length, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(ctx->d_plan, base.data(), base_size, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(ctx->d_status, 0, 4, s) != hipSuccess)
      break;
    KParams kp{};
    kp.stream = ctx->d_in;
    kp.stream_len = in_length;
    kp.out = ctx->d_out;
    kp.out_cap = out_len;
    kp.plan = ctx->d_plan;
    kp.status = ctx->d_status;
    kp.ckpt_states = d_ck_states;
    kp.ckpt_words = d_ck_words;
    kp.ckpt_interval = index_interval;
    kp.ckpt_groups = d_groups;
    kp.n_ckpt_groups = (uint32_t)(groups ? n_groups : 0);
    kp.walk_blocks = d_walk_blocks;
    kp.walk_states = d_walk_states;
    kp.walk_count = d_walk_count;
    kp.walk_max_blocks = (uint32_t)(max_blocks > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_blocks);
    PlanHeader hl = h;
    hl.shared_hist = 0; // private tables: every chain of the pass builds its own (raw has one chain, mt_ one per block)
    if (launch_decode(kp, hl, ctx->geom, s, nullptr) != hipSuccess)
      break;
    if (hipMemcpyAsync(ck_states.data(), d_ck_states, n_ck * S * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(ck_words.data(), d_ck_words, n_ck * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&status, ctx->d_status, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      break;
    if (status != 0)
      break;
    PlanBuilder pb;
    pb.begin(container, states, bits, out_len, in_length);
    pb.hdr.interval = index_interval;
    if (container == HSRANS_RAW)
    {
      uint16_t counts[256];
      memcpy(counts, in + pc0[0].hist_off, 512);
      pb.set_hist(counts);
    }
    if (walk)
    {
      uint32_t n_blocks = 0;
      if (hipMemcpy(&n_blocks, d_walk_count, 4, hipMemcpyDeviceToHost) != hipSuccess || n_blocks == 0 || n_blocks > max_blocks)
        break;
      std::vector<uint64_t> blocks((size_t)n_blocks * 3);
      std::vector<uint32_t> bstates((size_t)n_blocks * S);
      if (hipMemcpy(blocks.data(), d_walk_blocks, blocks.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(bstates.data(), d_walk_states, bstates.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        break;
      const uint64_t whole_file = out_len / S; // whole groups of the file (block_rANS32x64_16w_decode.cpp:82-88)
      const uint64_t tail = out_len - whole_file * S;
      bool ok = true;
      for (uint32_t b = 0; b < n_blocks && ok; b++)
      {
        const uint64_t pos = blocks[3 * (size_t)b], at = blocks[3 * (size_t)b + 1], hdr = blocks[3 * (size_t)b + 2];
        const bool last = b + 1 == n_blocks;
        if (hdr >> 63)
        {
          Piece p{};
          p.out_off = at;
          p.hist_off = (hdr >> 54) & 0xFF;
          p.fill_len = hdr & (((uint64_t)1 << 54) - 1);
          p.flags = kPieceChainStart | kPieceFill;
          pb.add_chain(p, nullptr);
          ok = !(last && at + p.fill_len < out_len); // a tail behind a single-symbol block has no histogram
          continue;
        }
        const uint64_t g0 = at / S;
        const uint64_t g1 = std::min<uint64_t>((at + hdr + S - 1) / S, whole_file); // the decoder stops at the last whole group
        const uint64_t T = g1 > g0 ? g1 - g0 : 0;
        for (uint64_t g = 0; g < T || g == 0; g += index_interval)
        {
          Piece p{};
          p.hist_off = pos + 8;
          p.out_off = at + g * S;
          const uint64_t slot = (g0 + g) / index_interval;
          p.words_off = g == 0 ? pos + 8 + 512 : ck_words[slot];
          const uint64_t steps = T - g < index_interval ? T - g : index_interval;
          p.steps = (uint32_t)steps;
          p.tail = (uint16_t)(last && g + steps >= T ? tail : 0);
          pb.add_chain(p, g == 0 ? &bstates[(size_t)b * S] : &ck_states[slot * S]);
        }
      }
      if (!ok)
        break;
    }
    else if (groups != nullptr)
    {
      size_t k = 0; // next boundary
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
        uint64_t g = 0; // groups of this piece already assigned to chains
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
    }
    else
      add_interval_chains(pb, h, cf0, pc0, st0, index_interval, ck_states.data(), ck_words.data());
    result = pb.serialize(plan_out, plan_capacity);
  } while (false);
  if (d_ck_states)
    (void)hipFree(d_ck_states);
  if (d_ck_words)
    (void)hipFree(d_ck_words);
  if (d_groups)
    (void)hipFree(d_groups);
  if (d_walk_blocks)
    (void)hipFree(d_walk_blocks);
  if (d_walk_states)
    (void)hipFree(d_walk_states);
  if (d_walk_count)
    (void)hipFree(d_walk_count);
  return result;
}

size_t hsrans_index_build(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint32_t index_interval,
                          uint8_t *plan_out, size_t plan_capacity)
try
{
  return index_build_impl(ctx, container, states, bits, in, in_length, index_interval, nullptr, 0, plan_out, plan_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

size_t hsrans_index_build_at(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, const uint64_t *groups,
                             size_t n_groups, uint8_t *plan_out, size_t plan_capacity)
try
{
  if (groups == nullptr)
    return 0;
  return index_build_impl(ctx, container, states, bits, in, in_length, 0, groups, n_groups, plan_out, plan_capacity);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return 0;
}

// The first decode of a stream that came without an index (a reference-emitted mt_ stream planned by hsrans_plan_build or on the
// device by hsrans_dplan_create_from_device_stream: one chain per block, most wave slots empty) also RECORDS the coder states and
// the read cursor every `index_interval` groups — two stores per checkpoint on a pass that is latency-bound anyway — and returns
// the plan with those checkpoints for every later decode of the same stream.  The stream never leaves device memory; the plan
// blob (chain table, a few MB) is assembled on the host as in hsrans_index_build, whose result it equals byte for byte.
int hsrans_decode_device_indexing(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                                  uint32_t index_interval, void *hip_stream, hsrans_dplan **indexed)
{
  return decode_device_indexing_impl(ctx, d, d_stream, stream_length, d_out, out_capacity, index_interval, hip_stream, indexed, false);
}

// have_lock: the caller (hsrans_decode_host) already holds ctx->lock
static int decode_device_indexing_impl(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                                       uint32_t index_interval, void *hip_stream, hsrans_dplan **indexed, bool have_lock)
try
{
  if (ctx == nullptr || d == nullptr || d_stream == nullptr || d_out == nullptr || indexed == nullptr || d->ctx != ctx)
    return HSRANS_E_ARG;
  *indexed = nullptr;
  if (((uintptr_t)d_stream & 15) != 0 || ((uintptr_t)d_out & 3) != 0 || index_interval == 0 || (index_interval % 4) != 0)
    return HSRANS_E_ARG;
  const PlanHeader &h = d->hdr;
  // base plans only: one single-piece chain per block (raw: one chain), no inline-header walk (block_ streams: hsrans_index_build)
  if ((h.flags & kPlanWalk) || h.n_pieces != h.n_chains || h.interval != 0 || d->d_plan == nullptr || d->plan_bytes == 0)
    return HSRANS_E_ARG;
  if (stream_length < h.stream_len || out_capacity < h.decoded_len)
    return HSRANS_E_FORMAT;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hipStream_t s = (hipStream_t)hip_stream;
  const bool trace = getenv("HSRANS_INDEXING_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  const uint32_t S = h.states;
  const uint64_t n_ck = h.decoded_len / S / index_interval + 2;
  // mt_ streams (one single-piece chain per block, histograms in the stream): the indexed plan is assembled ON THE DEVICE behind the
  // recording pass — one allocation, three launches, one synchronisation; nothing but two words comes back to the host
  // (HSRANS_INDEX_ASSEMBLE_ON_HOST=1: round 3's path — checkpoints down, blob built by one core, blob up — still what raw plans take)
  if (h.container == HSRANS_MT && (h.flags & (kPlanWalk | kPlanHasHist | kPlanMergeable)) == 0 && getenv("HSRANS_INDEX_ASSEMBLE_ON_HOST") == nullptr)
  {
    std::unique_lock<std::mutex> guard(ctx->lock, std::defer_lock); // (the checkpoint buffer belongs to the context)
    if (!have_lock)
      guard.lock();
    const uint64_t max_chains64 = std::min<uint64_t>((uint64_t)h.n_chains + n_ck, 0xFFFFFFF0u);
    const uint32_t max_chains = (uint32_t)max_chains64;
    const size_t st_bytes = (size_t)n_ck * S * 4, wd_bytes = (size_t)n_ck * 8;
    if (!grow(&ctx->d_enc_ck, &ctx->d_enc_ck_cap, st_bytes + wd_bytes))
      return HSRANS_E_HIP;
    const uint32_t nb = h.n_chains;
    // few large blocks: every block's chains in parts, so that there are about two workgroup tasks per resident workgroup (as dplan_fill)
    const size_t want = (size_t)kGroupPartsPerCU * ctx->geom.num_cus;
    uint32_t group_split = 1;
    if (nb < want)
      group_split = (uint32_t)std::max<size_t>(1, std::min<size_t>({(want + nb - 1) / nb, (size_t)(n_ck / nb + 1) / kGroupPartChains, (size_t)64}));
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t plan_max = (size_t)plan_size(max_chains, max_chains, S, 0);
    const size_t counter_bytes = (size_t)kCounterSets * kDynQueues * kDynQueueStride * 8;
    const size_t group_bytes = (size_t)nb * group_split * sizeof(Group);
    hsrans_dplan *nd = new (std::nothrow) hsrans_dplan;
    if (nd == nullptr)
      return HSRANS_E_HIP;
    nd->ctx = ctx;
    const size_t arena = 256 + up256(counter_bytes) + up256(plan_max) + up256(group_bytes) + up256((size_t)nb * 4) + 256;
    if (!grow(&nd->d_arena, &nd->d_arena_cap, arena))
    {
      hsrans_dplan_destroy(nd);
      return HSRANS_E_HIP;
    }
    uint8_t *at = nd->d_arena;
    auto carve = [&](size_t bytes) { uint8_t *ptr = at; at += up256(bytes); return ptr; };
    nd->d_status = (uint32_t *)carve(64);
    nd->d_counters = (unsigned long long *)carve(counter_bytes);
    nd->d_plan = carve(plan_max);
    nd->d_plan_cap = plan_max;
    nd->d_groups = carve(group_bytes);
    nd->d_groups_cap = group_bytes;
    uint32_t *d_chain_off = (uint32_t *)carve((size_t)nb * 4);
    uint64_t *d_result = (uint64_t *)carve(64);
    nd->arena_used = (size_t)(at - nd->d_arena);
    KParams kp{};
    kp.stream = (const uint8_t *)d_stream;
    kp.stream_len = stream_length;
    kp.out = (uint8_t *)d_out;
    kp.out_cap = out_capacity;
    kp.plan = d->d_plan;
    kp.status = d->d_status;
    kp.ckpt_states = (uint32_t *)ctx->d_enc_ck;
    kp.ckpt_words = (uint64_t *)(ctx->d_enc_ck + st_bytes);
    kp.ckpt_interval = index_interval;
    PlanHeader hl = h;
    hl.shared_hist = 0; // private tables, as in hsrans_index_build's pass
    IndexArgs ia{};
    ia.base = d->d_plan;
    ia.n_base = nb;
    ia.S = S;
    ia.interval = index_interval;
    ia.ck_states = kp.ckpt_states;
    ia.ck_words = kp.ckpt_words;
    ia.chain_off = d_chain_off;
    ia.result = d_result;
    ia.plan = nd->d_plan;
    ia.max_chains = max_chains;
    ia.groups = (Group *)nd->d_groups;
    ia.group_split = group_split;
    ia.stream_len = h.stream_len;
    uint32_t status = 0xFFFFFFFF;
    uint64_t counted[4] = {}; // chains in all, blocks with a histogram, the (one) histogram's offset, fewest chains of a coded block but the last
    const uint64_t &total = counted[0];
    const bool ok = hipMemsetAsync(nd->d_arena, 0, nd->arena_used, s) == hipSuccess && launch_decode(kp, hl, ctx->geom, s, nullptr) == hipSuccess &&
                    launch_index_assemble(ia, s) == hipSuccess && hipMemcpyAsync(counted, d_result, sizeof(counted), hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipMemcpyAsync(&status, d->d_status, 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    const bool synced = hipStreamSynchronize(s) == hipSuccess; // (nothing queued above may still be running when this returns, whatever failed)
    int rc = ok && synced ? HSRANS_OK : HSRANS_E_HIP;
    if (rc == HSRANS_OK && status != 0) // the pass found a bad histogram / header: reported and cleared like hsrans_dplan_status does
      rc = hipMemsetAsync(d->d_status, 0, 4, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess ? HSRANS_E_DEVICE : HSRANS_E_HIP;
    if (rc == HSRANS_OK && (total < nb || total > max_chains))
      rc = HSRANS_E_FORMAT;
    if (rc != HSRANS_OK)
    {
      (void)hipGetLastError();
      hsrans_dplan_destroy(nd);
      return rc;
    }
    nd->hdr = h;
    nd->hdr.n_chains = nd->hdr.n_pieces = (uint32_t)total;
    nd->hdr.interval = index_interval;
    nd->hdr.shared_hist = counted[1] == 1 ? 1 : 0;
    nd->hdr.aux_off = nd->hdr.shared_hist ? counted[2] : 0;
    nd->plan_bytes = (size_t)plan_size((uint32_t)total, (uint32_t)total, S, 0);
    nd->out_hi = h.decoded_len;
    const bool grouped = total > nb; // (no checkpoint fell inside any block: one chain per block, the ungrouped launch)
    nd->n_groups = grouped ? nb * group_split : 0;
    nd->groups_lean = grouped && S == 64;
    const uint64_t fewest = counted[3] == 0 ? ~0ull : ~counted[3]; // ([3]: ~(the fewest chains of a coded block that is not the last); 0 = there is none)
    nd->spread_min_block = nd->groups_lean ? (uint32_t)std::min<uint64_t>(fewest, 0xFFFFFFFFu) : 0;
    if (!grouped)
      nd->d_groups = nullptr, nd->d_counters = nullptr;
    if (getenv("HSRANS_DEBUG_STAMPS") && hipMalloc((void **)&nd->d_stamps, kStampWaves * 8 * 8) == hipSuccess)
      (void)hipMemset(nd->d_stamps, 0, kStampWaves * 8 * 8);
    if (trace)
      fprintf(stderr, "hsrans_decode_device_indexing: on the device: %.3f ms in all (%llu chains, %zu plan bytes)\n", ms(t0, now()), (unsigned long long)total, nd->plan_bytes);
    *indexed = nd;
    return HSRANS_OK;
  }
  // page-locked staging (kept by the context): [checkpoint states | cursors | base plan] down, then the new plan blob up —
  // from pageable memory these copies (12.5 MB of states each way for 100 MB at 32 groups) took 15 ms, the decode 0.25
  std::unique_lock<std::mutex> guard(ctx->lock, std::defer_lock);
  if (!have_lock)
    guard.lock();
  const size_t st_bytes = (size_t)n_ck * S * 4, wd_bytes = (size_t)n_ck * 8, base_bytes = (d->plan_bytes + 15) / 16 * 16;
  const size_t new_cap = (size_t)plan_size((uint32_t)std::min<uint64_t>(h.n_chains + n_ck, 0xFFFFFFF0u), (uint32_t)std::min<uint64_t>(h.n_chains + n_ck, 0xFFFFFFF0u), S, kPlanHasHist);
  if (!grow_pinned(&ctx->h_pin, &ctx->h_pin_cap, st_bytes + wd_bytes + base_bytes + new_cap))
    return HSRANS_E_HIP;
  uint32_t *ck_states = (uint32_t *)ctx->h_pin;
  uint64_t *ck_words = (uint64_t *)(ctx->h_pin + st_bytes);
  uint8_t *base = ctx->h_pin + st_bytes + wd_bytes;
  uint8_t *plan = base + base_bytes;
  size_t plan_bytes = 0;
  // (the checkpoints land in the context's checkpoint buffer — the GPU encoder's, kept and grown — not in fresh allocations)
  if (!grow(&ctx->d_enc_ck, &ctx->d_enc_ck_cap, st_bytes + wd_bytes))
    return HSRANS_E_HIP;
  uint32_t *d_ck_states = (uint32_t *)ctx->d_enc_ck;
  uint64_t *d_ck_words = (uint64_t *)(ctx->d_enc_ck + st_bytes);
  int rc = HSRANS_E_HIP;
  do
  {
    KParams kp{};
    kp.stream = (const uint8_t *)d_stream;
    kp.stream_len = stream_length;
    kp.out = (uint8_t *)d_out;
    kp.out_cap = out_capacity;
    kp.plan = d->d_plan;
    kp.status = d->d_status;
    kp.ckpt_states = d_ck_states;
    kp.ckpt_words = d_ck_words;
    kp.ckpt_interval = index_interval;
    PlanHeader hl = h;
    hl.shared_hist = 0; // private tables, as in hsrans_index_build's pass
    uint32_t status = 0xFFFFFFFF;
    if (launch_decode(kp, hl, ctx->geom, s, nullptr) != hipSuccess ||
        hipMemcpyAsync(base, d->d_plan, d->plan_bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(ck_states, d_ck_states, st_bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(ck_words, d_ck_words, wd_bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&status, d->d_status, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      break;
    const auto t1 = now();
    if (status != 0) // the pass found a bad histogram / header: reported and cleared like hsrans_dplan_status does
    {
      rc = hipMemsetAsync(d->d_status, 0, 4, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess ? HSRANS_E_DEVICE : HSRANS_E_HIP;
      break;
    }
    PlanHeader hb;
    if (!read_header(base, d->plan_bytes, &hb) || hb.n_chains != h.n_chains || hb.n_pieces != h.n_pieces || hb.states != h.states ||
        !plan_validate(base, d->plan_bytes, h.stream_len, h.decoded_len))
    {
      rc = HSRANS_E_FORMAT;
      break;
    }
    PlanBuilder pb;
    pb.begin((int)h.container, (int)S, h.bits, h.decoded_len, h.stream_len);
    pb.reserve((size_t)h.n_chains + n_ck);
    pb.hdr.interval = index_interval;
    if (hb.flags & kPlanHasHist)
    {
      uint16_t counts[256];
      memcpy(counts, base + plan_hist_off(hb.n_chains, hb.n_pieces, hb.states), 512);
      pb.set_hist(counts);
    }
    add_interval_chains(pb, hb, (const uint32_t *)(base + plan_chain_first_off()), (const Piece *)(base + plan_pieces_off(hb.n_chains)),
                        (const uint32_t *)(base + plan_states_off(hb.n_chains, hb.n_pieces)), index_interval, ck_states, ck_words);
    const auto t2 = now();
    plan_bytes = pb.serialize(plan, new_cap);
    rc = plan_bytes == 0 ? HSRANS_E_FORMAT : HSRANS_OK;
    if (trace)
      fprintf(stderr, "hsrans_decode_device_indexing: pass + copies %.3f ms, validate + chains %.3f ms, serialize %.3f ms (%zu bytes)\n", ms(t0, t1), ms(t1, t2), ms(t2, now()), plan_bytes);
  } while (false);
  if (rc != HSRANS_OK)
  {
    (void)hipStreamSynchronize(s); // nothing queued above may still be writing the staging buffers (or the caller's d_out) after the return
    (void)hipGetLastError();
    return rc;
  }
  const auto t3 = now();
  const int rc2 = hsrans_dplan_create(ctx, plan, plan_bytes, indexed);
  if (trace)
    fprintf(stderr, "hsrans_decode_device_indexing: hsrans_dplan_create %.3f ms\n", ms(t3, now()));
  return rc2;
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

// ---- host buffers, PCIe legs overlapped ---------------------------------------------------------------------------
struct hsrans_hpipe
{
  hsrans_ctx *ctx = nullptr;
  PlanHeader hdr{};
  struct Slice
  {
    hsrans_dplan *dplan = nullptr;
    uint64_t in_ranges[4] = {}; // {head_begin, head_end, body_begin, body_end} of the stream (hsrans_plan_stream_ranges)
    uint64_t out_begin = 0, out_end = 0;
    hipEvent_t up_done = nullptr, dec_done = nullptr;
  };
  std::vector<Slice> slices;
  uint8_t *d_stream = nullptr, *d_out = nullptr;
  hipStream_t up = nullptr, dec = nullptr, down = nullptr;
  uint32_t *h_status = nullptr; // pinned, one word per slice
  std::mutex lock;              // one decode at a time per pipe: its buffers, streams and events are shared
};

void hsrans_hpipe_destroy(hsrans_hpipe *p)
{
  if (p == nullptr)
    return;
  if (p->ctx)
    (void)hipSetDevice(p->ctx->device);
  for (auto &sl : p->slices)
  {
    if (sl.dplan)
      hsrans_dplan_destroy(sl.dplan);
    if (sl.up_done)
      (void)hipEventDestroy(sl.up_done);
    if (sl.dec_done)
      (void)hipEventDestroy(sl.dec_done);
  }
  if (p->d_stream)
    (void)hipFree(p->d_stream);
  if (p->d_out)
    (void)hipFree(p->d_out);
  // (up / dec / down belong to the context)
  if (p->h_status)
    (void)hipHostFree(p->h_status);
  delete p;
}

int hsrans_hpipe_create(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, uint32_t n_slices, hsrans_hpipe **out_pipe)
try
{
  if (ctx == nullptr || out_pipe == nullptr)
    return HSRANS_E_ARG;
  *out_pipe = nullptr;
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || !plan_validate(plan, plan_size, h.stream_len, h.decoded_len) || (h.flags & kPlanWalk))
    return HSRANS_E_FORMAT;
  if (n_slices == 0) // auto: slices of >= 16 MiB of output, 2..16 (the first slice's upload is the only leg nothing overlaps with)
    n_slices = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(2, h.decoded_len >> 24));
  if (n_slices > h.n_chains)
    n_slices = h.n_chains;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_hpipe *p = new (std::nothrow) hsrans_hpipe;
  if (p == nullptr)
    return HSRANS_E_HIP;
  p->ctx = ctx;
  p->hdr = h;
  int rc = HSRANS_E_HIP;
  do
  {
    {
      std::lock_guard<std::mutex> guard(ctx->stream_lock);
      bool made = true;
      for (hipStream_t &st : ctx->pipe_streams)
        if (st == nullptr && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
          made = false;
      if (!made)
        break;
    }
    p->up = ctx->pipe_streams[0];
    p->dec = ctx->pipe_streams[1];
    p->down = ctx->pipe_streams[2];
    // (d_out, the staging buffer of the output, is allocated by the first decode that needs it: a page-locked `out` does not)
    if (hipMalloc((void **)&p->d_stream, (h.stream_len + 15) / 16 * 16 + 16) != hipSuccess ||
        hipHostMalloc((void **)&p->h_status, n_slices * 4, hipHostMallocDefault) != hipSuccess)
      break;
    // chains -> n_slices contiguous runs of (nearly) equal decoded bytes (chains are in output order)
    const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
    const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
    std::vector<uint64_t> chain_end(h.n_chains);
    uint64_t run = 0;
    for (uint32_t c = 0; c < h.n_chains; c++)
    {
      for (uint32_t i = cf[c]; i < cf[c + 1]; i++)
        run += (pc[i].flags & kPieceFill) ? pc[i].fill_len : (uint64_t)pc[i].steps * h.states + pc[i].tail;
      chain_end[c] = run;
    }
    std::vector<uint8_t> blob(plan_size);
    uint32_t first = 0;
    bool ok = true;
    for (uint32_t k = 0; k < n_slices && ok; k++)
    {
      uint32_t last = k + 1 == n_slices ? h.n_chains : (uint32_t)(std::upper_bound(chain_end.begin(), chain_end.end(), run * (k + 1) / n_slices) - chain_end.begin());
      if (last <= first)
        last = first + 1;
      if (last > h.n_chains)
        last = h.n_chains;
      if (first >= h.n_chains)
        break;
      hsrans_hpipe::Slice sl;
      const size_t bytes = plan_slice(plan, plan_size, first, last - first, blob.data(), blob.size());
      PlanHeader hs;
      ok = bytes != 0 && read_header(blob.data(), bytes, &hs) && plan_stream_ranges(plan, plan_size, first, last - first, sl.in_ranges) &&
           plan_chain_range(plan, plan_size, first, last - first, &sl.out_begin, &sl.out_end);
      if (ok)
      {
        sl.dplan = new (std::nothrow) hsrans_dplan;
        ok = sl.dplan != nullptr;
      }
      if (ok)
      {
        sl.dplan->ctx = ctx;
        ok = dplan_fill(sl.dplan, blob.data(), bytes, hs, nullptr) == HSRANS_OK && hipStreamSynchronize(nullptr) == hipSuccess &&
             hipEventCreateWithFlags(&sl.up_done, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&sl.dec_done, hipEventDisableTiming) == hipSuccess;
      }
      p->slices.push_back(sl); // (pushed even on failure so that destroy releases what exists)
      first = last;
    }
    if (!ok)
      break;
    rc = HSRANS_OK;
  } while (false);
  if (rc != HSRANS_OK)
  {
    hsrans_hpipe_destroy(p);
    return rc;
  }
  *out_pipe = p;
  return HSRANS_OK;
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

size_t hsrans_hpipe_decode(hsrans_hpipe *p, const uint8_t *in, size_t in_length, uint8_t *out, size_t out_capacity)
{
  if (p == nullptr || in == nullptr || out == nullptr || in_length < p->hdr.stream_len || out_capacity < p->hdr.decoded_len)
    return 0;
  std::lock_guard<std::mutex> guard(p->lock); // a pipe's device buffers, streams and events serve one decode at a time
  if (hipSetDevice(p->ctx->device) != hipSuccess)
    return 0;
  // Output leg: staged through d_out and copied down slice by slice on the third stream, so that the copy engines carry both PCIe
  // directions at once while the kernels run at HBM speed.  HSRANS_HPIPE_DIRECT=1 (page-locked, 4-byte-aligned `out` only): the
  // decode kernels store STRAIGHT into it instead — no device-side output buffer, no download copies; every wavefront's streaming
  // stores cross PCIe themselves.  Measured with the context's shared streams (see hsrans_ctx::pipe_streams — per-pipe streams had
  // made every comparison before that a comparison of hardware-queue assignments): 2^30 bytes 47.4 GB/s either way; 100 MB
  // 36.2-38.8 k MiB/s staged against 29.3-33.8 k direct (the 32-state kernels' 128-byte rows make poor PCIe writes), 27.4-30.9 k
  // for upload, decode, download one after the other.
  uint8_t *out_view = getenv("HSRANS_HPIPE_DIRECT") != nullptr && ((uintptr_t)out & 3) == 0 ? device_view_of_host(out, (size_t)p->hdr.decoded_len) : nullptr;
  const bool direct = out_view != nullptr;
  if (!direct && p->d_out == nullptr && hipMalloc((void **)&p->d_out, p->hdr.decoded_len + 16) != hipSuccess)
    return 0;
  bool ok = true;
  // HSRANS_HPIPE_TRACE=1: per-slice timeline on stderr (timing events around every leg; diagnostics only)
  const bool trace = getenv("HSRANS_HPIPE_TRACE") != nullptr;
  std::vector<hipEvent_t> tev;
  auto mark = [&](hipStream_t st) {
    if (!trace)
      return;
    hipEvent_t e;
    if (hipEventCreate(&e) == hipSuccess)
    {
      (void)hipEventRecord(e, st);
      tev.push_back(e);
    }
  };
  mark(p->up);
  // leg 1: every slice's stream bytes, in order, on the upload stream (a raw stream's shared histogram goes up once)
  bool head_done = false;
  for (auto &sl : p->slices)
  {
    const uint64_t *r = sl.in_ranges;
    if (r[1] > r[0] && !head_done)
    {
      ok = ok && hipMemcpyAsync(p->d_stream + r[0], in + r[0], r[1] - r[0], hipMemcpyHostToDevice, p->up) == hipSuccess;
      head_done = true;
    }
    if (r[3] > r[2])
      ok = ok && hipMemcpyAsync(p->d_stream + r[2], in + r[2], r[3] - r[2], hipMemcpyHostToDevice, p->up) == hipSuccess;
    ok = ok && hipEventRecord(sl.up_done, p->up) == hipSuccess;
    mark(p->up);
    if (!ok)
      break;
  }
  // leg 2: slice k decodes as soon as its bytes are up (leg 3, staged mode only: its output comes down as soon as it is decoded)
  for (size_t k = 0; ok && k < p->slices.size(); k++)
  {
    auto &sl = p->slices[k];
    ok = hipStreamWaitEvent(p->dec, sl.up_done, 0) == hipSuccess;
    mark(p->dec);
    ok = ok && dplan_launch(sl.dplan, p->d_stream, (size_t)p->hdr.stream_len, direct ? out_view : p->d_out, (size_t)p->hdr.decoded_len, p->dec) == HSRANS_OK;
    mark(p->dec);
    if (ok && !direct)
    {
      ok = hipEventRecord(sl.dec_done, p->dec) == hipSuccess && hipStreamWaitEvent(p->down, sl.dec_done, 0) == hipSuccess;
      if (ok && sl.out_end > sl.out_begin)
        ok = hipMemcpyAsync(out + sl.out_begin, p->d_out + sl.out_begin, sl.out_end - sl.out_begin, hipMemcpyDeviceToHost, p->down) == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(p->h_status + k, sl.dplan->d_status, 4, hipMemcpyDeviceToHost, direct ? p->dec : p->down) == hipSuccess;
  }
  // whatever happened, nothing that was queued may still be reading `in` or writing `out` when this returns
  const bool s1 = hipStreamSynchronize(p->up) == hipSuccess, s2 = hipStreamSynchronize(p->dec) == hipSuccess, s3 = hipStreamSynchronize(p->down) == hipSuccess;
  if (trace && tev.size() == 1 + 3 * p->slices.size())
  {
    const size_t K = p->slices.size();
    fprintf(stderr, "hpipe %s, %zu slices (ms from the first upload's start): ", direct ? "direct" : "staged", K);
    for (size_t k = 0; k < K; k++)
    {
      float up = 0, k0 = 0, k1 = 0;
      (void)hipEventElapsedTime(&up, tev[0], tev[1 + k]);
      (void)hipEventElapsedTime(&k0, tev[0], tev[1 + K + 2 * k]);
      (void)hipEventElapsedTime(&k1, tev[0], tev[2 + K + 2 * k]);
      fprintf(stderr, "[up %.3f kernel %.3f..%.3f] ", up, k0, k1);
    }
    fprintf(stderr, "\n");
  }
  for (hipEvent_t e : tev)
    (void)hipEventDestroy(e);
  if (!ok || !s1 || !s2 || !s3)
    return 0;
  bool good = true;
  for (size_t k = 0; k < p->slices.size(); k++)
    if (p->h_status[k] != 0)
    {
      good = false;
      (void)hipMemsetAsync(p->slices[k].dplan->d_status, 0, 4, p->down);
    }
  if (!good)
    (void)hipStreamSynchronize(p->down);
  return good ? (size_t)p->hdr.decoded_len : 0;
}

size_t hsrans_decode_host_pipelined(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out,
                                    size_t out_capacity, const uint8_t *plan, size_t plan_size, uint32_t n_slices)
{
  if (ctx == nullptr || in == nullptr || out == nullptr || plan == nullptr || !valid_codec(container, states, bits))
    return 0;
  PlanHeader h;
  if (!read_header(plan, plan_size, &h) || (int)h.container != container || (int)h.states != states || h.bits != bits || h.stream_len > in_length ||
      h.decoded_len > out_capacity)
    return 0;
  // The pipeline (slice plans on the device, streams, buffers) is kept for the plan seen last, recognised by address, size and
  // a checksum over EVERYTHING the kernels take an address or a length from — header, chain table and piece records, all of them
  // (a plan rewritten in place that differs in one words_off / out_off must not meet the old slice plans: ADVICE r3) — and, of
  // the start states behind them (most of the blob: random 32-bit words), 64 bytes of every 4 KiB and the last 64 bytes (a
  // whole-plan checksum cost more than the decode it guards: 2.5 ms for the 12.9 MB index of a 100 MB stream; the records of
  // that index are 2.5 MB).  Whatever plan a pipe holds was validated when the pipe was made.
  uint64_t sum = 0x9E3779B97F4A7C15ull ^ n_slices;
  auto mix = [&](size_t from, size_t to) {
    for (size_t i = from; i + 8 <= to; i += 8)
    {
      uint64_t v;
      memcpy(&v, plan + i, 8);
      sum = (sum ^ v) * 0x100000001B3ull + (sum >> 29);
    }
  };
  const size_t records_end = std::min(plan_size, (size_t)plan_states_off(h.n_chains, h.n_pieces));
  mix(0, records_end);
  for (size_t at = records_end & ~(size_t)7; at < plan_size; at += 4096)
    mix(at, std::min(at + 64, plan_size));
  mix(plan_size >= 64 ? plan_size - 64 : 0, plan_size);
  std::lock_guard<std::mutex> guard(ctx->lock);
  const uint64_t key[3] = {(uint64_t)(uintptr_t)plan, (uint64_t)plan_size, sum};
  if (ctx->cached_pipe == nullptr || memcmp(key, ctx->cached_pipe_key, sizeof(key)) != 0)
  {
    if (ctx->cached_pipe)
      hsrans_hpipe_destroy(ctx->cached_pipe);
    ctx->cached_pipe = nullptr;
    if (hsrans_hpipe_create(ctx, plan, plan_size, n_slices, &ctx->cached_pipe) != HSRANS_OK)
      return 0;
    memcpy(ctx->cached_pipe_key, key, sizeof(key));
  }
  return hsrans_hpipe_decode(ctx->cached_pipe, in, in_length, out, out_capacity);
}

// ---- per-device fit of the one-chain-per-wave index ----------------------------------------------------------------
// The SIMDs serve their oldest wave first and the decode loop is issue-bound, so the 8 wave classes of the one-chain-per-wave
// launch (workgroup in the grid's first / second half x wave / 4) decode at different rates and the index gives them chains of
// different lengths (hsrans_index_boundaries).  The lengths compiled in were fitted on one box; how early the second workgroup
// of a CU becomes resident, and with it the right lengths, differs from box to box by a few per cent (r03: classes of the
// second half done 2 us before the first half's on another box).  This fits them to the context's own device: synthetic
// enwik8-shaped bytes, encoded once on the host; per iteration an index at the current lengths (one host decode pass that
// records the checkpoints), a few launches whose waves leave their finish time, and every class length moved towards
// length x (mean finish / class finish) ^ 0.8.  The best lengths seen stay in the context: hsrans_index_boundaries(ctx, ...)
// and the launch info (class_weights) use them from then on.
static int calibrate_impl(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, uint32_t copies, hsrans_calibration *report)
{
  if (ctx == nullptr || bits < 10 || bits > 12 || copies < 1 || copies > 16) // (the fitted kernel is k_decode_direct<3>: 64 states, 8-byte table, one chain per wave)
    return HSRANS_E_ARG;
  if (iterations == 0)
    iterations = copies == 1 ? 4 : 7; // (a longer run starts from lengths fitted for another run length: further to go)
  if (iterations > 16)
    iterations = 16;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  const size_t n = (size_t)48 << 20;
  std::vector<uint8_t> data(n), stream(capacity(HSRANS_RAW, 64, n));
  {
    // Zipf(1.2) over 205 symbols through a 65,536-entry inverse-CDF table, xorshift64* indices: the shape of the benchmark's data
    std::vector<uint8_t> inv(65536);
    double w[205], sum = 0;
    for (int r = 0; r < 205; r++)
      sum += (w[r] = 1.0 / pow((double)(r + 1), 1.2));
    double acc = 0;
    size_t at = 0;
    for (int r = 0; r < 205; r++)
    {
      acc += w[r] / sum;
      const size_t end = r == 204 ? 65536 : (size_t)(acc * 65536.0);
      for (; at < end && at < 65536; at++)
        inv[at] = (uint8_t)((r * 37 + 11) & 0xFF); // (any fixed symbol -> byte map)
    }
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i += 4)
    {
      x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
      const uint64_t v = x * 0x2545F4914F6CDD1Dull;
      data[i] = inv[v & 0xFFFF], data[i + 1] = inv[(v >> 16) & 0xFFFF], data[i + 2] = inv[(v >> 32) & 0xFFFF], data[i + 3] = inv[v >> 48];
    }
  }
  const size_t stream_len = encode(HSRANS_RAW, 64, bits, data.data(), n, stream.data(), stream.size(), nullptr, nullptr);
  if (stream_len == 0)
    return HSRANS_E_FORMAT;
  // Scope guards first: whatever leaves this function — a return, or an exception on its way to the handler below (bad_alloc from
  // one of the vectors) — frees the device buffers, destroys the plan of the iteration in flight and puts the context's launch
  // geometry back (the iterations overwrite it with trial lengths).  The context's lock is held throughout: other entries read
  // ctx->geom (hsrans_index_boundaries, every launch_shape).
  std::lock_guard<std::mutex> calibration_guard(ctx->lock);
  struct DeviceBuffers
  {
    uint8_t *stream = nullptr, *out = nullptr;
    uint64_t *finish = nullptr;
    hsrans_dplan *dplan = nullptr;
    hsrans_batch *batch = nullptr;        // copies > 1: the iteration's batch of `copies` members and its other members' plans
    std::vector<hsrans_dplan *> more;
    void drop_iteration()
    {
      if (batch)
        hsrans_dplan_batch_destroy(batch);
      batch = nullptr;
      for (hsrans_dplan *d : more)
        hsrans_dplan_destroy(d);
      more.clear();
      if (dplan)
        hsrans_dplan_destroy(dplan);
      dplan = nullptr;
    }
    ~DeviceBuffers()
    {
      drop_iteration();
      if (stream)
        (void)hipFree(stream);
      if (out)
        (void)hipFree(out);
      if (finish)
        (void)hipFree(finish);
    }
  } dev;
  struct GeomRestore
  {
    hsrans_ctx *ctx;
    DeviceGeom saved;
    ~GeomRestore() { ctx->geom = saved; }
  } geom_restore{ctx, ctx->geom};
  uint8_t *&d_stream = dev.stream, *&d_out = dev.out;
  uint64_t *&d_finish = dev.finish;
  std::vector<uint64_t> groups(1 << 16), finish;
  std::vector<uint8_t> plan(plan_capacity_chains(HSRANS_RAW, 64, n, 1 << 14, 0));
  uint32_t best_w[8] = {}, cur_w[8];
  double best_last = 1e30, first_last = 0, first_spread = 0, best_spread = 0;
  int rc = HSRANS_E_HIP;
  do
  {
    // The launches that are measured look like the ones the fit is for: back to back, and every one on another (stream, output)
    // pair of a set larger than the 256 MB Infinity Cache — a lone launch on warm buffers shows the youngest wave class only
    // 0.3 us late, a launch of a sustained rotation 1.5 us (its prologue loads and its stores are served last), and chains
    // fitted to the former leave that class to finish the rotated launch alone.  (One pair if the device cannot spare 400 MB.)
    const size_t stream_stride = ((stream_len + 15) / 16 * 16 + 255) / 256 * 256 + 256;
    uint32_t pairs = copies > 1 ? copies + 2 : 5; // (a batch launch writes `copies` outputs: the next launch's are other buffers)
    if (hipMalloc((void **)&d_stream, pairs * stream_stride) != hipSuccess || hipMalloc((void **)&d_out, pairs * n) != hipSuccess)
    {
      (void)hipGetLastError();
      if (d_stream)
        (void)hipFree(d_stream);
      d_stream = nullptr;
      pairs = copies;
      if (hipMalloc((void **)&d_stream, pairs * stream_stride) != hipSuccess || hipMalloc((void **)&d_out, pairs * n) != hipSuccess)
        break;
    }
    bool uploaded = true;
    for (uint32_t k = 0; k < pairs && uploaded; k++)
      uploaded = hipMemcpy(d_stream + k * stream_stride, stream.data(), stream_len, hipMemcpyHostToDevice) == hipSuccess;
    if (!uploaded)
      break;
    // start from the lengths in use (the compiled-in fit, or an earlier calibration)
    {
      PlanHeader h{};
      h.states = 64, h.bits = bits, h.shared_hist = 1, h.n_chains = 1u << 30;
      const TableChoice tc = choose_table(bits, 64, true);
      const LaunchShape L = launch_shape(h, ctx->geom, true, tc.mode, 0, false, true, tc.dual);
      if (L.dual || L.waves != 16 || L.grid <= ctx->geom.num_cus) // not the launch shape the classes are defined for: nothing to fit
      {
        rc = HSRANS_E_ARG;
        break;
      }
      for (int k = 0; k < 8; k++)
        cur_w[k] = L.weights[k];
    }
    bool failed = false;
    for (uint32_t it = 0; it < iterations && !failed; it++)
    {
      ctx->geom.have_direct_weights = 1;
      ctx->geom.n_weight_sets = 0; // (the trial lengths, not an interpolation of earlier fits)
      for (int k = 0; k < 8; k++)
        ctx->geom.direct_weights[k] = cur_w[k];
      const uint64_t T = (n - 63) / 64; // whole groups (hsrans_index_boundaries)
      size_t chains;
      BatchShape bshape{};
      if (copies == 1)
        chains = direct_boundaries(ctx->geom, 64, bits, T, groups.data(), groups.size());
      else
      {
        // `copies` members of this one stream in one launch: every member indexed for its share of the wave slots (batch_boundaries)
        bshape = batch_direct_shape(ctx->geom, bits, 0);
        std::vector<uint64_t> totals(copies, T);
        chains = batch_boundaries(totals.data(), copies, 0, bshape.grid, bshape.waves, cur_w, groups.data(), groups.size());
      }
      if (chains < 2)
      {
        failed = true;
        break;
      }
      const size_t plan_len = cpu::index_build(cpu::best_level(), 1, HSRANS_RAW, 64, bits, stream.data(), stream_len, groups.data(), chains - 1, plan.data(), plan.size());
      hsrans_dplan *&dp = dev.dplan; // (owned by the guard until the iteration hands it back)
      if (plan_len == 0 || hsrans_dplan_create(ctx, plan.data(), plan_len, &dp) != HSRANS_OK)
      {
        failed = true;
        break;
      }
      for (uint32_t k = 1; k < copies && !failed; k++)
      {
        hsrans_dplan *extra = nullptr;
        if (hsrans_dplan_create(ctx, plan.data(), plan_len, &extra) != HSRANS_OK)
          failed = true;
        else
          dev.more.push_back(extra);
      }
      if (!failed && copies > 1)
      {
        std::vector<hsrans_dplan *> all{dp};
        all.insert(all.end(), dev.more.begin(), dev.more.end());
        // (the batch's own weights must be the trial lengths: HSRANS_BATCH_WEIGHTS aside, batch_direct_shape reads ctx->geom, set above)
        if (hsrans_dplan_batch_create(ctx, all.data(), copies, &dev.batch) != HSRANS_OK || dev.batch->direct.size() != 1 || !dev.batch->solo.empty())
          failed = true;
      }
      if (failed)
        break;
      const uint32_t W = copies == 1 ? (uint32_t)chains : bshape.grid * bshape.waves; // one chain per wave
      if (W > (1u << 14)) // (the finish-time buffer below is sized for 16,384 waves: twice an MI355X)
      {
        failed = true;
        break;
      }
      const size_t finish_stride = ((size_t)1 << 14) + 1; // words per launch: one finish time per wave + the first wave's entry
      const uint32_t batch = pairs == 1 ? 4 : 2 * pairs;  // launches per batch, back to back, launch l on pair l % pairs
      if (d_finish == nullptr && hipMalloc((void **)&d_finish, (size_t)batch * finish_stride * 8) != hipSuccess)
        failed = true;
      double cls_t[8] = {}, cls_n[8] = {}, last = 0;
      // Rounds of back-to-back launches; only the last one is measured.  The ones before it run until the device has been busy for
      // 20 ms (and at least twice): every iteration begins with host work (the index pass, the device plans) during which the GPU
      // idles, and a GPU that wakes from idle runs the first ~12 ms at other clocks than it then keeps (tools/settle_probe.py,
      // profiles/r04_settle.txt) — lengths fitted in that transient left the youngest class 4 % short on some boxes (round 5:
      // the classes of a 4 x 100 MB batch finished 5 us apart after a fit whose own last round had them within 0.5 us).
      const auto t_busy = std::chrono::steady_clock::now();
      bool measured = false;
      for (int round = 0; !measured && !failed; round++)
      {
        const bool settle = round < 2 || std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_busy).count() < 20.0;
        measured = !settle;
        failed = hipMemset(d_finish, 0, (size_t)batch * finish_stride * 8) != hipSuccess;
        for (uint32_t l = 0; l < batch && !failed; l++)
        {
          if (copies == 1)
          {
            dp->d_finish = d_finish + l * finish_stride;
            failed = dplan_launch(dp, d_stream + (l % pairs) * stream_stride, stream_len, d_out + (size_t)(l % pairs) * n, n, nullptr) != HSRANS_OK;
            continue;
          }
          const void *ins[16];
          void *outs[16];
          size_t in_len[16], out_cap[16];
          for (uint32_t k = 0; k < copies; k++)
          {
            const uint32_t buf = (l * copies + k) % pairs;
            ins[k] = d_stream + buf * stream_stride, in_len[k] = stream_len;
            outs[k] = d_out + (size_t)buf * n, out_cap[k] = n;
          }
          if (dev.batch->finish_owned) // (HSRANS_BATCH_STAMPS=1 gave the batch a buffer of its own: this fit uses its own)
          {
            (void)hipFree(dev.batch->d_finish);
            dev.batch->finish_owned = false;
          }
          dev.batch->d_finish = d_finish + l * finish_stride;
          failed = hsrans_decode_device_batch(ctx, dev.batch, ins, in_len, outs, out_cap, nullptr) != HSRANS_OK;
        }
        failed = hipDeviceSynchronize() != hipSuccess || failed; // (nothing may still be writing the buffers, whatever failed)
        if (failed || settle)
          continue;
        finish.resize((size_t)batch * finish_stride);
        if (hipMemcpy(finish.data(), d_finish, finish.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        {
          failed = true;
          break;
        }
        const uint32_t waves = copies == 1 ? dp->info.waves_per_block : bshape.waves, grid = copies == 1 ? dp->info.grid : bshape.grid, first_half = (grid + 1) / 2;
        if (waves != 16 || (uint64_t)grid * waves != W)
        {
          failed = true;
          break;
        }
        for (uint32_t l = 0; l < batch; l++)
        {
          const uint64_t *f = finish.data() + l * finish_stride;
          double launch_last = 0;
          for (uint32_t w = 0; w < W; w++)
          {
            const double t = (double)(f[w] - f[W]) / 100.0; // us
            const uint32_t cls = (w / waves >= first_half ? 4 : 0) + (w % waves) / 4;
            cls_t[cls] += t, cls_n[cls] += 1;
            launch_last = t > launch_last ? t : launch_last;
          }
          last += launch_last / batch; // (mean over the launches of each launch's last wave: one late wave in one launch does not decide)
        }
      }
      dp->d_finish = nullptr;
      uint32_t status_ok = hsrans_dplan_status(ctx, dp, nullptr) == HSRANS_OK;
      for (hsrans_dplan *extra : dev.more)
        status_ok = hsrans_dplan_status(ctx, extra, nullptr) == HSRANS_OK && status_ok;
      dev.drop_iteration();
      if (failed || !status_ok)
      {
        failed = true;
        break;
      }
      double mean = 0, lo = 1e30, hi = 0;
      for (int k = 0; k < 8; k++)
      {
        cls_t[k] /= cls_n[k] > 0 ? cls_n[k] : 1;
        mean += cls_t[k] / 8;
        lo = cls_t[k] < lo ? cls_t[k] : lo, hi = cls_t[k] > hi ? cls_t[k] : hi;
      }
      if (it == 0)
        first_last = last, first_spread = hi - lo;
      if (last < best_last)
      {
        best_last = last, best_spread = hi - lo;
        for (int k = 0; k < 8; k++)
          best_w[k] = cur_w[k];
      }
      if (report)
        for (int k = 0; k < 8; k++)
          report->class_finish_us_last_iteration[k] = cls_t[k];
      double nw[8], s = 0;
      for (int k = 0; k < 8; k++)
        s += (nw[k] = (double)cur_w[k] * pow(mean / cls_t[k], 0.8));
      for (int k = 0; k < 8; k++)
        cur_w[k] = (uint32_t)(nw[k] * 8000.0 / s + 0.5);
    }
    if (failed)
      break;
    rc = HSRANS_OK;
  } while (false);
  if (rc == HSRANS_OK)
  {
    // what the guard puts back: the geometry as it was, with the fitted lengths — as the default set (the one-stream fit) and as
    // the set of this run length (groups per wave)
    DeviceGeom &g = geom_restore.saved;
    if (copies == 1)
    {
      g.have_direct_weights = 1;
      for (int k = 0; k < 8; k++)
        g.direct_weights[k] = best_w[k];
    }
    const uint32_t run = (uint32_t)((uint64_t)copies * ((n - 63) / 64) / (2 * 16 * (uint64_t)g.num_cus));
    uint32_t at = 0;
    while (at < g.n_weight_sets && g.set_run[at] < run)
      at++;
    if (!(at < g.n_weight_sets && g.set_run[at] == run))
    {
      if (g.n_weight_sets == 4) // (full: the nearest one goes)
        at = at < 4 ? at : 3;
      else
      {
        for (uint32_t k = g.n_weight_sets; k > at; k--)
        {
          g.set_run[k] = g.set_run[k - 1];
          memcpy(g.set_weights[k], g.set_weights[k - 1], sizeof(g.set_weights[k]));
        }
        g.n_weight_sets++;
      }
    }
    g.set_run[at] = run;
    for (int k = 0; k < 8; k++)
      g.set_weights[at][k] = best_w[k];
    if (report)
    {
      for (int k = 0; k < 8; k++)
        report->class_weights[k] = best_w[k];
      report->last_wave_us_before = first_last, report->last_wave_us_after = best_last;
      report->class_spread_us_before = first_spread, report->class_spread_us_after = best_spread;
      report->iterations = iterations;
      report->bytes = (uint64_t)copies * n;
    }
  }
  return rc;
}

int hsrans_ctx_calibrate(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, hsrans_calibration *report)
try
{
  return calibrate_impl(ctx, bits, iterations, 1, report);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

int hsrans_ctx_calibrate_runs(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, uint32_t copies, hsrans_calibration *report)
try
{
  return calibrate_impl(ctx, bits, iterations, copies, report);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
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
