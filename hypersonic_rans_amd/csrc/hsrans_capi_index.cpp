// hsrans_capi_index.cpp — sidecar indexes of existing streams: hsrans_index_build[_at], hsrans_decode_device_indexing (first decode that leaves its index behind).
// Part of the C ABI of libhsrans_hip.so (include/hsrans_hip.h); split out of hsrans_capi.cpp in round 5 by concern.
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
  // byte).  mt_ blocks (one wavefront each, in parallel) and block_ streams
  // (the walk that also reports the inline headers) stay on the GPU.
  if (container == HSRANS_RAW)
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
extern "C++" int decode_device_indexing_impl(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
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
    if (grouped)
      dplan_blocks_from_device_groups(nd, s); // (k_decode_dealt's dealing wants the blocks as chain ranges)
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


} // extern "C"
