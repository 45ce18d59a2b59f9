// C ABI of libhsrans_hip.so (declared in include/hsrans_hip.h).  Nothing here decodes on the CPU: every decode entry
// ends in a launch of the gfx950 kernels in hsrans_kernels.hip and fails when no usable device exists.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/hsrans_hip.h"
#include "hsrans_host.h"
#include "hsrans_encode.h"
#include "hsrans_kernels.h"

using namespace hsrans;

struct hsrans_ctx
{
  int device = 0;
  char name[256] = {};
  std::mutex lock; // guards the staging buffers of the host-pointer entry
  hipStream_t stream = nullptr;
  uint8_t *d_in = nullptr;
  size_t d_in_cap = 0;
  uint8_t *d_out = nullptr;
  size_t d_out_cap = 0;
  uint8_t *d_plan = nullptr;
  size_t d_plan_cap = 0;
  uint32_t *d_status = nullptr;
  uint8_t *d_enc_scratch = nullptr; // hsrans_encode_device: block images, then {image_bytes, image_off, result}
  size_t d_enc_scratch_cap = 0;
  uint8_t *d_enc_meta = nullptr;
  size_t d_enc_meta_cap = 0;
  uint8_t *d_enc_ck = nullptr; // checkpoint states / cursors of the blocks being encoded
  size_t d_enc_ck_cap = 0;
};

struct hsrans_dplan
{
  hsrans_ctx *ctx = nullptr;
  PlanHeader hdr{};
  uint8_t *d_plan = nullptr;
  uint32_t *d_status = nullptr;
  size_t plan_bytes = 0;
  uint64_t *d_stamps = nullptr; // diagnostics (HSRANS_DEBUG_STAMPS=1)
  unsigned long long *d_counters = nullptr; // persistent launches: monotonic queue heads
  uint2 *d_table = nullptr;                 // host-built decode table (plans that carry their histogram)
  Group *d_groups = nullptr;                // grouped launches (block_/mt_ plans with checkpoints)
  uint32_t n_groups = 0;
  PersistentArgs pa{};
  LaunchInfo info{};
};

namespace
{
constexpr size_t kStampWaves = 16384;

bool grow(uint8_t **p, size_t *cap, size_t need)
{
  if (need <= *cap)
    return true;
  if (*p)
    (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 8 + 4096;
  if (hipMalloc((void **)p, want) != hipSuccess)
    return false;
  *cap = want;
  return true;
}

bool read_header(const uint8_t *plan, size_t size, PlanHeader *h)
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return false;
  memcpy(h, plan, sizeof(PlanHeader));
  return memcmp(h->magic, "HSRPLAN1", 8) == 0;
}
} // namespace

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
{
  return encode(container, states, bits, in, length, out, out_capacity, hist, nullptr);
}

size_t hsrans_encode_ex(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t out_capacity,
                        const hsrans_hist *hist, hsrans_encode_opts *opts)
{
  if (opts)
    opts->plan_size = 0;
  return encode(container, states, bits, in, length, out, out_capacity, hist, opts);
}

size_t hsrans_plan_capacity(int container, int states, size_t decoded_size, uint32_t index_interval, uint32_t block_size)
{
  if (!valid_codec(container, states, 10))
    return 0;
  return plan_capacity(container, states, decoded_size, index_interval, block_size);
}

size_t hsrans_plan_build(int container, int states, uint32_t bits, const uint8_t *stream, size_t stream_length, size_t out_capacity, uint8_t *plan_out,
                         size_t plan_capacity)
{
  if (plan_out == nullptr)
    return 0;
  return plan_build(container, states, bits, stream, stream_length, out_capacity, plan_out, plan_capacity);
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
{
  if (out == nullptr)
    return 0;
  return plan_slice(plan, plan_size, first_chain, chain_count, out, out_capacity);
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
  if (prepare_kernels() != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&ctx->d_status, 64) != hipSuccess)
  {
    hsrans_ctx_destroy(ctx);
    return HSRANS_E_HIP;
  }
  *out_ctx = ctx;
  return HSRANS_OK;
}

void hsrans_ctx_destroy(hsrans_ctx *ctx)
{
  if (ctx == nullptr)
    return;
  (void)hipSetDevice(ctx->device);
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
  delete ctx;
}

const char *hsrans_ctx_device_name(const hsrans_ctx *ctx) { return ctx ? ctx->name : ""; }

size_t hsrans_decode_host(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out, size_t out_capacity,
                          const uint8_t *plan, size_t plan_size)
{
  if (ctx == nullptr || in == nullptr || out == nullptr || !valid_codec(container, states, bits))
    return 0;

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
    own_plan.resize(plan_capacity(container, states, (size_t)out_len, 0, 0));
    const size_t n = plan_build(container, states, bits, in, in_length, out_capacity, own_plan.data(), own_plan.size());
    if (n == 0)
      return 0;
    plan = own_plan.data();
    plan_size = n;
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
  if (!grow(&ctx->d_in, &ctx->d_in_cap, in_pad) || !grow(&ctx->d_out, &ctx->d_out_cap, (size_t)h.decoded_len + 16) ||
      !grow(&ctx->d_plan, &ctx->d_plan_cap, plan_size))
    return 0;
  hipStream_t s = ctx->stream;
  if (hipMemcpyAsync(ctx->d_in, in, in_length, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(ctx->d_plan, plan, plan_size, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(ctx->d_status, 0, 4, s) != hipSuccess)
    return 0;
  KParams kp{};
  kp.stream = ctx->d_in;
  kp.stream_len = in_length;
  kp.out = ctx->d_out;
  kp.out_cap = h.decoded_len;
  kp.plan = ctx->d_plan;
  kp.status = ctx->d_status;
  if (launch_decode(kp, h, s, nullptr) != hipSuccess)
    return 0;
  uint32_t status = 0xFFFFFFFF;
  if (hipMemcpyAsync(out, ctx->d_out, (size_t)h.decoded_len, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(&status, ctx->d_status, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return 0;
  return status == 0 ? (size_t)h.decoded_len : 0;
}

int hsrans_dplan_create(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, hsrans_dplan **out_dplan)
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
  d->hdr = h;
  d->plan_bytes = plan_size;
  if (hipMalloc((void **)&d->d_plan, plan_size) != hipSuccess || hipMalloc((void **)&d->d_status, 64) != hipSuccess ||
      hipMemcpy(d->d_plan, plan, plan_size, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d->d_status, 0, 64) != hipSuccess)
  {
    hsrans_dplan_destroy(d);
    return HSRANS_E_HIP;
  }
  if ((h.flags & kPlanMergeable) && h.container == HSRANS_RAW && h.interval != 0)
  {
    // persistent launch arguments, taken from the plan once (hsrans_kernels.h PersistentArgs)
    const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
    const Piece &first = pc[0], &last = pc[h.n_pieces - 1];
    bool uniform = h.n_pieces == h.n_chains;
    for (uint32_t i = 0; uniform && i + 1 < h.n_pieces; i++)
      uniform = pc[i].steps == h.interval && pc[i].state_idx == i;
    uniform = uniform && last.steps >= 1 && last.steps <= h.interval && last.state_idx == h.n_pieces - 1;
    if (uniform && hipMalloc((void **)&d->d_counters, kDynQueues * kDynQueueStride * 8) == hipSuccess &&
        hipMemset(d->d_counters, 0, kDynQueues * kDynQueueStride * 8) == hipSuccess)
    {
      d->pa.pieces = (const Piece *)(d->d_plan + plan_pieces_off(h.n_chains));
      d->pa.states = (const uint32_t *)(d->d_plan + plan_states_off(h.n_chains, h.n_pieces));
      d->pa.n_chains = h.n_chains;
      d->pa.interval = h.interval;
      d->pa.S = h.states;
      d->pa.bits = h.bits;
      d->pa.out_base = first.out_off;
      d->pa.steps_total = (uint64_t)(h.n_chains - 1) * h.interval + last.steps;
      d->pa.hist_off = h.aux_off;
      d->pa.tail = last.tail;
      d->pa.counters = d->d_counters;
      if ((h.flags & kPlanHasHist) && h.bits <= pack64_max_bits())
      {
        // decode table for the shared-table kernel (MODE 3): {freq | sym << 24, slot - cumul} per slot, the same
        // entries build_table<kModePack64> produces (hist.cpp:291-306 / :308-324 for the sum check)
        const uint16_t *counts = (const uint16_t *)(plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
        const uint32_t total = 1u << h.bits;
        std::vector<uint2> tab(total);
        uint32_t cum = 0;
        for (uint32_t s = 0; s < 256; s++)
        {
          for (uint32_t k = 0; k < counts[s] && cum + k < total; k++)
            tab[cum + k] = make_uint2((uint32_t)counts[s] | (s << 24), k);
          cum += counts[s];
        }
        if (cum != total)
        {
          hsrans_dplan_destroy(d);
          return HSRANS_E_FORMAT;
        }
        if (hipMalloc((void **)&d->d_table, total * sizeof(uint2)) == hipSuccess &&
            hipMemcpy(d->d_table, tab.data(), total * sizeof(uint2), hipMemcpyHostToDevice) == hipSuccess)
        {
          d->pa.table = d->d_table;
          d->pa.table_mode = 3;
          d->pa.hist_copy = (const uint16_t *)(d->d_plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
        }
      }
      else if ((h.flags & kPlanHasHist) && h.bits >= 13 && h.states == 64 && getenv("HSRANS_NO_COARSE_TABLE") == nullptr)
      {
        // wider histograms: the coarse + fine table pair (kModeCoarse), 36 / 40 / 48 KiB instead of 64 / 128 / 256 KiB
        const uint16_t *counts = (const uint16_t *)(plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
        std::vector<uint2> tab(coarse_table_entries(h.bits));
        if (build_coarse_table(counts, h.bits, tab.data(), tab.size()) == 0)
        {
          hsrans_dplan_destroy(d);
          return HSRANS_E_FORMAT;
        }
        if (hipMalloc((void **)&d->d_table, tab.size() * sizeof(uint2)) == hipSuccess &&
            hipMemcpy(d->d_table, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice) == hipSuccess)
        {
          d->pa.table = d->d_table;
          d->pa.table_mode = 4;
          d->pa.hist_copy = (const uint16_t *)(d->d_plan + plan_hist_off(h.n_chains, h.n_pieces, h.states));
        }
      }
    }
  }
  if (!(h.flags & (kPlanWalk | kPlanMergeable)) && h.n_chains > 1)
  {
    // group consecutive chains that decode with the same histogram (= the chains of one block_/mt_ block)
    const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
    const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
    std::vector<Group> groups;
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
          if (!(cf[ch] - cf[ch - 1] == 1 && q.tail == 0 && q.out_off + (uint64_t)q.steps * h.states == p.out_off && q.words_off <= p.words_off))
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
        g.flags = fill ? kGroupFill : (single ? kGroupMergeable : 0);
        g.hist_off = fill ? 0 : p.hist_off;
        g.words_end = h.stream_len;
        // the previous rANS group's words end no later than this group's histogram / header
        if (!fill && !groups.empty())
          for (size_t k = groups.size(); k-- > 0 && groups[k].words_end == h.stream_len;)
            groups[k].words_end = p.hist_off;
        groups.push_back(g);
      }
    }
    // Few, large blocks would leave workgroup slots empty (one workgroup per group): cut mergeable groups into parts of
    // >= 128 chains (8 per wave: below that the per-part prologue costs more than the idle slots) while there are fewer
    // groups than resident workgroups.  A part is a group of its own: same histogram, a sub-range of the chains, and its
    // words end where the next part's first chain starts reading.  Measured: 382 groups of 128 chains are best left alone.
    const size_t want = (size_t)resident_workgroups_hint();
    if (groups.size() < h.n_chains && groups.size() < want)
    {
      const uint32_t k_max = (uint32_t)((want + groups.size() - 1) / groups.size());
      std::vector<Group> parts;
      for (const Group &g : groups)
      {
        uint32_t k = (g.flags & kGroupMergeable) ? std::min(k_max, (g.count + 143) / 144) : 1;
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
          q.count = hi - lo;
          if (part + 1 < k)
            q.words_end = pc[cf[g.begin + hi]].words_off;
          parts.push_back(q);
        }
      }
      groups.swap(parts);
    }
    if (groups.size() < h.n_chains && hipMalloc((void **)&d->d_groups, groups.size() * sizeof(Group)) == hipSuccess &&
        hipMemcpy(d->d_groups, groups.data(), groups.size() * sizeof(Group), hipMemcpyHostToDevice) == hipSuccess)
      d->n_groups = (uint32_t)groups.size();
  }
  if (getenv("HSRANS_DEBUG_STAMPS") && hipMalloc((void **)&d->d_stamps, kStampWaves * 4 * 8) == hipSuccess)
    (void)hipMemset(d->d_stamps, 0, kStampWaves * 4 * 8);
  *out_dplan = d;
  return HSRANS_OK;
}

size_t hsrans_debug_read_stamps(hsrans_dplan *d, uint64_t *out, size_t capacity_u64)
{
  if (d == nullptr || d->d_stamps == nullptr || out == nullptr)
    return 0;
  const size_t n = capacity_u64 < kStampWaves * 4 ? capacity_u64 : kStampWaves * 4;
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
  delete d;
}

int hsrans_decode_device(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, void *hip_stream)
{
  if (ctx == nullptr || d == nullptr || d_stream == nullptr || d_out == nullptr)
    return HSRANS_E_ARG;
  if (((uintptr_t)d_stream & 15) != 0 || ((uintptr_t)d_out & 3) != 0)
    return HSRANS_E_ARG;
  if (stream_length < d->hdr.stream_len || out_capacity < d->hdr.decoded_len)
    return HSRANS_E_FORMAT;
  hipStream_t s = (hipStream_t)hip_stream;
  // the status word is sticky: kernels only ever OR error bits into it and hsrans_dplan_status() clears it after
  // reporting, so the launch path is exactly one kernel node (no memset node in front of it)
  KParams kp{};
  kp.stream = (const uint8_t *)d_stream;
  kp.stream_len = stream_length;
  kp.out = (uint8_t *)d_out;
  kp.out_cap = out_capacity;
  kp.plan = d->d_plan;
  kp.status = d->d_status;
  kp.stamps = d->d_stamps;
  kp.pa = d->pa;
  if (d->n_groups)
  {
    kp.groups = d->d_groups;
    kp.n_groups = d->n_groups;
  }
  return launch_decode(kp, d->hdr, s, &d->info) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
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
  return HSRANS_OK;
}

size_t hsrans_encode_device(hsrans_ctx *ctx, int container, int states, uint32_t bits, const void *d_in, size_t length, void *d_out, size_t out_capacity,
                            uint32_t block_size, uint32_t index_interval, void *hip_stream, hsrans_dplan **out_dplan)
{
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
  const size_t meta_bytes = (nb * 2 + kEncResultWords) * 8 + nb * 2 * 4 + (stamps ? nb * 4 * 8 : 0) + 64;
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
  ep.ck_states = (uint32_t *)ctx->d_enc_ck;
  ep.ck_pos = ep.ck_states + ck_slots * ep.S;
  hipStream_t s = (hipStream_t)hip_stream;
  uint64_t result[kEncResultWords] = {};
  if (launch_encode(ep, s) != hipSuccess)
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
  const size_t want = (size_t)resident_workgroups_hint();
  ep.group_split = 1;
  if (grouped && nb < want)
    ep.group_split = (uint32_t)std::max<size_t>(1, std::min<size_t>({(want + nb - 1) / nb, (size_t)(ep.max_ck + 1) / 128, (size_t)64}));
  bool ok = hipMalloc((void **)&d->d_plan, bytes) == hipSuccess && hipMalloc((void **)&d->d_status, 64) == hipSuccess &&
            (!grouped || hipMalloc((void **)&d->d_groups, nb * ep.group_split * sizeof(Group)) == hipSuccess) && hipMemsetAsync(d->d_plan, 0, bytes, s) == hipSuccess &&
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
  d->n_groups = grouped ? ep.n_blocks * ep.group_split : 0;
  *out_dplan = d;
  return total;
}

size_t hsrans_index_build(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint32_t index_interval,
                          uint8_t *plan_out, size_t plan_capacity)
{
  // One pass over an existing stream that records {states, read cursor} every `index_interval` groups inside every rANS
  // piece of the stream's own plan (raw: one sequential wavefront; mt_: one wavefront per block, in parallel); the
  // checkpoints then become additional chains.  A block_ stream is one chain with inline headers (the position of a block's
  // header is only known once the block before it is decoded): the single wavefront that walks it also reports every block
  // header it meets and the states it enters the block with, and the plan gets one chain per block plus the checkpoints.
  if (ctx == nullptr || in == nullptr || plan_out == nullptr || !valid_codec(container, states, bits))
    return 0;
  if (index_interval == 0 || index_interval % 4 != 0 || in_length < 16)
    return 0;
  uint64_t out_len;
  memcpy(&out_len, in, 8);
  std::vector<uint8_t> base(hsrans::plan_capacity(container, states, (size_t)out_len, 0, 0));
  const size_t base_size = plan_build(container, states, bits, in, in_length, (size_t)out_len, base.data(), base.size());
  if (base_size == 0)
    return 0;
  PlanHeader h;
  memcpy(&h, base.data(), sizeof(h));
  const uint32_t *cf0 = (const uint32_t *)(base.data() + plan_chain_first_off());
  const Piece *pc0 = (const Piece *)(base.data() + plan_pieces_off(h.n_chains));
  const uint32_t *st0 = (const uint32_t *)(base.data() + plan_states_off(h.n_chains, h.n_pieces));
  const uint32_t S = (uint32_t)states;
  const bool walk = (h.flags & kPlanWalk) != 0;
  if (!walk && h.n_pieces != h.n_chains) // the planner only produces single-piece chains for raw and mt_
    return 0;
  const uint64_t n_ck = out_len / S / index_interval + 2;
  // block_: room for blocks of >= 4 KiB on average (the reference's smallest block is 32 KiB, block_rANS32x64_16w_encode.cpp:21-39)
  const uint64_t max_blocks = walk ? out_len / 4096 + 16 : 0;

  std::lock_guard<std::mutex> guard(ctx->lock);
  if (hipSetDevice(ctx->device) != hipSuccess)
    return 0;
  const size_t in_pad = (in_length + 15) / 16 * 16;
  if (!grow(&ctx->d_in, &ctx->d_in_cap, in_pad) || !grow(&ctx->d_out, &ctx->d_out_cap, (size_t)out_len + 16) || !grow(&ctx->d_plan, &ctx->d_plan_cap, base_size))
    return 0;
  uint32_t *d_ck_states = nullptr;
  uint64_t *d_ck_words = nullptr;
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
    kp.walk_blocks = d_walk_blocks;
    kp.walk_states = d_walk_states;
    kp.walk_count = d_walk_count;
    kp.walk_max_blocks = (uint32_t)(max_blocks > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_blocks);
    PlanHeader hl = h;
    hl.shared_hist = 0; // private tables: every chain of the pass builds its own (raw has one chain, mt_ one per block)
    if (launch_decode(kp, hl, s, nullptr) != hipSuccess)
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
    else
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
    result = pb.serialize(plan_out, plan_capacity);
  } while (false);
  if (d_ck_states)
    (void)hipFree(d_ck_states);
  if (d_ck_words)
    (void)hipFree(d_ck_words);
  if (d_walk_blocks)
    (void)hipFree(d_walk_blocks);
  if (d_walk_states)
    (void)hipFree(d_walk_states);
  if (d_walk_count)
    (void)hipFree(d_walk_count);
  return result;
}

} // extern "C"
