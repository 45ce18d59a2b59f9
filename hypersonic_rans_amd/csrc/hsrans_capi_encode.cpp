// hsrans_capi_encode.cpp — GPU encoder entries: hsrans_encode_device (mt_, one wavefront per block), hsrans_encode_device_raw.
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
  const bool wide_hist = true; // (false: the coding wavefront counts its own block, as in rounds 1-3: 29.5 us per 64 KiB block — one wavefront's LDS atomics — against 23.5 us for the whole input by K_hist; the knob is gone)
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
  // the result words land in page-locked host memory the kernels write directly (the stream is synchronised before they are read)
  if (ctx->h_enc_result == nullptr && hipHostMalloc((void **)&ctx->h_enc_result, kEncResultWords * 8, hipHostMallocMapped) != hipSuccess)
  {
    ctx->h_enc_result = nullptr;
    return 0;
  }
  void *d_result = nullptr;
  if (hipHostGetDevicePointer(&d_result, ctx->h_enc_result, 0) != hipSuccess)
    return 0;
  ep.result = (uint64_t *)d_result;
  ep.fits = ep.image_off + nb;
  uint64_t *after = ep.image_off + nb + kEncResultWords; // (the words' old place in the meta buffer stays reserved)
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
  if (hipStreamSynchronize(s) != hipSuccess)
    return 0;
  memcpy(result, ctx->h_enc_result, sizeof(result));
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
      for (int k = 1; k < 3; k++) // the spread over the blocks: the kernel lasts as long as its slowest block
      {
        std::vector<uint64_t> d(ep.n_blocks);
        for (uint32_t b = 0; b < ep.n_blocks; b++)
          d[b] = st[b * 4 + k + 1] - st[b * 4 + k];
        std::sort(d.begin(), d.end());
        fprintf(stderr, "[hsrans encode stamps]   %s us: min %.1f  p10 %.1f  p25 %.1f  median %.1f  p75 %.1f  p90 %.1f  max %.1f\n", k == 1 ? "normalise+table" : "rANS pass      ",
                d[0] / 100.0, d[d.size() / 10] / 100.0, d[d.size() / 4] / 100.0, d[d.size() / 2] / 100.0, d[d.size() * 3 / 4] / 100.0, d[d.size() * 9 / 10] / 100.0, d.back() / 100.0);
      }
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
  // ONE device allocation for the status word, the ticket counters of the dynamic group order (as dplan_fill; without them the launch
  // falls back to the static order), the plan and the group list, one memset over the first three, one synchronisation (round 4:
  // up to four hipMalloc, three memsets, three synchronisations — 0.14 ms on top of a 0.2 ms encode)
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cbytes = grouped ? (size_t)kCounterSets * kDynQueues * kDynQueueStride * 8 : 0;
  const size_t gbytes = grouped ? nb * ep.group_split * sizeof(Group) : 0;
  const size_t off_counters = 256, off_plan = off_counters + up256(cbytes), off_groups = off_plan + up256(bytes);
  const size_t arena = off_groups + up256(gbytes);
  bool ok = hipMalloc((void **)&d->d_arena, arena) == hipSuccess;
  if (ok)
  {
    d->d_arena_cap = arena;
    d->arena_used = arena;
    d->d_status = (uint32_t *)d->d_arena;
    d->d_counters = cbytes ? (unsigned long long *)(d->d_arena + off_counters) : nullptr;
    d->d_plan = d->d_arena + off_plan;
    d->d_plan_cap = bytes;
    d->d_groups = gbytes ? d->d_arena + off_groups : nullptr;
    d->d_groups_cap = gbytes;
    ok = hipMemsetAsync(d->d_arena, 0, off_plan + bytes, s) == hipSuccess && hipMemcpyAsync(d->d_plan, &h, sizeof(h), hipMemcpyHostToDevice, s) == hipSuccess;
  }
  else
    (void)hipGetLastError();
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
  dplan_blocks_from_device_groups(d, s); // (k_decode_dealt's dealing wants the blocks as chain ranges: 32 bytes a group, once)
  *out_dplan = d;
  return total;
}


} // extern "C"
