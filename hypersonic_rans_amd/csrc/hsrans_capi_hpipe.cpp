// hsrans_capi_hpipe.cpp — host-resident streams, PCIe legs overlapped: hsrans_hpipe_*, hsrans_decode_host_pipelined.
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


} // extern "C"
