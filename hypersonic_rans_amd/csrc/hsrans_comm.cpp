// hsrans_comm.cpp — ONE stream decoded by the GPUs of a node, behind the C ABI (include/hsrans_hip.h: hsrans_comm_*, hsrans_shard_layout,
// hsrans_sharded_*, hsrans_decode_sharded): SURVEY.md §8(b) "what a C-ABI GPU replacement must add (3): context create/destroy (device id,
// pinned staging, RCCL comm)".  The reference's counterpart is the fan-out of independent blocks to its thread pool behind the
// `thread_pool *` argument of mt_rANS32x64_16w_decode_mt (src/mt_rANS32x64_16w.h:23-28, src/mt_rANS32x64_16w_decode.cpp:217-220) and the
// join in thread_pool_await (:262); here a "thread" is a GPU (one process per GPU), a block is a chain of the decode plan, and the
// join is a point-to-point exchange of the decoded ranges over xGMI (RCCL ncclSend / ncclRecv groups), pipelined behind the decode.
//
// RCCL is bound at run time (dlopen of the librccl the process already has, else the system's): libhsrans_hip.so itself has no
// dependency on it, so single-GPU callers never load a collective library, and a process that already carries one (PyTorch ships
// its own librccl.so) does not end up with two.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "hsrans_host.h"
#include "hsrans_internal.h"

namespace
{

struct Rccl
{
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  bool ok = false;
};

const Rccl &rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the library this process already has (RTLD_NOLOAD), then by name
    for (const char *name : {"librccl.so", "librccl.so.1"})
      if (r.lib == nullptr)
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (r.lib == nullptr)
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib == nullptr)
      return;
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
  });
  return r;
}

} // namespace

struct hsrans_comm
{
  hsrans_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr; // the exchange's own stream: transfers run beside the decode kernels of the caller's stream
};

struct hsrans_sharded
{
  hsrans_ctx *ctx = nullptr;
  hsrans_comm *comm = nullptr;
  uint32_t world = 1, rank = 0, parts = 1;
  int root = -1;
  std::vector<hsrans_shard> shards; // [world * parts]
  std::vector<hsrans_dplan *> part_plans; // this rank's sub-runs (null: no chains)
  uint64_t window_lo = 0, window_hi = 0, out_base = 0, out_len = 0, total = 0, stream_len = 0;
  std::vector<hipEvent_t> part_done; // sub-run k's decode has been queued up to here (the exchange's stream waits for it)
  hipEvent_t exchanged = nullptr;
  // Round 6: the sub-runs as ONE launch.  `whole` = this rank's whole run as one device plan whose group list knows the sub-runs;
  // the launch counts finished groups into d_words[0 .. parts) and publishes `seq` to sub-run k's completion word (d_words + 16 + 16 k,
  // a cache line each) when its last group is in; the exchange's stream waits for the word (hipStreamWaitValue32) instead of for an
  // event between launches.  The reference joins its pool once per stream too (mt_rANS32x64_16w_decode.cpp:262).
  hsrans_dplan *whole = nullptr;
  uint32_t *d_words = nullptr;
  uint32_t seq = 0;
};
constexpr uint32_t kWordStride = 16; // uint32s between completion words

extern "C"
{

int hsrans_comm_unique_id(uint8_t id[HSRANS_COMM_ID_BYTES])
{
  static_assert(HSRANS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");
  if (id == nullptr)
    return HSRANS_E_ARG;
  const Rccl &r = rccl();
  if (!r.ok)
    return HSRANS_E_NO_DEVICE;
  ncclUniqueId u;
  if (r.GetUniqueId(&u) != ncclSuccess)
    return HSRANS_E_HIP;
  memcpy(id, u.internal, HSRANS_COMM_ID_BYTES);
  return HSRANS_OK;
}

int hsrans_comm_create(hsrans_ctx *ctx, const uint8_t id[HSRANS_COMM_ID_BYTES], int rank, int world, hsrans_comm **out_comm)
{
  if (ctx == nullptr || id == nullptr || out_comm == nullptr || world < 1 || rank < 0 || rank >= world)
    return HSRANS_E_ARG;
  *out_comm = nullptr;
  const Rccl &r = rccl();
  if (!r.ok)
    return HSRANS_E_NO_DEVICE;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_comm *c = new (std::nothrow) hsrans_comm;
  if (c == nullptr)
    return HSRANS_E_HIP;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclUniqueId u;
  memcpy(u.internal, id, HSRANS_COMM_ID_BYTES);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || r.CommInitRank(&c->comm, world, u, rank) != ncclSuccess)
  {
    hsrans_comm_destroy(c);
    return HSRANS_E_HIP;
  }
  *out_comm = c;
  return HSRANS_OK;
}

void hsrans_comm_destroy(hsrans_comm *c)
{
  if (c == nullptr)
    return;
  if (c->ctx)
    (void)hipSetDevice(c->ctx->device);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl().ok)
    (void)rccl().CommDestroy(c->comm);
  if (c->stream)
    (void)hipStreamDestroy(c->stream);
  delete c;
}

int hsrans_comm_rank(const hsrans_comm *c) { return c ? c->rank : -1; }
int hsrans_comm_world(const hsrans_comm *c) { return c ? c->world : 0; }
int hsrans_comm_rccl_version(void)
{
  int v = 0;
  return rccl().ok && rccl().GetVersion && rccl().GetVersion(&v) == ncclSuccess ? v : 0;
}

int hsrans_shard_layout(const uint8_t *plan, size_t plan_size, uint32_t world, uint32_t parts, const double *weights, hsrans_shard *shards, uint64_t *windows)
try
{
  return hsrans::shard_layout(plan, plan_size, world, parts, weights, shards, windows);
}
catch (...)
{
  return HSRANS_E_HIP;
}

void hsrans_sharded_destroy(hsrans_sharded *s)
{
  if (s == nullptr)
    return;
  if (s->ctx)
    (void)hipSetDevice(s->ctx->device);
  for (hsrans_dplan *d : s->part_plans)
    if (d)
      hsrans_dplan_destroy(d);
  if (s->whole)
    hsrans_dplan_destroy(s->whole);
  if (s->d_words)
    (void)hipFree(s->d_words);
  for (hipEvent_t e : s->part_done)
    if (e)
      (void)hipEventDestroy(e);
  if (s->exchanged)
    (void)hipEventDestroy(s->exchanged);
  delete s;
}

static int sharded_create_impl(hsrans_ctx *ctx, hsrans_comm *comm, int rank, int world, const uint8_t *plan, size_t plan_size, uint32_t parts, const double *weights,
                               int root, hsrans_sharded **out)
{
  if (ctx == nullptr || out == nullptr || parts == 0 || parts > 64 || world < 1 || rank < 0 || rank >= world || root >= world)
    return HSRANS_E_ARG;
  *out = nullptr;
  PlanHeader h;
  if (!read_header(plan, plan_size, &h))
    return HSRANS_E_FORMAT;
  hsrans_sharded *s = new (std::nothrow) hsrans_sharded;
  if (s == nullptr)
    return HSRANS_E_HIP;
  s->ctx = ctx;
  s->comm = comm;
  s->world = (uint32_t)world;
  s->rank = (uint32_t)rank;
  s->parts = parts;
  s->root = root < 0 ? -1 : root;
  s->total = h.decoded_len;
  s->stream_len = h.stream_len;
  s->shards.resize((size_t)s->world * parts);
  std::vector<uint64_t> windows(2 * (size_t)s->world);
  int rc = hsrans_shard_layout(plan, plan_size, s->world, parts, weights, s->shards.data(), windows.data());
  if (rc == HSRANS_OK && hipSetDevice(ctx->device) != hipSuccess)
    rc = HSRANS_E_HIP;
  s->window_lo = windows[2 * s->rank];
  s->window_hi = windows[2 * s->rank + 1];
  // the output bytes this rank has to hold: everything, or (a gather to one rank, and this is another) its own range
  s->out_base = 0;
  s->out_len = s->total;
  if (s->root >= 0 && (int)s->rank != s->root)
  {
    uint64_t b = ~(uint64_t)0, e = 0;
    for (uint32_t k = 0; k < parts; k++)
    {
      const hsrans_shard &sh = s->shards[(size_t)s->rank * parts + k];
      if (sh.chain_count == 0)
        continue;
      b = std::min(b, sh.out_begin);
      e = std::max(e, sh.out_end);
    }
    s->out_base = e > b ? b : 0;
    s->out_len = e > b ? e - b : 0;
  }
  s->part_plans.assign(parts, nullptr);
  s->part_done.assign(parts, nullptr);
  std::vector<uint8_t> slice(plan_size + 1024);
  // the sub-runs as one launch: block_/mt_ plans with checkpoints (the grouped / spread kernels count their units into the sub-runs), where the
  // device can make a stream wait for a word in memory; HSRANS_SHARD_ONE_LAUNCH=0 keeps a launch per sub-run (comparison: tools/shard_projection.py)
  if (rc == HSRANS_OK && parts > 1 && parts <= kMaxLaunchParts && !(getenv("HSRANS_SHARD_ONE_LAUNCH") && atoi(getenv("HSRANS_SHARD_ONE_LAUNCH")) == 0))
  {
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, ctx->device);
    uint32_t first = 0, count = 0;
    std::vector<uint32_t> ends(parts, 0);
    for (uint32_t k = 0; k < parts; k++)
    {
      const hsrans_shard &sh = s->shards[(size_t)s->rank * parts + k];
      if (sh.chain_count != 0 && count == 0)
        first = sh.first_chain;
      if (sh.chain_count != 0 && sh.first_chain != first + count) // (sub-runs of a rank are back to back: hsrans_shard_layout)
        can_wait = 0;
      count += sh.chain_count;
      ends[k] = count; // in the slice's own chain numbers
    }
    if (can_wait && count != 0)
    {
      const size_t n = hsrans_plan_slice(plan, plan_size, first, count, slice.data(), slice.size());
      hsrans_dplan *w = nullptr;
      if (n != 0 && dplan_create_with_parts(ctx, slice.data(), n, ends, &w) == HSRANS_OK && w->n_groups != 0 && w->groups_lean && w->part_units.size() == parts &&
          hipMalloc((void **)&s->d_words, (size_t)(parts + 1) * kWordStride * 4) == hipSuccess && hipMemset(s->d_words, 0, (size_t)(parts + 1) * kWordStride * 4) == hipSuccess)
        s->whole = w;
      else
      {
        (void)hipGetLastError();
        if (w)
          hsrans_dplan_destroy(w);
        if (s->d_words)
          (void)hipFree(s->d_words);
        s->d_words = nullptr;
      }
    }
  }
  for (uint32_t k = 0; k < parts && rc == HSRANS_OK; k++)
  {
    const hsrans_shard &sh = s->shards[(size_t)s->rank * parts + k];
    if (hipEventCreateWithFlags(&s->part_done[k], hipEventDisableTiming) != hipSuccess)
      rc = HSRANS_E_HIP;
    if (sh.chain_count == 0 || rc != HSRANS_OK || s->whole != nullptr)
      continue;
    const size_t n = hsrans_plan_slice(plan, plan_size, sh.first_chain, sh.chain_count, slice.data(), slice.size());
    rc = n == 0 ? HSRANS_E_FORMAT : hsrans_dplan_create(ctx, slice.data(), n, &s->part_plans[k]);
  }
  if (rc == HSRANS_OK && hipEventCreateWithFlags(&s->exchanged, hipEventDisableTiming) != hipSuccess)
    rc = HSRANS_E_HIP;
  if (rc != HSRANS_OK)
  {
    hsrans_sharded_destroy(s);
    return rc;
  }
  *out = s;
  return HSRANS_OK;
}

int hsrans_sharded_create(hsrans_ctx *ctx, hsrans_comm *comm, const uint8_t *plan, size_t plan_size, uint32_t parts, const double *weights, int root,
                          hsrans_sharded **out)
try
{
  if (comm == nullptr || comm->ctx != ctx)
    return HSRANS_E_ARG;
  return sharded_create_impl(ctx, comm, comm->rank, comm->world, plan, plan_size, parts, weights, root, out);
}
catch (...)
{
  return HSRANS_E_HIP;
}

int hsrans_sharded_create_rank(hsrans_ctx *ctx, int rank, int world, const uint8_t *plan, size_t plan_size, uint32_t parts, const double *weights, int root,
                               hsrans_sharded **out)
try
{
  return sharded_create_impl(ctx, nullptr, rank, world, plan, plan_size, parts, weights, root, out);
}
catch (...)
{
  return HSRANS_E_HIP;
}

int hsrans_sharded_info(const hsrans_sharded *s, hsrans_sharded_info_t *info, hsrans_shard *shards, size_t shard_capacity)
{
  if (s == nullptr || info == nullptr)
    return HSRANS_E_ARG;
  info->world = s->world;
  info->rank = s->rank;
  info->parts = s->parts;
  info->root = s->root;
  info->window_begin = s->window_lo;
  info->window_end = s->window_hi;
  info->out_base = s->out_base;
  info->out_length = s->out_len;
  info->decoded_length = s->total;
  info->stream_length = s->stream_len;
  info->one_launch = s->whole != nullptr ? 1 : 0;
  info->reserved = 0;
  if (shards != nullptr)
  {
    if (shard_capacity < s->shards.size())
      return HSRANS_E_ARG;
    memcpy(shards, s->shards.data(), s->shards.size() * sizeof(hsrans_shard));
  }
  return HSRANS_OK;
}

hsrans_dplan *hsrans_sharded_part_plan(hsrans_sharded *s, uint32_t part) { return s && part < s->parts ? s->part_plans[part] : nullptr; }
hsrans_dplan *hsrans_sharded_whole_plan(hsrans_sharded *s) { return s ? s->whole : nullptr; }

// The exchange of sub-run k's ranges, queued on the communicator's own stream: one grouped batch of point-to-point operations —
// every rank that owns bytes of part k sends them to every peer that wants them (all peers, or the root only), straight from its
// output buffer into the same bytes of the receiver's; every byte crosses exactly one xGMI link.
static int post_part(hsrans_sharded *s, uint32_t k, uint8_t *out)
{
  const Rccl &r = rccl();
  hsrans_comm *c = s->comm;
  const hsrans_shard &mine = s->shards[(size_t)s->rank * s->parts + k];
  bool any = false;
  for (uint32_t peer = 0; peer < s->world && !any; peer++)
  {
    if (peer == s->rank)
      continue;
    const hsrans_shard &theirs = s->shards[(size_t)peer * s->parts + k];
    any = (mine.out_end > mine.out_begin && (s->root < 0 || s->root == (int)peer)) || (theirs.out_end > theirs.out_begin && (s->root < 0 || s->root == (int)s->rank));
  }
  if (!any)
    return HSRANS_OK;
  bool ok = r.GroupStart() == ncclSuccess;
  for (uint32_t peer = 0; peer < s->world && ok; peer++)
  {
    if (peer == s->rank)
      continue;
    if (mine.out_end > mine.out_begin && (s->root < 0 || s->root == (int)peer))
      ok = r.Send(out + (mine.out_begin - s->out_base), mine.out_end - mine.out_begin, ncclUint8, (int)peer, c->comm, c->stream) == ncclSuccess;
    const hsrans_shard &theirs = s->shards[(size_t)peer * s->parts + k];
    if (ok && theirs.out_end > theirs.out_begin && (s->root < 0 || s->root == (int)s->rank))
      ok = r.Recv(out + (theirs.out_begin - s->out_base), theirs.out_end - theirs.out_begin, ncclUint8, (int)peer, c->comm, c->stream) == ncclSuccess;
  }
  ok = r.GroupEnd() == ncclSuccess && ok;
  return ok ? HSRANS_OK : HSRANS_E_HIP;
}

int hsrans_decode_sharded(hsrans_sharded *s, const void *d_window, void *d_out, int gather, void *hip_stream)
{
  if (s == nullptr || d_out == nullptr || (d_window == nullptr && s->window_hi > s->window_lo))
    return HSRANS_E_ARG;
  hsrans_ctx *ctx = s->ctx;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  if (gather != HSRANS_SHARD_DECODE_ONLY && gather != HSRANS_SHARD_DECODE_AND_EXCHANGE && gather != HSRANS_SHARD_EXCHANGE_ONLY)
    return HSRANS_E_ARG;
  if (gather != HSRANS_SHARD_DECODE_ONLY && s->comm == nullptr && s->world > 1) // (hsrans_sharded_create_rank: one rank's share, no communicator)
    return HSRANS_E_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  hipStream_t xs = s->comm ? s->comm->stream : nullptr;
  const bool exchange = gather != HSRANS_SHARD_DECODE_ONLY && s->world > 1;
  const bool decode = gather != HSRANS_SHARD_EXCHANGE_ONLY;
  const bool i_am_root = s->root >= 0 && (int)s->rank == s->root;
  if (exchange)
  {
    // the exchange's stream starts behind whatever the caller's stream holds (the previous step may still be reading `d_out`)
    if (hipEventRecord(s->exchanged, st) != hipSuccess || hipStreamWaitEvent(xs, s->exchanged, 0) != hipSuccess)
      return HSRANS_E_HIP;
    // a receiving root posts all its receives first: none of them depends on its own decode
    if (i_am_root)
      for (uint32_t k = 0; k < s->parts; k++)
      {
        const int rc = post_part(s, k, (uint8_t *)d_out);
        if (rc != HSRANS_OK)
          return rc;
      }
  }
  if (s->whole != nullptr)
  {
    // ONE launch for all sub-runs; sub-run k's ranges go onto the links when the launch publishes its completion word
    if (decode)
    {
      PartArgs pw{};
      pw.seq = ++s->seq;
      pw.count = s->d_words;
      for (uint32_t k = 0; k < s->parts; k++)
        pw.done[k] = s->d_words + (size_t)(k + 1) * kWordStride;
      const int rc = dplan_launch_ranges(ctx, s->whole, d_window, s->window_lo, s->window_hi - s->window_lo, d_out, s->out_base, s->out_len, st, &pw);
      if (rc != HSRANS_OK)
        return rc;
    }
    for (uint32_t k = 0; k < s->parts && exchange && !i_am_root; k++)
    {
      const hsrans_shard &mine = s->shards[(size_t)s->rank * s->parts + k];
      if (decode && mine.chain_count != 0 &&
          hipStreamWaitValue32(xs, s->d_words + (size_t)(k + 1) * kWordStride, s->seq, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess)
        return HSRANS_E_HIP;
      const int rc = post_part(s, k, (uint8_t *)d_out);
      if (rc != HSRANS_OK)
        return rc;
    }
  }
  for (uint32_t k = 0; k < s->parts && s->whole == nullptr; k++)
  {
    if (decode && s->part_plans[k] != nullptr)
    {
      const int rc = hsrans_decode_device_ranges(ctx, s->part_plans[k], d_window, s->window_lo, s->window_hi - s->window_lo, d_out, s->out_base, s->out_len, st);
      if (rc != HSRANS_OK)
        return rc;
    }
    if (decode && hipEventRecord(s->part_done[k], st) != hipSuccess) // (also what hsrans_sharded_wait_part waits for)
      return HSRANS_E_HIP;
    if (!exchange || i_am_root)
      continue;
    // sub-run k's ranges go onto the links as soon as its decode is done, while sub-run k + 1 decodes
    if ((!decode && hipEventRecord(s->part_done[k], st) != hipSuccess) || hipStreamWaitEvent(xs, s->part_done[k], 0) != hipSuccess)
      return HSRANS_E_HIP;
    const int rc = post_part(s, k, (uint8_t *)d_out);
    if (rc != HSRANS_OK)
      return rc;
  }
  if (exchange) // the caller's stream continues when the transfers are done
    if (hipEventRecord(s->exchanged, xs) != hipSuccess || hipStreamWaitEvent(st, s->exchanged, 0) != hipSuccess)
      return HSRANS_E_HIP;
  return HSRANS_OK;
}

int hsrans_sharded_wait_part(hsrans_sharded *s, uint32_t part, void *hip_stream)
{
  if (s == nullptr || part >= s->parts)
    return HSRANS_E_ARG;
  if (hipSetDevice(s->ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  const hsrans_shard &mine = s->shards[(size_t)s->rank * s->parts + part];
  if (mine.chain_count == 0) // nothing to wait for
    return HSRANS_OK;
  if (s->whole != nullptr)
  {
    if (s->seq == 0) // no decode queued yet
      return HSRANS_E_ARG;
    return hipStreamWaitValue32((hipStream_t)hip_stream, s->d_words + (size_t)(part + 1) * kWordStride, s->seq, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess ? HSRANS_OK
                                                                                                                                                                 : HSRANS_E_HIP;
  }
  return hipStreamWaitEvent((hipStream_t)hip_stream, s->part_done[part], 0) == hipSuccess ? HSRANS_OK : HSRANS_E_HIP;
}

int hsrans_sharded_status(hsrans_sharded *s, void *hip_stream)
{
  if (s == nullptr)
    return HSRANS_E_ARG;
  int worst = s->whole != nullptr ? hsrans_dplan_status(s->ctx, s->whole, hip_stream) : HSRANS_OK;
  for (hsrans_dplan *d : s->part_plans)
    if (d != nullptr)
    {
      const int rc = hsrans_dplan_status(s->ctx, d, hip_stream);
      if (rc != HSRANS_OK && worst == HSRANS_OK)
        worst = rc;
    }
  return worst;
}

} // extern "C"
