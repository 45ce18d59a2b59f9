// hsrans_queue.cpp — open-ended submission of independent streams (include/hsrans_hip.h: hsrans_queue_*).
//
// The reference's pool takes tasks as they arrive (thread_pool_add, src/thread_pool.cpp:124-133) and its benchmark loop hands it one file
// after another (src/main.cpp:841-898, through decode_with_thread_pool_wrapper :163-170).  hsrans_dplan_batch_create wants all K members up
// front; a caller that meets its streams one by one got one launch each (0.51 of the roofline for 100 MB streams instead of the batch's 0.575).
// A queue collects submissions on the host — nothing is launched by a submit — and a flush decodes everything pending with ONE batch
// launch.  What a batch costs to make (the members' piece records read back, the dealing, three allocations: about a millisecond) is
// paid once per SHAPE: the queue keeps the batches it made, and
//   * the same device plans in the same order (a loop over a set of resident streams) reuse their batch as it is;
//   * other plans of the same shapes — raw plans whose chains start at the same groups: same size and index geometry, hsrans_dplan's
//     deal_sig — reuse its dealing: only the small member table (64 bytes a member: plan arrays, decode table, status word) is rewritten,
//     asynchronously in front of the launch, from page-locked staging.
// One flush stream at a time: a batch's member table and a plan's ticket counters belong to the launches queued on one stream.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "hsrans_internal.h"

namespace
{
struct Pending
{
  hsrans_dplan *dplan;
  const void *d_stream;
  size_t stream_length;
  void *d_out;
  size_t out_capacity;
};
struct Cached
{
  hsrans_batch *batch = nullptr;
  std::vector<uint64_t> sigs;        // the members' deal signatures, in order (0: a member that can only be matched as the very same plan)
  std::vector<uint64_t> uids;        // the plans its member table currently points at (hsrans_dplan::uid: an address may come back with another plan)
  struct Shape
  {
    uint32_t n_chains, bits, states, table_mode, dual;
  };
  std::vector<Shape> shapes;         // what the dealing and the launch shape took from the plans the batch was made for
  bool retargetable = false;         // every member sits in a one-chain-per-wave launch (no grouped launch, no launch of its own)
  uint64_t last_used = 0;
};
constexpr uint32_t kCacheSize = 8, kStagingSlots = 8;
} // namespace

struct hsrans_queue
{
  hsrans_ctx *ctx = nullptr;
  uint32_t max_members = 0;
  std::vector<Pending> pending;
  std::vector<Cached> cache;
  uint64_t clock = 0;
  hsrans_queue_stats_t stats{};
  // page-locked staging for re-targeted member tables: kStagingSlots x kBatchMax records, an event behind each slot's last copy
  BatchMember *h_staging = nullptr;
  hipEvent_t staged[kStagingSlots] = {};
  uint32_t next_slot = 0;
};

static void member_record(BatchMember *bm, const hsrans_dplan *d)
{
  *bm = BatchMember{};
  bm->pieces = d->pa.pieces;
  bm->states = d->pa.states;
  bm->table = d->pa.table;
  bm->hist_copy = d->pa.hist_copy;
  bm->status = d->d_status;
  bm->hist_off = d->pa.hist_off;
  bm->n_chains = d->hdr.n_chains;
  bm->bits = d->hdr.bits;
  bm->S = d->hdr.states;
}

// points a cached batch's member tables at other plans of the same shapes (checked by the caller); asynchronous on s
static int retarget(hsrans_queue *q, Cached &c, const std::vector<hsrans_dplan *> &plans, hipStream_t s)
{
  hsrans_batch *b = c.batch;
  for (hsrans_batch::DirectLaunch &L : b->direct)
  {
    const uint32_t slot = q->next_slot++ % kStagingSlots;
    if (hipEventSynchronize(q->staged[slot]) != hipSuccess) // (the copy that last read this staging slot: long done unless 8 flushes are in flight)
      return HSRANS_E_HIP;
    BatchMember *bm = q->h_staging + (size_t)slot * kBatchMax;
    for (size_t i = 0; i < L.member_idx.size(); i++)
    {
      const hsrans_dplan *d = plans[L.member_idx[i]];
      // (what the dealing and the launch shape took from the plan it was made for must hold for this one: same chains, same table kind)
      const Cached::Shape &was = c.shapes[L.member_idx[i]];
      if (d->hdr.n_chains != was.n_chains || d->hdr.bits != was.bits || d->hdr.states != was.states || d->pa.table_mode != was.table_mode || d->pa.dual != was.dual ||
          d->pa.table == nullptr || d->pa.pieces == nullptr)
        return HSRANS_E_ARG;
      member_record(bm + i, d);
    }
    if (hipMemcpyAsync((void *)L.d_members, bm, L.member_idx.size() * sizeof(BatchMember), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipEventRecord(q->staged[slot], s) != hipSuccess)
      return HSRANS_E_HIP;
  }
  for (size_t k = 0; k < plans.size(); k++)
  {
    b->members[k] = plans[k];
    c.uids[k] = plans[k]->uid;
  }
  q->stats.retargeted++;
  return HSRANS_OK;
}

extern "C"
{

int hsrans_queue_create(hsrans_ctx *ctx, uint32_t max_members, hsrans_queue **out_queue)
try
{
  if (ctx == nullptr || out_queue == nullptr || max_members == 0 || max_members > kBatchMax)
    return HSRANS_E_ARG;
  *out_queue = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_queue *q = new (std::nothrow) hsrans_queue;
  if (q == nullptr)
    return HSRANS_E_HIP;
  q->ctx = ctx;
  q->max_members = max_members;
  q->pending.reserve(max_members);
  bool ok = hipHostMalloc((void **)&q->h_staging, (size_t)kStagingSlots * kBatchMax * sizeof(BatchMember), hipHostMallocDefault) == hipSuccess;
  for (uint32_t k = 0; k < kStagingSlots && ok; k++)
    ok = hipEventCreateWithFlags(&q->staged[k], hipEventDisableTiming) == hipSuccess;
  if (!ok)
  {
    (void)hipGetLastError();
    hsrans_queue_destroy(q);
    return HSRANS_E_HIP;
  }
  *out_queue = q;
  return HSRANS_OK;
}
catch (...)
{
  return HSRANS_E_HIP;
}

void hsrans_queue_destroy(hsrans_queue *q)
{
  if (q == nullptr)
    return;
  if (q->ctx)
    (void)hipSetDevice(q->ctx->device);
  (void)hipDeviceSynchronize(); // (a batch of the cache may still be decoding)
  for (Cached &c : q->cache)
    if (c.batch)
      hsrans_dplan_batch_destroy(c.batch);
  for (hipEvent_t e : q->staged)
    if (e)
      (void)hipEventDestroy(e);
  if (q->h_staging)
    (void)hipHostFree(q->h_staging);
  delete q;
}

int hsrans_queue_flush(hsrans_queue *q, void *hip_stream)
try
{
  if (q == nullptr)
    return HSRANS_E_ARG;
  if (q->pending.empty())
    return HSRANS_OK;
  hsrans_ctx *ctx = q->ctx;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hipStream_t s = (hipStream_t)hip_stream;
  std::vector<Pending> work;
  work.swap(q->pending); // (whatever happens below, the queue is empty afterwards: a failed flush does not come back with the next one)
  q->stats.flushes++;
  const uint32_t K = (uint32_t)work.size();
  if (K == 1) // nothing to share a launch with
  {
    q->stats.launches++;
    return hsrans_decode_device(ctx, work[0].dplan, work[0].d_stream, work[0].stream_length, work[0].d_out, work[0].out_capacity, hip_stream);
  }
  std::vector<hsrans_dplan *> plans(K);
  std::vector<uint64_t> sigs(K), uids(K);
  std::vector<const void *> ins(K);
  std::vector<void *> outs(K);
  std::vector<size_t> in_len(K), out_cap(K);
  for (uint32_t k = 0; k < K; k++)
  {
    plans[k] = work[k].dplan;
    sigs[k] = work[k].dplan->deal_sig;
    uids[k] = work[k].dplan->uid;
    ins[k] = work[k].d_stream;
    outs[k] = work[k].d_out;
    in_len[k] = work[k].stream_length;
    out_cap[k] = work[k].out_capacity;
  }
  Cached *hit = nullptr;
  bool same_plans = false;
  for (Cached &c : q->cache) // the very plans first, then their shapes
    if (c.uids == uids)
    {
      hit = &c;
      same_plans = true;
      break;
    }
  if (hit == nullptr)
    for (Cached &c : q->cache)
      if (c.retargetable && c.sigs == sigs && std::find(sigs.begin(), sigs.end(), 0ull) == sigs.end())
      {
        hit = &c;
        break;
      }
  if (hit != nullptr && !same_plans && retarget(q, *hit, plans, s) != HSRANS_OK)
    hit = nullptr; // (not the same shape after all: a batch of its own)
  if (hit == nullptr)
  {
    hsrans_batch *b = nullptr;
    const int rc = hsrans_dplan_batch_create(ctx, plans.data(), K, &b);
    if (rc != HSRANS_OK)
      return rc;
    q->stats.batches_made++;
    Cached c;
    c.batch = b;
    c.sigs = sigs;
    c.uids = uids;
    for (const hsrans_dplan *d : plans)
      c.shapes.push_back(Cached::Shape{d->hdr.n_chains, d->hdr.bits, d->hdr.states, d->pa.table_mode, d->pa.dual});
    c.retargetable = b->grouped.empty() && b->solo.empty() && !b->direct.empty();
    if (q->cache.size() >= kCacheSize) // the least recently used batch goes (hipFree inside waits for the device: nothing of it is in flight afterwards)
    {
      auto lru = std::min_element(q->cache.begin(), q->cache.end(), [](const Cached &a, const Cached &b2) { return a.last_used < b2.last_used; });
      hsrans_dplan_batch_destroy(lru->batch);
      *lru = std::move(c);
      hit = &*lru;
    }
    else
    {
      q->cache.push_back(std::move(c));
      hit = &q->cache.back();
    }
  }
  else
    q->stats.batches_reused++;
  hit->last_used = ++q->clock;
  hsrans_batch_info info{};
  (void)hsrans_dplan_batch_info(hit->batch, &info);
  q->stats.launches += info.launches;
  return hsrans_decode_device_batch(ctx, hit->batch, ins.data(), in_len.data(), outs.data(), out_cap.data(), hip_stream);
}
catch (...)
{
  return HSRANS_E_HIP;
}

int hsrans_queue_submit(hsrans_queue *q, hsrans_dplan *dplan, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, void *hip_stream)
try
{
  if (q == nullptr || dplan == nullptr || dplan->ctx != q->ctx || d_stream == nullptr || d_out == nullptr)
    return HSRANS_E_ARG;
  if (((uintptr_t)d_stream & 15) != 0 || ((uintptr_t)d_out & 3) != 0)
    return HSRANS_E_ARG;
  if (stream_length < dplan->hdr.stream_len || out_capacity < dplan->hdr.decoded_len)
    return HSRANS_E_FORMAT;
  // a plan's status word and counters belong to one member of a launch: the same plan again starts the next batch
  for (const Pending &p : q->pending)
    if (p.dplan == dplan)
    {
      const int rc = hsrans_queue_flush(q, hip_stream);
      if (rc != HSRANS_OK)
        return rc;
      break;
    }
  q->pending.push_back(Pending{dplan, d_stream, stream_length, d_out, out_capacity});
  q->stats.submitted++;
  if (q->pending.size() >= q->max_members)
    return hsrans_queue_flush(q, hip_stream);
  return HSRANS_OK;
}
catch (...)
{
  return HSRANS_E_HIP;
}

uint32_t hsrans_queue_pending(const hsrans_queue *q) { return q ? (uint32_t)q->pending.size() : 0; }

int hsrans_queue_stats(const hsrans_queue *q, hsrans_queue_stats_t *out)
{
  if (q == nullptr || out == nullptr)
    return HSRANS_E_ARG;
  *out = q->stats;
  return HSRANS_OK;
}

} // extern "C"
