// hsrans_batch.cpp — K independent streams decoded by ONE launch (include/hsrans_hip.h: hsrans_dplan_batch_*, hsrans_decode_device_batch).
//
// The reference's analogue is its pool of independent work items: a task per mt_ block (mt_rANS32x64_16w_decode.cpp:182-224), a file
// after another in its benchmark loop (main.cpp:841-898).  Here the pool is the device's resident wave slots.  This file is the host
// side: which members of a batch can share a launch, how the launch's wave slots are dealt to them (batch_deal), and the launch itself.
// The device side is kernels_batch.h (one-chain-per-wave form) and run_grouped's Group::member (block_/mt_ plans with checkpoints).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "hsrans_batch.h"
#include "hsrans_internal.h"


// ---------------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------------
namespace
{
// a plan the one-chain-per-wave batch kernel can take: raw, mergeable, 64 states, the 8-byte host-built table (bits <= 12), one chain per wave
bool direct_eligible(const hsrans_dplan *d)
{
  if (d->pa.pieces == nullptr || d->pa.table == nullptr || (d->hdr.flags & kPlanMergeable) == 0 || d->hdr.container != HSRANS_RAW || d->hdr.n_chains < 1)
    return false;
  if (d->hdr.bits <= 12) // the 8-byte table, one chain per wave (64 states) or two halves per wave (32 states)
    return d->pa.table_mode == 3 && d->pa.dual == 0 && (d->hdr.states == 64 || d->hdr.states == 32);
  // 13-15 bits, 64 states: k_decode_batch_dual takes the 8-byte table at 13 bits and the rank table at 14 / 15
  return d->hdr.states == 64 && d->hdr.bits <= 15 && d->pa.table_mode == (d->hdr.bits == 13 ? 3u : 4u);
}
// a plan the grouped batch kernel can take: block_/mt_ with checkpoints whose groups are all mergeable runs or fills, 64 states, <= 12 bits
bool grouped_eligible(const hsrans_dplan *d)
{
  return d->n_groups != 0 && d->groups_lean && d->d_groups != nullptr && d->hdr.states == 64 && d->hdr.bits <= 12 && d->hdr.n_pieces != 0;
}
} // namespace

extern "C"
{

void hsrans_dplan_batch_destroy(hsrans_batch *b)
{
  if (b == nullptr)
    return;
  if (b->ctx)
    (void)hipSetDevice(b->ctx->device);
  if (b->d_arena)
    (void)hipFree(b->d_arena);
  if (b->d_finish && b->finish_owned)
    (void)hipFree(b->d_finish);
  for (hsrans_batch::GroupedLaunch &g : b->grouped)
  {
    if (g.d_members)
      (void)hipFree((void *)g.d_members); // (members, groups and tickets are one allocation)
    delete g.epoch;
  }
  delete b;
}

int hsrans_dplan_batch_create(hsrans_ctx *ctx, hsrans_dplan *const *dplans, uint32_t count, hsrans_batch **out_batch)
try
{
  if (ctx == nullptr || dplans == nullptr || out_batch == nullptr || count == 0)
    return HSRANS_E_ARG;
  *out_batch = nullptr;
  for (uint32_t k = 0; k < count; k++)
  {
    if (dplans[k] == nullptr || dplans[k]->ctx != ctx)
      return HSRANS_E_ARG;
    for (uint32_t j = 0; j < k; j++) // (a plan's status word and ticket counters belong to one member)
      if (dplans[j] == dplans[k])
        return HSRANS_E_ARG;
  }
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hsrans_batch *b = new (std::nothrow) hsrans_batch;
  if (b == nullptr)
    return HSRANS_E_HIP;
  struct Guard // (an exception on its way to the handler below — bad_alloc from one of the vectors — must not leak the batch and its device memory)
  {
    hsrans_batch *b;
    bool armed = true;
    ~Guard()
    {
      if (armed)
        hsrans_dplan_batch_destroy(b);
    }
  } guard{b};
  b->ctx = ctx;
  b->members.assign(dplans, dplans + count);
  b->order_run.assign(count, 0);
  std::vector<uint32_t> eligible, gr_eligible;
  for (uint32_t k = 0; k < count; k++)
    (direct_eligible(dplans[k]) ? eligible : grouped_eligible(dplans[k]) ? gr_eligible : b->solo).push_back(k);
  // grouped members: one launch per histogram width (the LDS layout follows the width), at most kBatchMax members each; a lone member
  // of its width keeps its own launch (its own dealing, incl. k_decode_spread)
  for (uint32_t bits = 10; bits <= 12; bits++)
  {
    std::vector<uint32_t> of_width;
    for (uint32_t k : gr_eligible)
      if (dplans[k]->hdr.bits == bits)
        of_width.push_back(k);
    if (of_width.size() == 1)
      b->solo.push_back(of_width[0]);
    if (of_width.size() < 2)
      continue;
    const size_t n_gl = (of_width.size() + kBatchMax - 1) / kBatchMax;
    for (size_t l = 0; l < n_gl; l++)
    {
      b->grouped.emplace_back(); // (owned by the batch from the start: whatever fails below, the batch's destructor frees it)
      hsrans_batch::GroupedLaunch &G = b->grouped.back();
      G.member_idx.assign(of_width.begin() + of_width.size() * l / n_gl, of_width.begin() + of_width.size() * (l + 1) / n_gl);
      G.bits = bits;
      std::vector<Group> all;
      std::vector<GroupMember> gm(G.member_idx.size());
      uint64_t chains = 0;
      for (size_t i = 0; i < G.member_idx.size(); i++)
      {
        const hsrans_dplan *d = dplans[G.member_idx[i]];
        std::vector<Group> mine(d->n_groups);
        if (hipMemcpy(mine.data(), d->d_groups, (size_t)d->n_groups * sizeof(Group), hipMemcpyDeviceToHost) != hipSuccess)
        {
          return HSRANS_E_HIP;
        }
        for (Group &g : mine)
          g.flags = (g.flags & ((1u << kGroupMemberShift) - 1)) | ((uint32_t)i << kGroupMemberShift);
        all.insert(all.end(), mine.begin(), mine.end());
        chains += d->hdr.n_chains;
        gm[i].chain_first = (const uint32_t *)(d->d_plan + plan_chain_first_off());
        gm[i].pieces = (const Piece *)(d->d_plan + plan_pieces_off(d->hdr.n_chains));
        gm[i].states = (const uint32_t *)(d->d_plan + plan_states_off(d->hdr.n_chains, d->hdr.n_pieces));
        gm[i].status = d->d_status;
      }
      // longest groups first: the launch hands groups out in list order (the first `grid` statically, the rest by ticket), and it
      // finishes evenly when what is handed out last is short
      std::stable_sort(all.begin(), all.end(), [](const Group &x, const Group &y) { return x.count > y.count; });
      G.n_groups = (uint32_t)all.size();
      G.shape = batch_grouped_shape(ctx->geom, bits, G.n_groups, chains);
      const size_t counter_bytes = (size_t)kCounterSets * kDynQueueStride * 8;
      auto up256g = [](size_t v) { return (v + 255) & ~(size_t)255; };
      const size_t bytes = up256g(gm.size() * sizeof(GroupMember)) + up256g(all.size() * sizeof(Group)) + counter_bytes;
      uint8_t *dev = nullptr;
      if (hipMalloc((void **)&dev, bytes) != hipSuccess)
      {
        (void)hipGetLastError();
        return HSRANS_E_HIP;
      }
      G.d_members = (const GroupMember *)dev;
      G.d_groups = (const Group *)(dev + up256g(gm.size() * sizeof(GroupMember)));
      G.d_tickets = (unsigned long long *)(dev + up256g(gm.size() * sizeof(GroupMember)) + up256g(all.size() * sizeof(Group)));
      G.epoch = new (std::nothrow) std::atomic<uint32_t>(0);
      const bool ok = G.epoch != nullptr && hipMemset(dev, 0, bytes) == hipSuccess && hipMemcpy(dev, gm.data(), gm.size() * sizeof(GroupMember), hipMemcpyHostToDevice) == hipSuccess &&
                      hipMemcpy((void *)G.d_groups, all.data(), all.size() * sizeof(Group), hipMemcpyHostToDevice) == hipSuccess;
      if (!ok)
      {
        return HSRANS_E_HIP;
      }
    }
  }
  std::sort(b->solo.begin(), b->solo.end());
  // 64- and 32-state members take launches of their own kind (k_decode_batch / k_decode_batch_pair); a lone member of its kind keeps a
  // launch of its own: the same thing, with the plan's own dealing
  std::vector<std::vector<uint32_t>> launch_members;
  auto kind_of = [&](uint32_t k) { // which shared kernel a member runs on (hsrans_kernels.h: kBatch*)
    const PlanHeader &ph = dplans[k]->hdr;
    return ph.states == 32 ? kBatchPair : ph.bits <= 12 ? kBatchDirect : ph.bits == 13 ? kBatchDualPack : kBatchDualRank;
  };
  for (uint32_t kind : {kBatchDirect, kBatchPair, kBatchDualPack, kBatchDualRank})
  {
    std::vector<uint32_t> of_kind;
    for (uint32_t k : eligible)
      if (kind_of(k) == kind)
        of_kind.push_back(k);
    if (of_kind.size() == 1)
      b->solo.push_back(of_kind[0]);
    if (of_kind.size() < 2)
      continue;
    const size_t n_l = (of_kind.size() + kBatchMax - 1) / kBatchMax;
    for (size_t l = 0; l < n_l; l++) // (launches of nearly equal member counts rather than 32 + the rest)
      launch_members.emplace_back(of_kind.begin() + of_kind.size() * l / n_l, of_kind.begin() + of_kind.size() * (l + 1) / n_l);
  }
  std::sort(b->solo.begin(), b->solo.end());
  const size_t n_launches = launch_members.size();
  // one allocation: per launch its member records and its slot table
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t arena = 0;
  std::vector<std::vector<uint8_t>> host_blobs;
  for (size_t l = 0; l < n_launches; l++)
  {
    hsrans_batch::DirectLaunch L;
    L.member_idx = launch_members[l];
    const uint32_t launch_states = dplans[L.member_idx[0]]->hdr.states;
    uint32_t max_bits = 0;
    for (uint32_t k : L.member_idx)
      max_bits = std::max(max_bits, dplans[k]->hdr.bits);
    // the members' chain starts (in groups) and word offsets, from the device copies of their plans
    std::vector<std::vector<uint64_t>> starts(L.member_idx.size());
    std::vector<BatchDealMember> deal_in(L.member_idx.size());
    std::vector<std::vector<Piece>> pieces(L.member_idx.size());
    for (size_t i = 0; i < L.member_idx.size(); i++)
    {
      const hsrans_dplan *d = dplans[L.member_idx[i]];
      const uint32_t nc = d->hdr.n_chains;
      pieces[i].resize(nc);
      if (hipMemcpy(pieces[i].data(), d->pa.pieces, (size_t)nc * sizeof(Piece), hipMemcpyDeviceToHost) != hipSuccess)
      {
        return HSRANS_E_HIP;
      }
      starts[i].resize(nc + 1);
      uint64_t g = 0;
      for (uint32_t c = 0; c < nc; c++)
      {
        starts[i][c] = g;
        g += pieces[i][c].steps;
      }
      starts[i][nc] = g;
      deal_in[i].chain_start = starts[i].data();
      deal_in[i].n_chains = nc;
      deal_in[i].total_groups = g;
    }
    uint64_t launch_groups = 0;
    for (const BatchDealMember &dm : deal_in)
      launch_groups += dm.total_groups;
    L.shape = batch_direct_shape(ctx->geom, max_bits, launch_groups, launch_states);
    const BatchDeal deal = batch_deal(deal_in, L.shape.grid, L.shape.waves, L.shape.weights, L.shape.kind == kBatchDirect ? 1 : 2);
    // a wave reads its run through one 32-bit window of the stream (run_direct_span: win_open)
    for (const BatchSlot &s : deal.slots)
    {
      const hsrans_dplan *d = dplans[L.member_idx[s.member]];
      if (s.begin >= s.end || s.end > d->hdr.n_chains)
        continue;
      const std::vector<Piece> &pc = pieces[s.member];
      const uint64_t w_end = s.end < d->hdr.n_chains ? pc[s.end].words_off : d->hdr.stream_len;
      if (w_end - pc[s.begin].words_off >= 0xFFFF0000ull)
      {
        return HSRANS_E_FORMAT;
      }
    }
    L.imbalance = deal.imbalance;
    for (size_t i = 0; i < L.member_idx.size(); i++)
      b->order_run[L.member_idx[i]] = deal.order_run[i];
    // blobs to upload: members, then slots
    std::vector<uint8_t> blob(up256(L.member_idx.size() * sizeof(BatchMember)) + up256(deal.slots.size() * sizeof(BatchSlot)));
    BatchMember *bm = (BatchMember *)blob.data();
    for (size_t i = 0; i < L.member_idx.size(); i++)
    {
      const hsrans_dplan *d = dplans[L.member_idx[i]];
      bm[i] = BatchMember{};
      bm[i].pieces = d->pa.pieces;
      bm[i].states = d->pa.states;
      bm[i].table = d->pa.table;
      bm[i].hist_copy = d->pa.hist_copy;
      bm[i].status = d->d_status;
      bm[i].hist_off = d->pa.hist_off;
      bm[i].n_chains = d->hdr.n_chains;
      bm[i].bits = d->hdr.bits;
      bm[i].S = d->hdr.states;
    }
    memcpy(blob.data() + up256(L.member_idx.size() * sizeof(BatchMember)), deal.slots.data(), deal.slots.size() * sizeof(BatchSlot));
    arena += blob.size();
    host_blobs.push_back(std::move(blob));
    b->direct.push_back(std::move(L));
  }
  if (arena != 0)
  {
    if (hipMalloc((void **)&b->d_arena, arena) != hipSuccess)
    {
      (void)hipGetLastError();
      return HSRANS_E_HIP;
    }
    size_t off = 0;
    for (size_t l = 0; l < n_launches; l++)
    {
      if (hipMemcpy(b->d_arena + off, host_blobs[l].data(), host_blobs[l].size(), hipMemcpyHostToDevice) != hipSuccess)
      {
        return HSRANS_E_HIP;
      }
      b->direct[l].d_members = (const BatchMember *)(b->d_arena + off);
      b->direct[l].d_slots = (const BatchSlot *)(b->d_arena + off + up256(b->direct[l].member_idx.size() * sizeof(BatchMember)));
      off += host_blobs[l].size();
    }
  }
  if (getenv("HSRANS_BATCH_STAMPS") != nullptr && !b->direct.empty())
  {
    b->finish_slots = b->direct[0].shape.grid * b->direct[0].shape.waves;
    if (hipMalloc((void **)&b->d_finish, ((size_t)b->finish_slots + 1) * 8) != hipSuccess)
      b->d_finish = nullptr, b->finish_slots = 0;
    else
      b->finish_owned = true, (void)hipMemset(b->d_finish, 0, ((size_t)b->finish_slots + 1) * 8);
  }
  guard.armed = false;
  *out_batch = b;
  return HSRANS_OK;
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

int hsrans_decode_device_batch(hsrans_ctx *ctx, hsrans_batch *b, const void *const *d_streams, const size_t *stream_lengths, void *const *d_outs,
                               const size_t *out_capacities, void *hip_stream)
{
  if (ctx == nullptr || b == nullptr || b->ctx != ctx || d_streams == nullptr || stream_lengths == nullptr || d_outs == nullptr || out_capacities == nullptr)
    return HSRANS_E_ARG;
  const uint32_t count = (uint32_t)b->members.size();
  // everything is checked before anything is launched: a batch decodes as a whole or not at all
  for (uint32_t k = 0; k < count; k++)
  {
    const hsrans_dplan *d = b->members[k];
    if (d_streams[k] == nullptr || d_outs[k] == nullptr || ((uintptr_t)d_streams[k] & 15) != 0 || ((uintptr_t)d_outs[k] & 3) != 0)
      return HSRANS_E_ARG;
    if (stream_lengths[k] < d->hdr.stream_len || out_capacities[k] < d->hdr.decoded_len)
      return HSRANS_E_FORMAT;
  }
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  hipStream_t s = (hipStream_t)hip_stream;
  bool first = true;
  for (const hsrans_batch::DirectLaunch &L : b->direct)
  {
    BatchParams bp{};
    for (size_t i = 0; i < L.member_idx.size(); i++)
    {
      const uint32_t k = L.member_idx[i];
      bp.io[i].stream = (const uint8_t *)d_streams[k];
      bp.io[i].stream_len = stream_lengths[k];
      bp.io[i].out = (uint8_t *)d_outs[k];
      bp.io[i].out_cap = out_capacities[k];
    }
    bp.members = L.d_members;
    bp.slots = L.d_slots;
    bp.finish = first ? b->d_finish : nullptr;
    first = false;
    if (launch_batch_direct(bp, L.shape, s) != hipSuccess)
      return HSRANS_E_HIP;
  }
  for (const hsrans_batch::GroupedLaunch &G : b->grouped)
  {
    BatchGroupParams gp{};
    for (size_t i = 0; i < G.member_idx.size(); i++)
    {
      const uint32_t k = G.member_idx[i];
      gp.io[i].stream = (const uint8_t *)d_streams[k];
      gp.io[i].stream_len = std::min<uint64_t>(stream_lengths[k], b->members[k]->hdr.stream_len);
      gp.io[i].out = (uint8_t *)d_outs[k];
      gp.io[i].out_cap = std::min<uint64_t>(out_capacities[k], b->members[k]->hdr.decoded_len);
    }
    gp.members = G.d_members;
    gp.groups = G.d_groups;
    gp.n_groups = G.n_groups;
    gp.bits = G.bits;
    gp.group_prio = 350; // (run_grouped: the share of its run the younger half of a workgroup decodes at raised priority; dplan_launch's default)
    gp.tickets = G.d_tickets + (size_t)(G.epoch->fetch_add(1, std::memory_order_relaxed) % kCounterSets) * kDynQueueStride;
    memcpy(gp.group_cum, G.shape.group_cum, sizeof(gp.group_cum));
    if (launch_batch_grouped(gp, G.shape, s) != hipSuccess)
      return HSRANS_E_HIP;
  }
  for (uint32_t k : b->solo)
  {
    const int rc = dplan_launch(b->members[k], d_streams[k], stream_lengths[k], d_outs[k], out_capacities[k], s);
    if (rc != HSRANS_OK)
      return rc;
  }
  return HSRANS_OK;
}

int hsrans_dplan_batch_status(hsrans_ctx *ctx, hsrans_batch *b, void *hip_stream, int *member_status)
{
  if (ctx == nullptr || b == nullptr || b->ctx != ctx)
    return HSRANS_E_ARG;
  int worst = HSRANS_OK;
  for (size_t k = 0; k < b->members.size(); k++)
  {
    const int rc = hsrans_dplan_status(ctx, b->members[k], hip_stream);
    if (member_status != nullptr)
      member_status[k] = rc;
    if (rc != HSRANS_OK && worst == HSRANS_OK)
      worst = rc;
  }
  return worst;
}

int hsrans_dplan_batch_info(const hsrans_batch *b, hsrans_batch_info *info)
{
  if (b == nullptr || info == nullptr)
    return HSRANS_E_ARG;
  memset(info, 0, sizeof(*info));
  info->members = (uint32_t)b->members.size();
  info->launches = (uint32_t)(b->direct.size() + b->grouped.size() + b->solo.size());
  for (const hsrans_batch::GroupedLaunch &G : b->grouped)
    info->grouped_members += (uint32_t)G.member_idx.size();
  info->solo_members = (uint32_t)b->solo.size();
  for (const hsrans_batch::DirectLaunch &L : b->direct)
  {
    info->direct_members += (uint32_t)L.member_idx.size();
    info->imbalance = std::max(info->imbalance, L.imbalance);
  }
  if (!b->direct.empty())
  {
    info->grid = b->direct[0].shape.grid;
    info->block = b->direct[0].shape.waves * 64;
    info->lds_bytes = b->direct[0].shape.lds;
    for (int k = 0; k < 8; k++)
      info->class_weights[k] = b->direct[0].shape.weights[k];
  }
  return HSRANS_OK;
}

size_t hsrans_index_boundaries_batch(const hsrans_ctx *ctx, int states, uint32_t bits, const size_t *decoded_sizes, uint32_t count, uint32_t member,
                                     uint64_t *groups_out, size_t capacity)
try
{
  if ((states != 64 && states != 32) || bits < 10 || bits > (states == 64 ? 15u : 12u) || decoded_sizes == nullptr || groups_out == nullptr || count == 0 || count > kBatchMax || member >= count)
    return 0;
  const DeviceGeom dg = ctx ? ctx->geom : default_geom();
  std::vector<uint64_t> totals(count);
  uint64_t all = 0;
  for (uint32_t k = 0; k < count; k++)
  {
    totals[k] = decoded_sizes[k] + 1 >= (size_t)states ? (decoded_sizes[k] - states + 1 + states - 1) / states : 0; // whole groups, as hsrans_index_boundaries
    all += totals[k];
  }
  const BatchShape shape = batch_direct_shape(dg, bits, all, (uint32_t)states);
  // (32 states: two chains per wave slot, one per wave half — run_batch_pair)
  const size_t chains = batch_boundaries(totals.data(), count, member, shape.grid, shape.waves, shape.weights, groups_out, capacity, shape.kind == kBatchDirect ? 1 : 2);
  return chains > 1 ? chains - 1 : 0;
}
catch (...)
{
  return 0;
}

double hsrans_batch_deal(const uint64_t *const *chain_starts, const uint32_t *n_chains, uint32_t members, uint32_t grid, uint32_t waves, const uint32_t *weights,
                         uint32_t *slots_out)
try
{
  if (chain_starts == nullptr || n_chains == nullptr || slots_out == nullptr || members == 0 || grid < 2 || waves == 0 || (uint64_t)grid * waves > (1u << 24))
    return -1.0;
  std::vector<BatchDealMember> in(members);
  for (uint32_t m = 0; m < members; m++)
  {
    if (chain_starts[m] == nullptr || n_chains[m] == 0)
      return -1.0;
    for (uint32_t c = 0; c < n_chains[m]; c++)
      if (chain_starts[m][c] > chain_starts[m][c + 1])
        return -1.0;
    in[m].chain_start = chain_starts[m];
    in[m].n_chains = n_chains[m];
    in[m].total_groups = chain_starts[m][n_chains[m]];
  }
  uint32_t w8[8];
  if (weights == nullptr)
  {
    const BatchShape shape = batch_direct_shape(default_geom(), 11);
    memcpy(w8, shape.weights, sizeof(w8));
  }
  else
    memcpy(w8, weights, sizeof(w8));
  const BatchDeal deal = batch_deal(in, grid, waves, w8);
  memcpy(slots_out, deal.slots.data(), deal.slots.size() * sizeof(BatchSlot));
  return deal.imbalance;
}
catch (...)
{
  return -1.0;
}

size_t hsrans_dplan_batch_read_finish(hsrans_batch *b, uint64_t *out, size_t capacity_u64)
{
  if (b == nullptr || b->d_finish == nullptr || out == nullptr)
    return 0;
  const size_t n = std::min<size_t>(capacity_u64, (size_t)b->finish_slots + 1);
  return hipMemcpy(out, b->d_finish, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? n : 0;
}

} // extern "C"
