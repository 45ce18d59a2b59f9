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

namespace hsrans
{

// wave `pos` of a workgroup's slot order with class runs of `run` waves: the classes (wave / 4) take turns, `run` waves each.
// run == 4 is the natural order 0..15; run == 1 is 0, 4, 8, 12, 1, 5, ...  (16-wave workgroups)
static uint32_t order_wave(uint32_t pos, uint32_t run)
{
  const uint32_t per_turn = 4 * run; // waves placed per turn over the four classes
  const uint32_t turn = pos / per_turn, in_turn = pos % per_turn;
  const uint32_t cls = in_turn / run, u = in_turn % run;
  return cls * 4 + turn * run + u;
}

// Workgroup PAIRS per member (one workgroup from each half of the grid: a CU's older and its younger workgroup, so that every
// member gets waves of all 8 age classes in the launch's own proportion).  Min-max: one pair each, then one more to whoever has
// the most groups per pair, until the pairs are used up; a member cannot use more waves than it has chains (max_chains, or null).
std::vector<uint32_t> batch_apportion(const uint64_t *total_groups, const uint64_t *max_chains, uint32_t M, uint32_t pairs, uint32_t waves)
{
  std::vector<uint32_t> n(M, 0);
  uint32_t used = 0;
  for (uint32_t m = 0; m < M && used < pairs; m++)
    if (total_groups[m] != 0 || (max_chains != nullptr && max_chains[m] != 0))
      n[m] = 1, used++;
  for (; used < pairs; used++)
  {
    uint32_t best = M;
    double best_load = 0;
    for (uint32_t m = 0; m < M; m++)
    {
      if (n[m] == 0 || (max_chains != nullptr && (uint64_t)n[m] * 2 * waves >= max_chains[m]))
        continue;
      const double load = (double)total_groups[m] / n[m];
      if (best == M || load > best_load)
        best = m, best_load = load;
    }
    if (best == M)
      break;
    n[best]++;
  }
  return n;
}

// Checkpoint positions (ascending group indices, multiples of 4) that give member `member` of a batch of streams with
// total_groups[0 .. M) whole groups exactly one chain per wave slot the batch launch will deal it, each sized by the slot's class:
// what direct_boundaries does for a stream that has the device to itself.  Returns the number of chains (boundaries + 1), 0 = capacity.
size_t batch_boundaries(const uint64_t *total_groups, uint32_t M, uint32_t member, uint32_t grid, uint32_t waves, const uint32_t weights[8], uint64_t *out, size_t cap)
{
  if (member >= M || grid < 2 || waves == 0)
    return 0;
  const uint32_t first_half = (grid + 1) / 2, pairs = grid - first_half;
  const std::vector<uint32_t> n = batch_apportion(total_groups, nullptr, M, pairs, waves);
  const uint64_t nslots = (uint64_t)n[member] * 2 * waves;
  const uint64_t all_units = total_groups[member] / 4; // boundaries in units of 4 groups (the decode loop stores 4 groups at a time)
  uint64_t chains = nslots;
  if (chains > all_units / 8) // (as direct_boundaries: a chain is worth its index entry and its prologue from about 32 groups on)
    chains = all_units / 8 ? all_units / 8 : 1;
  if (chains <= 1)
    return 1;
  if (chains - 1 > cap)
    return 0;
  const uint32_t per_class = waves >= 4 ? waves / 4 : 1;
  auto weight_of = [&](uint64_t i) { // slot i of the member in natural order: its first-half workgroups' waves, then its second-half ones
    const uint64_t wg_local = i / waves;
    const uint32_t wave = (uint32_t)(i % waves);
    return (uint64_t)weights[(wg_local >= n[member] ? 4 : 0) + std::min(3u, wave / per_class)];
  };
  uint64_t all = 0;
  for (uint64_t k = 0; k < chains; k++)
    all += weight_of(k);
  uint64_t cum = 0, prev = 0;
  size_t count = 0;
  for (uint64_t k = 0; k + 1 < chains; k++)
  {
    cum += weight_of(k);
    uint64_t b = (uint64_t)((unsigned __int128)all_units * cum / all);
    if (b <= prev)
      b = prev + 1;
    if (b >= all_units)
      break;
    out[count++] = b * 4;
    prev = b;
  }
  return count + 1;
}

// Deals wave slots to the members' chains.  See hsrans_batch.h.
BatchDeal batch_deal(const std::vector<BatchDealMember> &members, uint32_t grid, uint32_t waves, const uint32_t weights[8])
{
  BatchDeal out;
  const uint32_t M = (uint32_t)members.size();
  const uint32_t W = grid * waves;
  // (a slot no member's chains reach names member 0 and no chains: its wave only takes part in its workgroup's table copy)
  const uint32_t idle = members.empty() ? 0 : members[0].n_chains;
  out.slots.assign(W, BatchSlot{0, idle, idle, 0});
  out.wg_first.assign(M, 0);
  out.wg_count.assign(M, 0);
  out.order_run.assign(M, 4);
  if (M == 0 || grid < 2 || waves == 0)
    return out;
  // 1. workgroups
  const uint32_t first_half = (grid + 1) / 2, pairs = grid - first_half; // (grid odd: the first half's last workgroup stays idle)
  std::vector<uint64_t> totals(M), caps(M);
  for (uint32_t m = 0; m < M; m++)
    totals[m] = members[m].total_groups, caps[m] = members[m].n_chains;
  const std::vector<uint32_t> n = batch_apportion(totals.data(), caps.data(), M, pairs, waves);
  // 2. per member: its slots in order, its chains dealt to them by cumulative weight, boundaries at the nearest chain start.  The
  //    order of the slots inside a workgroup is tried three ways (class runs of 4, 2, 1 waves): an index made for a launch of its own
  //    (hsrans_index_boundaries: chains sized by class, four of a class in a row) is matched exactly by one of them when the member
  //    gets 1/1, 1/2 or 1/4 of the device; a uniform index does not care.  The order with the least (longest run / weight) wins.
  uint32_t wg0 = 0;
  double worst = 0, mean_num = 0, mean_den = 0;
  const bool can_reorder = waves == 16;
  for (uint32_t m = 0; m < M; m++)
  {
    const BatchDealMember &mem = members[m];
    out.wg_first[m] = wg0;
    out.wg_count[m] = n[m];
    if (n[m] == 0)
      continue;
    const uint32_t nslots = n[m] * 2 * waves;
    const uint32_t nc = mem.n_chains;
    std::vector<uint32_t> best_bounds;
    double best_cost = -1;
    uint32_t best_run = 4;
    std::vector<uint32_t> slot_of(nslots), bounds(nslots + 1);
    std::vector<uint64_t> cumw(nslots + 1);
    for (uint32_t run : {4u, 2u, 1u})
    {
      if (run != 4 && !can_reorder)
        break;
      // slot list: first-half workgroups wg0 .. wg0 + n, then second-half workgroups first_half + wg0 ..
      for (uint32_t i = 0; i < nslots; i++)
      {
        const uint32_t wg_local = i / waves, pos = i % waves;
        const uint32_t wg = wg_local < n[m] ? wg0 + wg_local : first_half + wg0 + (wg_local - n[m]);
        const uint32_t wave = can_reorder ? order_wave(pos, run) : pos;
        slot_of[i] = wg * waves + wave;
      }
      cumw[0] = 0;
      for (uint32_t i = 0; i < nslots; i++)
      {
        const uint32_t wg = slot_of[i] / waves, wave = slot_of[i] % waves;
        const uint32_t per_class = waves >= 4 ? waves / 4 : 1;
        const uint32_t cls = (wg >= first_half ? 4 : 0) + std::min(3u, wave / per_class);
        cumw[i + 1] = cumw[i] + weights[cls];
      }
      const uint64_t G = mem.chain_start[nc];
      bounds[0] = 0;
      double cost = 0;
      for (uint32_t i = 0; i < nslots; i++)
      {
        uint32_t b = nc;
        if (i + 1 < nslots)
        {
          const uint64_t target = (uint64_t)((unsigned __int128)G * cumw[i + 1] / cumw[nslots]);
          const uint64_t *lo = std::lower_bound(mem.chain_start, mem.chain_start + nc + 1, target);
          b = (uint32_t)(lo - mem.chain_start);
          if (b > 0 && b <= nc && target - mem.chain_start[b - 1] < mem.chain_start[std::min(b, nc)] - target)
            b--;
          b = std::min(std::max(b, bounds[i]), nc);
        }
        bounds[i + 1] = b;
        const uint64_t len = mem.chain_start[b] - mem.chain_start[bounds[i]];
        const uint32_t wt = (uint32_t)(cumw[i + 1] - cumw[i]);
        cost = std::max(cost, (double)len / (wt ? wt : 1));
      }
      if (best_cost < 0 || cost < best_cost * 0.999)
      {
        best_cost = cost;
        best_run = run;
        best_bounds = bounds;
      }
    }
    // write the winner's slots
    for (uint32_t i = 0; i < nslots; i++)
    {
      const uint32_t wg_local = i / waves, pos = i % waves;
      const uint32_t wg = wg_local < n[m] ? wg0 + wg_local : first_half + wg0 + (wg_local - n[m]);
      const uint32_t wave = can_reorder ? order_wave(pos, best_run) : pos;
      BatchSlot &s = out.slots[wg * waves + wave];
      s.member = m;
      s.begin = best_bounds[i] < best_bounds[i + 1] ? best_bounds[i] : nc;
      s.end = best_bounds[i] < best_bounds[i + 1] ? best_bounds[i + 1] : nc;
      s.flags = wg_local == 0 ? kBatchSlotCheckHist : 0;
    }
    out.order_run[m] = best_run;
    // (cost is groups per unit of weight: the launch is as long as its most loaded slot)
    uint64_t wsum = 0;
    for (uint32_t k = 0; k < 8; k++)
      wsum += weights[k];
    worst = std::max(worst, best_cost);
    mean_num += (double)mem.chain_start[nc];
    mean_den += (double)n[m] * (waves / 4) * wsum;
    wg0 += n[m];
  }
  out.imbalance = mean_num > 0 && mean_den > 0 ? worst / (mean_num / mean_den) : 1.0;
  return out;
}

} // namespace hsrans

// ---------------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------------
namespace
{
// a plan the one-chain-per-wave batch kernel can take: raw, mergeable, 64 states, the 8-byte host-built table (bits <= 12), one chain per wave
bool direct_eligible(const hsrans_dplan *d)
{
  return d->pa.pieces != nullptr && d->pa.table != nullptr && d->pa.table_mode == 3 && d->pa.dual == 0 && d->hdr.states == 64 && d->hdr.bits <= 12 &&
         (d->hdr.flags & kPlanMergeable) != 0 && d->hdr.container == HSRANS_RAW && d->hdr.n_chains >= 1;
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
      hsrans_batch::GroupedLaunch G;
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
          hsrans_dplan_batch_destroy(b);
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
        hsrans_dplan_batch_destroy(b);
        return HSRANS_E_HIP;
      }
      G.d_members = (const GroupMember *)dev;
      G.d_groups = (const Group *)(dev + up256g(gm.size() * sizeof(GroupMember)));
      G.d_tickets = (unsigned long long *)(dev + up256g(gm.size() * sizeof(GroupMember)) + up256g(all.size() * sizeof(Group)));
      G.epoch = new (std::nothrow) std::atomic<uint32_t>(0);
      const bool ok = G.epoch != nullptr && hipMemset(dev, 0, bytes) == hipSuccess && hipMemcpy(dev, gm.data(), gm.size() * sizeof(GroupMember), hipMemcpyHostToDevice) == hipSuccess &&
                      hipMemcpy((void *)G.d_groups, all.data(), all.size() * sizeof(Group), hipMemcpyHostToDevice) == hipSuccess;
      b->grouped.push_back(std::move(G)); // (owned by the batch from here: destroyed with it)
      if (!ok)
      {
        hsrans_dplan_batch_destroy(b);
        return HSRANS_E_HIP;
      }
    }
  }
  std::sort(b->solo.begin(), b->solo.end());
  if (eligible.size() == 1) // a launch of its own is the same thing, with the plan's own dealing
  {
    b->solo.push_back(eligible[0]);
    eligible.clear();
    std::sort(b->solo.begin(), b->solo.end());
  }
  const size_t n_launches = (eligible.size() + kBatchMax - 1) / kBatchMax;
  // one allocation: per launch its member records and its slot table
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t arena = 0;
  std::vector<std::vector<uint8_t>> host_blobs;
  for (size_t l = 0; l < n_launches; l++)
  {
    hsrans_batch::DirectLaunch L;
    // (launches of nearly equal member counts rather than 32 + the rest)
    const size_t lo = eligible.size() * l / n_launches, hi = eligible.size() * (l + 1) / n_launches;
    L.member_idx.assign(eligible.begin() + lo, eligible.begin() + hi);
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
        hsrans_dplan_batch_destroy(b);
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
    L.shape = batch_direct_shape(ctx->geom, max_bits, launch_groups);
    const BatchDeal deal = batch_deal(deal_in, L.shape.grid, L.shape.waves, L.shape.weights);
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
        hsrans_dplan_batch_destroy(b);
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
      hsrans_dplan_batch_destroy(b);
      return HSRANS_E_HIP;
    }
    size_t off = 0;
    for (size_t l = 0; l < n_launches; l++)
    {
      if (hipMemcpy(b->d_arena + off, host_blobs[l].data(), host_blobs[l].size(), hipMemcpyHostToDevice) != hipSuccess)
      {
        hsrans_dplan_batch_destroy(b);
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
  if (states != 64 || bits < 10 || bits > 12 || decoded_sizes == nullptr || groups_out == nullptr || count == 0 || count > kBatchMax || member >= count)
    return 0;
  const DeviceGeom dg = ctx ? ctx->geom : default_geom();
  std::vector<uint64_t> totals(count);
  uint64_t all = 0;
  for (uint32_t k = 0; k < count; k++)
  {
    totals[k] = decoded_sizes[k] + 1 >= 64 ? (decoded_sizes[k] - 64 + 1 + 63) / 64 : 0; // whole groups, as hsrans_index_boundaries
    all += totals[k];
  }
  const BatchShape shape = batch_direct_shape(dg, bits, all);
  const size_t chains = batch_boundaries(totals.data(), count, member, shape.grid, shape.waves, shape.weights, groups_out, capacity);
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
