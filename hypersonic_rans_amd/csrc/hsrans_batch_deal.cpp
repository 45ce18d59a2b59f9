// hsrans_batch_deal.cpp — the host-side arithmetic of a batch launch (no device, no HIP runtime): which workgroups each member gets,
// which run of its chains each wave slot decodes (batch_deal), and the checkpoint positions that make that dealing exact
// (batch_boundaries).  Split from hsrans_batch.cpp so that tools/fuzz builds it under AddressSanitizer + UBSan.
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "hsrans_batch.h"

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
std::vector<uint32_t> batch_apportion(const uint64_t *total_groups, const uint64_t *max_chains, uint32_t M, uint32_t pairs, uint32_t waves, uint32_t chains_per_slot)
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
      if (n[m] == 0 || (max_chains != nullptr && (uint64_t)n[m] * 2 * waves * chains_per_slot >= max_chains[m]))
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
size_t batch_boundaries(const uint64_t *total_groups, uint32_t M, uint32_t member, uint32_t grid, uint32_t waves, const uint32_t weights[8], uint64_t *out, size_t cap,
                        uint32_t chains_per_slot)
{
  if (chains_per_slot == 0)
    return 0;
  if (member >= M || grid < 2 || waves == 0)
    return 0;
  const uint32_t first_half = (grid + 1) / 2, pairs = grid - first_half;
  const std::vector<uint32_t> n = batch_apportion(total_groups, nullptr, M, pairs, waves, chains_per_slot);
  const uint64_t nslots = (uint64_t)n[member] * 2 * waves * chains_per_slot; // (in chains: a slot's chains share its weight)
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
    const uint64_t slot = i / chains_per_slot, wg_local = slot / waves;
    const uint32_t wave = (uint32_t)(slot % waves);
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
BatchDeal batch_deal(const std::vector<BatchDealMember> &members, uint32_t grid, uint32_t waves, const uint32_t weights[8], uint32_t chains_per_slot)
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
  const std::vector<uint32_t> n = batch_apportion(totals.data(), caps.data(), M, pairs, waves, chains_per_slot ? chains_per_slot : 1);
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
