// hsrans_batch.h — the host-side dealing of a batch launch's wave slots (hsrans_batch.cpp); pure arithmetic, no device needed.
#ifndef HSRANS_BATCH_H
#define HSRANS_BATCH_H

#include <stdint.h>

#include <vector>

#include "hsrans_kernels.h"

namespace hsrans
{

struct BatchDealMember
{
  const uint64_t *chain_start; // [n_chains + 1]: first group of chain c, counted from the member's first chain; [n_chains] = all its groups
  uint32_t n_chains;
  uint64_t total_groups;
};

struct BatchDeal
{
  std::vector<BatchSlot> slots;       // [grid * waves], slot of wave `wave` of workgroup `wg` at wg * waves + wave
  std::vector<uint32_t> wg_first;     // per member: its first workgroup in the grid's first half (the same offset in the second half)
  std::vector<uint32_t> wg_count;     // ... and how many it has in EACH half
  std::vector<uint32_t> order_run;    // per member: the class-run length of the slot order its chains were dealt with (4 = natural)
  double imbalance = 1.0;             // most loaded slot's (groups / class weight) over the launch's mean: 1.0 = every wave ends together
};

// Deals the launch's workgroups to the members (whole workgroups: one decode table each; the same number from each half of the grid,
// so that every member sees all 8 age classes) in proportion to their groups, then every member's chains to its wave slots as runs of
// consecutive chains whose lengths follow the slots' class weights.
std::vector<uint32_t> batch_apportion(const uint64_t *total_groups, const uint64_t *max_chains, uint32_t M, uint32_t pairs, uint32_t waves, uint32_t chains_per_slot = 1);
size_t batch_boundaries(const uint64_t *total_groups, uint32_t M, uint32_t member, uint32_t grid, uint32_t waves, const uint32_t weights[8], uint64_t *out, size_t cap,
                        uint32_t chains_per_slot = 1);
// (chains_per_slot: 2 for 32-state members — a slot's run is decoded as two halves side by side, so a member can use twice as many chains as slots)
BatchDeal batch_deal(const std::vector<BatchDealMember> &members, uint32_t grid, uint32_t waves, const uint32_t weights[8], uint32_t chains_per_slot = 1);

} // namespace hsrans

#endif // HSRANS_BATCH_H
