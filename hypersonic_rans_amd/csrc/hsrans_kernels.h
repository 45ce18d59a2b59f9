// Launch interface between the C ABI (hsrans_capi.cpp) and the gfx950 kernels (hsrans_kernels.hip).
#ifndef HSRANS_KERNELS_H
#define HSRANS_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hsrans_plan.h"

namespace hsrans
{

// Persistent launch of a kPlanMergeable plan: everything a wave needs without a dependent read of the plan header.
// Chain c (local index) covers groups [c*interval, min((c+1)*interval, steps_total)) of the run that starts at out_base.
struct PersistentArgs
{
  const Piece *pieces;   // null = not a persistent launch
  const uint32_t *states;
  uint32_t n_chains, interval, S, bits; // interval == 0: chains of any length, one per wave (run_direct); else uniform chains

  uint64_t out_base, steps_total, hist_off;
  uint32_t tail;            // symbols of the final partial group (after the last chain)
  uint32_t static_per_wave; // chains every wave decodes as one merged run before it starts pulling single chains (mean)
  // 64-state launches: the static run of a wave depends on its slot, class k = (workgroup in the grid's second half) * 4
  // + (wave in workgroup / 4): the SIMD arbiter serves its oldest wave first, so equal shares finish far apart
  // (run_persistent).  run_len[k] chains; wave j of class k in workgroup b' of its half starts at
  // half_base[h] + b' * wg_chains[h] + class_off[k] + (j & 3) * run_len[k].  static_total = all static chains.
  uint32_t run_len[8], class_off[8], wg_chains[2], half_base[2], static_total;
  const uint2 *table;       // host-built decode table (kPlanHasHist plans) or null: build it in the kernel
  uint32_t table_mode;      // 3: one uint2 per slot; 4: coarse + fine tables (bits 13..15); 5: as 3, left in global memory; see hsrans_kernels.hip
  uint32_t dual;            // two chains per wave: k_decode_dual
  const uint16_t *hist_copy; // the 256 counts that table was built from (device copy inside the plan)
  unsigned long long *counters; // [kDynQueues * kDynQueueStride] monotonic queue heads of THIS launch's counter set (never reset, see run_persistent)
  // one-chain-per-wave launches (interval == 0, 64 states): the plan's chains are dealt to the launch's waves as runs of run_chains
  // consecutive chains, each decoded as one chain (run_direct; 1 for the index hsrans_index_boundaries makes for this device)
  uint32_t run_chains;
};
constexpr uint32_t kDynQueues = 64;
constexpr uint32_t kDynQueueStride = 32; // in uint64: one head per 256 B, so that heads never share a line / atomic unit
// A device plan owns kCounterSets sets of queue heads and every launch takes the next one (round robin): launches of one
// plan may overlap (several streams, double-buffered outputs) as long as fewer than kCounterSets are in flight at once.
constexpr uint32_t kCounterSets = 32;

// Grouped launch (block_/mt_ plans with checkpoints): chains [begin, begin+count) share one histogram = one LDS table
// per workgroup; its waves split the chains evenly (merged into one run per wave when the chains are back to back).
struct Group
{
  uint32_t begin, count;
  uint32_t flags; // 1: chains are single back-to-back rANS pieces (mergeable); 2: fill chains only (no table)
  uint32_t piece0; // mergeable groups: index of chain `begin`'s piece (chain begin + i is piece piece0 + i, its states are states[begin + i])
  uint64_t hist_off;
  uint64_t words_end; // first stream byte after the group's words (next group's first header) or the stream length
};
constexpr uint32_t kGroupMergeable = 1;
constexpr uint32_t kGroupFill = 2;

// A rank's sub-runs as ONE launch (hsrans_decode_sharded, round 6).  The reference hands every block of a stream to its pool in one
// pass and joins once (mt_rANS32x64_16w_decode.cpp:182-224, :262); a rank of a sharded decode used to queue one launch per sub-run
// so that sub-run k's ranges could go onto the links while sub-run k + 1 decodes — at 8 ranks x 4 sub-runs each launch was 32 MiB
// on a device that needs 100 MB to fill, and every one paid prologue, tail and kernel boundary.  Now the sub-runs ("parts":
// contiguous chain ranges of the rank's plan) are decoded by one launch in list order, and the launch itself tells the exchange's
// stream when a part is complete: a unit of work (a group of the grouped launch, a workgroup's share of the spread launch) that
// is done makes its stores visible (release fence, device scope), then counts itself into every part it overlaps
// (PartArgs::count, monotonic, never reset); the unit whose increment reaches `target` publishes `seq` to the part's completion
// word, which hipStreamWaitValue32(..., Gte) on the exchange's stream is waiting for.
constexpr uint32_t kMaxLaunchParts = 16;
struct PartArgs
{
  uint32_t n;                           // 0: not such a launch
  uint32_t seq;                         // what a completion word receives (one more with every launch of the sharded decode)
  uint32_t *count;                      // [n] units seen so far, over all launches
  uint32_t *done[kMaxLaunchParts];      // completion words (signal memory: the command processor polls them)
  uint32_t target[kMaxLaunchParts];     // count[k] once this launch's last unit of part k is in
  uint32_t chain_end[kMaxLaunchParts];  // part k = chains [chain_end[k - 1], chain_end[k]) of the launch's plan (the spread launch's units are chain ranges)
};
// grouped launches: Group::flags carries the first / last part a group overlaps in bits 16..23 / 24..31 (a group is a block or a
// part of one; sub-runs are cut at chain boundaries, so a group can straddle two of them)
constexpr uint32_t kGroupPartShift = 16;

// k_decode_dealt (kernels_dealt.h): the host-dealt one-round launch of block_/mt_ plans with checkpoints
constexpr uint32_t kDealtGridMax = 512;
struct DealtTable
{
  uint32_t begin[kDealtGridMax + 1]; // chains; begin[grid] = n_chains
  uint16_t split[kDealtGridMax];     // chains of the share that belong to its first block (>= the share's length: one block only)
};
struct DealtParams
{
  const uint8_t *stream;
  uint64_t stream_len, stream_lo;
  uint8_t *out;
  uint64_t out_cap;
  const Piece *pieces;    // chain c = piece c with start states c (single-piece chains)
  const uint32_t *states;
  uint32_t *status;
  uint64_t *stamps;       // diagnostics
  uint32_t n_chains, bits;
  uint16_t cum[2][17];    // a wave's part of its workgroup's share: cumulative class weights by grid half (as KParams::group_cum)
  uint32_t gap_chains;    // what the second prologue of a wave that straddles its share's block boundary costs, in chains of this plan (run_dealt)
  PartArgs parts;
};

// k_decode_single: a plan of ONE chain of ONE rANS piece (a raw stream without an index), filled by the host from the plan
struct SingleArgs
{
  uint32_t valid;        // 0: not such a plan (or bits == 15: its 8-byte table does not fit LDS)
  uint32_t steps, tail, S, bits, ring_entries;
  uint64_t hist_off, words_off, out_off;
};

struct KParams
{
  const uint8_t *stream; // device, 16-byte aligned: where stream byte 0 is (or would be: see stream_lo)
  uint64_t stream_len;
  uint64_t stream_lo;    // first stream byte that really exists at stream + offset (window launches; else 0)
  uint8_t *out; // device, 4-byte aligned
  uint64_t out_cap;
  const uint8_t *plan; // device copy of the plan blob
  uint32_t *status;    // device status word (kStatus* bits)
  // index-build pass only (ckpt_interval != 0): checkpoint g / ckpt_interval receives the S states and the absolute
  // byte position of the read cursor at group boundary g
  uint32_t *ckpt_states;
  uint64_t *ckpt_words;
  uint32_t ckpt_interval;
  // the same pass with explicit checkpoint positions: ascending absolute group indices; boundary k -> slot k
  const uint64_t *ckpt_groups;
  uint32_t n_ckpt_groups;
  // index-build pass over a block_ stream (walk plan): block b's header position, output offset and header word go to
  // walk_blocks[3b .. 3b+2], the coder states on entry to walk_states[b * S ..]; walk_count[0] = blocks seen
  uint64_t *walk_blocks;
  uint32_t *walk_states;
  uint32_t *walk_count;
  uint32_t walk_max_blocks;
  // diagnostics only (HSRANS_DEBUG_STAMPS=1): per wave {entry, table built, stream ready, done} s_memtime stamps; null otherwise
  uint64_t *stamps;
  PersistentArgs pa;
  const Group *groups; // null = not a grouped launch
  uint32_t n_groups;
  // grouped launches with more groups than workgroups: the groups behind the first gridDim.x are handed out by a ticket counter
  // (monotonic, this launch's own: never reset — every launch draws exactly n_groups tickets — see run_grouped); null = static
  unsigned long long *group_tickets;
  // calibration launches only (hsrans_ctx_calibrate): wave w of a one-chain-per-wave launch leaves its finish time (s_memrealtime,
  // 100 MHz) in finish[w]; finish[gridDim.x * waves] = the first wave's entry time
  uint64_t *finish;
  uint32_t spread;        // grouped plans of single-piece chains: the fewest chains of a coded block that is not the last (k_decode_spread takes the launch when its longest share is shorter); 0 = never
  uint32_t groups_lean;   // 64-state plan, every group a mergeable run or fills only: the lean instantiation of k_decode_grouped
  uint32_t group_prio;    // grouped launches: per mille of its run the younger half of a workgroup's waves decodes at raised priority (s_setprio)
  // ... per wave class (class = (workgroup in the grid's second half) * 4 + wave / (waves / 4), as everywhere): [0..7] where a group's chains are
  // split evenly over the waves, [8..9] (by grid half) where the split already follows the class weights.  All zero: group_prio's rule.
  uint16_t group_prio_class[10];
  // grouped launches: wave k of a workgroup in grid half h takes chains [count * cum[h][k] / cum[h][waves], count * cum[h][k+1] / cum[h][waves])
  // of its group (the same age-class weights as PersistentArgs::run_len)
  uint16_t group_cum[2][17];
  SingleArgs single;
  const uint32_t *single_states; // the chain's S start states (device)
  // private-table launches of 32-state plans: every wave takes TWO chains (2w, 2w+1), one per wave half, each with its own
  // table (run_private_pair); the LDS layout then holds two tables per wave
  uint32_t private_pair;
  PartArgs parts; // a sharded decode's sub-runs in one launch (k_decode_grouped<.., true, true>, k_decode_spread<.., true>); n == 0 otherwise
};

// ---- K independent streams in one launch (kernels_batch.h, hsrans_batch.cpp) -----------------------------------------------
// what is fixed when the batch is made: a member's plan, table and status word (device memory, read through the scalar cache)
struct BatchMember
{
  const Piece *pieces;
  const uint32_t *states;
  const uint2 *table;        // host-built decode table (MODE 3)
  const uint16_t *hist_copy; // the 256 counts it was built from
  uint32_t *status;          // the member's own status word (its device plan's)
  uint64_t hist_off;
  uint32_t n_chains, bits, S, reserved;
};
static_assert(sizeof(BatchMember) == 64, "BatchMember layout");
// one per wave of the launch: chains [begin, end) of member `member` are the wave's run (begin == end == n_chains: none)
struct BatchSlot
{
  uint32_t member, begin, end, flags;
};
constexpr uint32_t kBatchSlotCheckHist = 1; // the member's first workgroup: compares the table's histogram with the stream's
// what changes from launch to launch: where the member's stream and output are.  Passed as kernel arguments (no copy in front of
// the launch, graph-capturable), which bounds a launch at kBatchMax members; larger batches take several launches.
struct BatchIO
{
  const uint8_t *stream;
  uint64_t stream_len;
  uint8_t *out;
  uint64_t out_cap;
};
constexpr uint32_t kBatchMax = 32;
struct BatchParams
{
  BatchIO io[kBatchMax];
  const BatchMember *members;
  const BatchSlot *slots;
  uint64_t *finish; // diagnostics (batch stamps): wave w's finish time, [n_slots] = the first wave's entry; null otherwise
  uint64_t *stamps;
};
// the grouped form (block_/mt_ members with checkpoints): Group::flags carries the member from bit kGroupMemberShift up
constexpr uint32_t kGroupMemberShift = 16;
struct GroupMember
{
  const uint32_t *chain_first;
  const Piece *pieces;
  const uint32_t *states;
  uint32_t *status;
};
struct BatchGroupParams
{
  BatchIO io[kBatchMax];
  const GroupMember *members;
  const Group *groups; // all members' groups, flags |= member << kGroupMemberShift
  unsigned long long *tickets;
  uint32_t n_groups, bits, group_prio, reserved;
  uint16_t group_cum[2][17];
};
struct BatchGroupShape
{
  uint32_t grid, waves, lds;
  uint16_t group_cum[2][17];
};
BatchGroupShape batch_grouped_shape(const struct DeviceGeom &dg, uint32_t bits, uint32_t n_groups, uint64_t n_chains);
hipError_t launch_batch_grouped(const BatchGroupParams &bp, const BatchGroupShape &shape, hipStream_t stream);

constexpr uint32_t kBatchDirect = 0, kBatchPair = 1, kBatchDualPack = 2, kBatchDualRank = 3; // which kernel a shared launch runs
struct BatchShape
{
  uint32_t grid, waves, lds, states, kind;
  uint32_t weights[8]; // per-mille run lengths of the 8 wave classes (class = (workgroup in the grid's second half) * 4 + wave / 4)
};

struct LaunchInfo
{
  uint32_t grid, block, lds_bytes, waves_per_block, chains, shared_table, walk, two_level, table_mode, chains_per_wave;
  uint32_t class_weights[8]; // the per-mille run lengths of the 8 wave classes this launch was shaped with (LaunchShape::weights)
  uint32_t dynamic_groups;   // grouped launches: groups handed out by the ticket counter (1) or in static order (0)
  uint32_t spread;           // grouped plan launched by k_decode_spread (chains dealt out evenly, two tables per workgroup); 2: by k_decode_dealt (host-dealt shares)
};

// what the launcher needs to know about the device a context lives on
struct DeviceGeom
{
  uint32_t num_cus, max_lds;
  // per-mille chain lengths of the 8 wave classes of the 64-state one-chain-per-wave launch, fitted to THIS device by
  // hsrans_ctx_calibrate (0 = not calibrated: the constants fitted on the development box, g_direct_weights)
  uint32_t have_direct_weights;
  uint32_t direct_weights[8];
  // The same fit at several RUN LENGTHS (mean groups per wave of the launch; ascending): a wave's time is its prologue plus its groups
  // at its class's rate, so the lengths that make all classes finish together depend on how long the runs are — an old wave's
  // head start counts for less in a long run.  hsrans_ctx_calibrate fits the 48 MiB launch (96 groups per wave),
  // hsrans_ctx_calibrate_runs longer ones; direct_weights_for interpolates between the fitted lengths (in log run length).
  uint32_t n_weight_sets;
  uint32_t set_run[4];
  uint32_t set_weights[4][8];
};
// the per-mille chain lengths of the 8 wave classes for a 64-state one-chain-per-wave launch whose runs average run_groups
void direct_weights_for(const DeviceGeom &dg, uint64_t run_groups, uint32_t out[8]);

// the launch all members of a batch of 64-state plans with 8-byte tables (bits <= 12) share; max_bits = the widest member
// (total_groups: all members' groups together — the class weights follow the launch's mean run length; 0 = the default set)
BatchShape batch_direct_shape(const DeviceGeom &dg, uint32_t max_bits, uint64_t total_groups = 0, uint32_t states = 64);
hipError_t launch_batch_direct(const BatchParams &bp, const BatchShape &shape, hipStream_t stream);

// a launch's shape as it follows from plan header + device (launch_shape)
struct LaunchShape
{
  int mode;         // decode-table layout (kMode* in hsrans_kernels.hip)
  bool shared, walk, dual;
  uint32_t waves, lds, grid, resident, private_pair;
  uint32_t weights[8]; // per-mille run length of the 8 wave classes
};

// K2: device-side walk of an mt_ stream's header chain (mt_rANS32x64_16w_decode.cpp:41-96), the device twin of the host
// planner.  Pass 1 (k_mt_chase, one wavefront) follows the chain with one 16-byte read per block and lists the blocks;
// pass 2 (k_mt_fill, one wavefront per block) writes the plan blob.
struct WalkResult
{
  uint32_t n_chains; // chains (= pieces = blocks) found
  uint32_t error;    // 0 ok; 7: the block list was too small; else the stream is malformed (the reference's "return 0" cases)
  uint64_t decoded_len;
};
hipError_t launch_mt_chase(const uint8_t *d_stream, uint64_t stream_len, uint64_t out_cap, uint32_t S, uint64_t *d_blocks, uint32_t max_blocks, WalkResult *d_result,
                           hipStream_t stream);
hipError_t launch_mt_fill(const uint8_t *d_stream, uint64_t stream_len, uint32_t S, uint32_t bits, const uint64_t *d_blocks, uint8_t *d_plan, uint32_t n_chains,
                          uint64_t out_len, WalkResult *d_result, hipStream_t stream);

// the indexed plan assembled on the device from a base plan + recorded checkpoints (hsrans_decode_device_indexing; kernels_walk.h)
// Few, large blocks (fewer groups than workgroup slots): a block's chains are cut into parts — groups of their own: same histogram,
// a sub-range of the chains — of at least kGroupPartChains chains until there are kGroupPartsPerCU parts per CU.  Round 4, in-process
// A/B on 100 MB (rotated, us; before: parts of >= 128 chains until two per CU): 256 KiB blocks 57.2 -> 51.3, 512 KiB 57.3 -> 53.2,
// 1 MiB 56.3 -> 52.1, 4 MiB 62.1 -> 51.1, 256 KiB at 13 bits 69.2 -> 58.1, at 15 bits 71.6 -> 62.2; parts of 48 or 32 chains lose again
// (profiles/r04_group_split_ab.txt).  Host (dplan_fill) and device (k_plan_blocks, k_index_fill) use the same rule.
constexpr uint32_t kGroupPartChains = 64, kGroupPartsPerCU = 3;
__host__ __device__ inline uint32_t group_parts_of(uint32_t chains, uint32_t k_max)
{
  const uint32_t by_size = chains / kGroupPartChains;
  const uint32_t k = by_size < k_max ? by_size : k_max;
  return k < 1 ? 1 : k;
}

// k_decode_spread (kernels_spread.h): G = two 16-wave workgroups per CU; workgroup b's share of the plan's N chains starts at
// spread_share_begin(b): the first half of the grid weighs w1 per workgroup, the second half w2 (the sums of their waves' age-class
// weights).  A share's piece records are kept in LDS: at most kSpreadMaxShare chains.
constexpr uint32_t kSpreadMaxShare = 127;
inline uint32_t spread_grid(const DeviceGeom &dg) { return 2 * dg.num_cus; }
__host__ __device__ inline uint32_t spread_share_begin(uint32_t n_chains, uint32_t b, uint32_t grid, uint32_t w1, uint32_t w2)
{
  const uint32_t fh = (grid + 1) / 2;
  const uint64_t total = (uint64_t)fh * w1 + (uint64_t)(grid - fh) * w2;
  const uint64_t cum = b <= fh ? (uint64_t)b * w1 : (uint64_t)fh * w1 + (uint64_t)(b - fh) * w2;
  return (uint32_t)((uint64_t)n_chains * cum / total);
}
// the longest share of a launch on this device (0 = the plan is too small or too large for the launch)
uint32_t spread_longest_share(const DeviceGeom &dg, uint64_t n_chains);

struct IndexArgs
{
  const uint8_t *base;       // base plan blob (device): one single-piece chain per mt_ block
  uint32_t n_base;           // its chains
  uint32_t S, interval;
  const uint32_t *ck_states; // [slot * S]: coder states at absolute group slot * interval
  const uint64_t *ck_words;  // [slot]: absolute stream byte of the read cursor there
  uint32_t *chain_off;       // [n_base] first chain of block b in the new plan (k_index_count)
  uint64_t *result;          // [0] chains of the new plan
  uint8_t *plan;             // the new plan blob (zeroed, sized for max_chains)
  uint32_t max_chains;
  Group *groups;             // [n_base * group_split] or null
  uint32_t group_split;
  uint64_t stream_len;
};
hipError_t launch_index_assemble(const IndexArgs &a, hipStream_t stream);
// 64-bit fingerprint of d_stream[0, stream_len) into *d_sum (16-byte aligned stream; asynchronous: memset + one launch)
hipError_t launch_stream_checksum(const uint8_t *d_stream, uint64_t stream_len, uint64_t *d_sum, hipStream_t stream);

DeviceGeom default_geom(); // MI355X: 256 CUs, 160 KiB LDS (used where no device is at hand: host-side index sizing)
LaunchShape launch_shape(const PlanHeader &h, const DeviceGeom &dg, bool persistent, uint32_t table_mode, uint32_t n_groups, bool index_pass, bool direct, bool dual);
struct TableChoice
{
  uint32_t mode; // 0: none (the kernel builds its own), else kMode* of the host-built table
  bool dual;     // two chains per wave (k_decode_dual)
};
TableChoice choose_table(uint32_t bits, uint32_t states, bool direct);
// chain boundaries (in groups) of the direct launch: one chain per resident wave, sized by class weight; see hsrans_kernels.hip
size_t direct_boundaries(const DeviceGeom &dg, uint32_t states, uint32_t bits, uint64_t total_groups, uint64_t *out, size_t cap);
bool table_spill(); // HSRANS_TABLE_SPILL: leave host-built tables in global memory (comparison only)
// host-side builder of the bits >= 13 coarse/fine decode table (layout: kModeCoarse in hsrans_kernels.hip); returns entries written
size_t build_rank_table(const uint16_t counts[256], uint32_t bits, uint2 *out, size_t capacity_entries);
size_t rank_table_entries(uint32_t bits);
// widest histogram the shared 8-byte-per-slot table (MODE 3) is used for
uint32_t pack64_max_bits();
// per device (call with the device current): raise the dynamic-LDS limit of every kernel variant to the gfx950 maximum
// (160 KiB) and report the device's geometry
hipError_t prepare_kernels(DeviceGeom *geom);
// asynchronous on `stream` of the current device; no allocation, no synchronisation (graph-capturable)
// `parts` (may be null): the launch decodes a rank's sub-runs and publishes a completion word per sub-run (PartArgs).  In: n,
// chain_end[], group_units[k] = groups of the plan's group list that overlap part k.  The launcher picks the kernel, works out how
// many units of that kernel overlap each part, adds them to cum[k] (the running total over all launches of this plan: the
// counters on the device are never reset) and hands the kernel cum[] as its targets.  Only plans the grouped (64 states, lean) or
// the spread kernel takes: hipErrorNotSupported otherwise, nothing launched.
struct PartPlan
{
  uint32_t n;
  const uint32_t *chain_end;
  const uint32_t *group_units;
  uint32_t *cum;
};
// `dealt` (may be null): the plan's shares for k_decode_dealt, from deal_shares with this device's current weights; the launcher takes that
// kernel when it is given (the caller has checked the plan: lean grouped, 64 states, <= 11 bits, no single-symbol blocks)
hipError_t launch_decode(const KParams &kp, const PlanHeader &h, const DeviceGeom &dg, hipStream_t stream, LaunchInfo *info, const PartPlan *parts = nullptr,
                         const DealtTable *dealt = nullptr, const uint32_t *dealt_weights = nullptr /* the 8 class weights `dealt` was made with */);
// The dealing of k_decode_dealt: block k = chains [block_begin[k], block_begin[k + 1]) (n_blocks + 1 entries, the last = n_chains), every one a coded
// block of single-piece mergeable chains.  Workgroup shares by age-class weight (the one-chain-per-wave launch's, this device's own once
// calibrated), each cut back where it would reach into a third block.  false: the plan does not suit the launch (too few chains for the
// device's waves, shares that would have to span more than two blocks, a share beyond 65,535 chains).  weights_out: the 8 class weights used
// (the caller's cache key: a calibration changes them).
bool deal_shares(const DeviceGeom &dg, const uint32_t *block_begin, uint32_t n_blocks, uint32_t n_chains, uint64_t total_groups, uint32_t bits, DealtTable *out,
                 uint32_t weights_out[8]);
void dealt_weights_now(const DeviceGeom &dg, uint64_t run_groups, uint32_t bits, uint32_t weights_out[8]); // run_groups: groups per wave of the launch, on average

} // namespace hsrans

#endif // HSRANS_KERNELS_H
