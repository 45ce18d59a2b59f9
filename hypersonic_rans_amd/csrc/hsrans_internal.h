// hsrans_internal.h — what the translation units behind the C ABI (hsrans_capi*.cpp, hsrans_batch.cpp, hsrans_comm.cpp) share: the
// context and device-plan objects behind the opaque handles of include/hsrans_hip.h, and the few helpers that work on them.
// Not installed; nothing outside csrc/ includes it.
#ifndef HSRANS_INTERNAL_H
#define HSRANS_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/hsrans_hip.h"
#include "hsrans_kernels.h"

using namespace hsrans;

struct hsrans_dplan;
struct hsrans_hpipe;

struct hsrans_ctx
{
  int device = 0;
  char name[256] = {};
  DeviceGeom geom{};     // CU count / LDS of THIS context's device (nothing about a device is process-global)
  bool enc_prepared = false;
  bool enc_raw_prepared = false;
  std::mutex lock; // guards the staging buffers of the host-pointer entries
  std::mutex stream_lock; // creation of pipe_streams (hsrans_hpipe_create may run under `lock` or without it)
  hipStream_t stream = nullptr;
  hsrans_dplan *host_dplan = nullptr; // device plan of the host-pointer entries, refilled per call (buffers are kept)
  // hsrans_decode_host without a plan (the plain decodeFunc): the index the first decode of a stream recorded, kept for the next
  // call on the same stream — {host address, length, codec, 64-bit fingerprint of all its bytes (computed on the device)}
  hsrans_dplan *host_index = nullptr;
  uint64_t host_index_key[4] = {};
  uint8_t host_index_head[128] = {}; // ... and the stream's first bytes, compared on the host first
  uint32_t host_index_head_len = 0;
  hsrans_hpipe *cached_pipe = nullptr; // hsrans_decode_host_pipelined: the pipeline of the plan used last
  uint64_t cached_pipe_key[3] = {};
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
  // the three streams of the host pipelines (upload / decode / download): created once and shared by every hsrans_hpipe of the
  // context.  Streams made per pipe were a trap: the second pipe of a process got streams on ONE hardware queue, its uploads and
  // kernels ran one after the other, and every codec after the first in the harness read 24-26 instead of 33 GiB/s.
  hipStream_t pipe_streams[3] = {nullptr, nullptr, nullptr};
  uint64_t *h_enc_result = nullptr; // page-locked, device-mapped: hsrans_encode_device's kernels write their result words straight into it (no copy kernel, no second launch gap); under `lock`
  uint8_t *h_pin = nullptr; // page-locked staging of hsrans_decode_device_indexing (checkpoints down, plan blob up); under `lock`
  size_t h_pin_cap = 0;
};

// every device plan (and every refill of one) has a number of its own: what hsrans_queue keys its cached batches by — an address can come back
// with another plan behind it
inline uint64_t next_dplan_uid()
{
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1, std::memory_order_relaxed);
}

struct hsrans_dplan
{
  hsrans_ctx *ctx = nullptr;
  uint64_t uid = next_dplan_uid();
  PlanHeader hdr{};
  uint8_t *d_plan = nullptr;
  size_t d_plan_cap = 0;
  uint32_t *d_status = nullptr;
  size_t plan_bytes = 0;
  // dplan_fill puts status word, ticket counters, plan blob, host-built table and group list into ONE device allocation (a
  // device plan used to cost up to five hipMalloc calls and three synchronisations: 3.2 ms for a 2.5 MB index): the pointers
  // below then point into d_arena and are not freed one by one.  Plans written on the device (K2, the GPU encoder) still own theirs.
  uint8_t *d_arena = nullptr;
  size_t d_arena_cap = 0, arena_used = 0;
  uint64_t *d_stamps = nullptr; // diagnostics (HSRANS_DEBUG_STAMPS=1)
  uint64_t *d_finish = nullptr; // hsrans_ctx_calibrate: per-wave finish times of the plan's launches (owned by the calibration)
  unsigned long long *d_counters = nullptr; // uniform persistent launches: kCounterSets sets of monotonic queue heads
  std::atomic<uint32_t> epoch{0};           // launches so far: launch k uses counter set k % kCounterSets
  uint8_t *d_table = nullptr;               // host-built decode table (plans that carry their histogram)
  size_t d_table_cap = 0;
  uint8_t *d_groups = nullptr;              // grouped launches (block_/mt_ plans with checkpoints)
  size_t d_groups_cap = 0;
  uint32_t n_groups = 0;
  bool groups_lean = false; // 64 states, every group a mergeable run or fills only
  uint32_t spread_min_block = 0; // plans k_decode_spread can take (single-piece chains, mergeable / fill groups): the fewest chains of a coded block that is not the last; else 0
  PersistentArgs pa{};
  SingleArgs single{};
  LaunchInfo info{};
  // what the plan's chains touch, recorded by dplan_fill: the lowest stream byte any of them reads (its own words, its
  // histogram / header, the shared histogram when the plan carries no copy of it) and the output bytes they write
  uint64_t body_lo = 0, out_lo = 0, out_hi = 0;
  // a rank's sub-runs decoded by ONE launch of this plan (hsrans_comm.cpp): part k = chains [part_ends[k - 1], part_ends[k]) — set before
  // dplan_fill, which then tags every group with the parts it overlaps (Group::flags, kGroupPartShift) and counts them: part_units[k];
  // part_cum[k] = units counted into part k by all launches so far (the device's counters are never reset: launch_decode, PartPlan)
  std::vector<uint32_t> part_ends, part_units, part_cum;
  // k_decode_dealt (kernels_dealt.h): the plan's blocks as chain ranges — block k = chains [block_begin[k], block_begin[k + 1]) — kept where the
  // plan is a lean grouped one of coded blocks only (no single-symbol blocks); empty otherwise.  `dealt` = the shares for `dealt_weights`
  // (dealt_state 1: valid, -1: the plan does not suit the launch with these weights, 0: not dealt yet); re-dealt when a calibration changes the weights.
  // hsrans_queue: raw plans whose chains start at the same groups are dealt alike in a batch launch — a hash over (states, bits, chain count,
  // every chain's groups), taken when the plan is filled from its host blob; 0 = none (plans written on the device, other containers)
  uint64_t deal_sig = 0;
  std::vector<uint32_t> block_begin;
  DealtTable dealt{};
  uint32_t dealt_weights[8] = {};
  int dealt_state = 0;
};
// fills d->block_begin from the device plan's group list (plans written on the device: the GPU encoder's, an indexing decode's); synchronises `s`
void dplan_blocks_from_device_groups(hsrans_dplan *d, hipStream_t s);

constexpr size_t kStampWaves = 16384;

inline bool grow(uint8_t **p, size_t *cap, size_t need)
{
  if (need <= *cap)
    return true;
  if (*p)
    (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 8 + 4096;
  if (hipMalloc((void **)p, want) != hipSuccess)
  {
    (void)hipGetLastError(); // (consumed here: the runtime's last error is sticky per thread and would surface at an unrelated launch)
    return false;
  }
  *cap = want;
  return true;
}

inline bool grow_pinned(uint8_t **p, size_t *cap, size_t need)
{
  if (need <= *cap)
    return true;
  if (*p)
    (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 4 + 65536;
  if (hipHostMalloc((void **)p, want, hipHostMallocDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  *cap = want;
  return true;
}

inline bool read_header(const uint8_t *plan, size_t size, PlanHeader *h)
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return false;
  memcpy(h, plan, sizeof(PlanHeader));
  return memcmp(h->magic, "HSRPLAN1", 8) == 0;
}

struct hsrans_batch
{
  hsrans_ctx *ctx = nullptr;
  std::vector<hsrans_dplan *> members;
  struct DirectLaunch
  {
    std::vector<uint32_t> member_idx; // batch member of launch-local member i
    BatchShape shape{};
    const BatchMember *d_members = nullptr;
    const BatchSlot *d_slots = nullptr;
    double imbalance = 1.0;
  };
  std::vector<DirectLaunch> direct;
  // block_/mt_ members with checkpoints (64 states, one width <= 12 bits per launch): all their groups in ONE k_decode_grouped_batch launch
  struct GroupedLaunch
  {
    std::vector<uint32_t> member_idx;
    BatchGroupShape shape{};
    const GroupMember *d_members = nullptr;
    const Group *d_groups = nullptr;
    unsigned long long *d_tickets = nullptr; // kCounterSets monotonic ticket counters, one per launch in flight (as a device plan's)
    uint32_t n_groups = 0, bits = 0;
    std::atomic<uint32_t> *epoch = nullptr; // (heap: the struct must stay movable)
  };
  std::vector<GroupedLaunch> grouped;
  std::vector<uint32_t> solo; // members that take a launch of their own (hsrans_decode_device's)
  uint8_t *d_arena = nullptr;
  // per-wave finish times of the first shared launch: diagnostics (HSRANS_BATCH_STAMPS=1: owned by the batch) and
  // hsrans_ctx_calibrate_runs (which points it at its own buffer launch by launch: finish_owned false)
  uint64_t *d_finish = nullptr;
  bool finish_owned = false;
  uint32_t finish_slots = 0;
  std::vector<uint32_t> order_run; // per member: the slot order its chains were dealt with (diagnostics)
};

// (Re)fills a device plan from a validated host plan blob; one launch of a filled device plan (hsrans_capi.cpp)
int dplan_fill(hsrans_dplan *d, const uint8_t *plan, size_t plan_size, const PlanHeader &h, hipStream_t s);
int dplan_launch(hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, hipStream_t s, uint64_t stream_lo = 0,
                 const PartArgs *part_words = nullptr);

// hsrans_decode_device_ranges' body: stream bytes [window_offset, +window_length) at d_window, output bytes [out_offset, +out_length) at d_out;
// part_words: a sharded decode's sub-runs in this one launch (completion words, sequence number; hsrans_comm.cpp)
int dplan_launch_ranges(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_window, size_t window_offset, size_t window_length, void *d_out, size_t out_offset,
                        size_t out_length, void *hip_stream, const PartArgs *part_words);

int dplan_create_with_parts(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, const std::vector<uint32_t> &part_ends, hsrans_dplan **out_dplan);

// a page-locked, device-mapped host range: the address the GPU reaches it at, else null (hsrans_capi.cpp)
uint8_t *device_view_of_host(const void *ptr, size_t bytes);
// hsrans_decode_device_indexing's body; have_lock: the caller holds ctx->lock already (hsrans_capi_index.cpp)
int decode_device_indexing_impl(hsrans_ctx *ctx, hsrans_dplan *d, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                                uint32_t index_interval, void *hip_stream, hsrans_dplan **indexed, bool have_lock);

#endif // HSRANS_INTERNAL_H
