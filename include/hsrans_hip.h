/* hsrans_hip.h — C ABI of libhsrans_hip.so: MI355X (gfx950) implementation of the hypersonic-rANS
 * interleaved 32-bit-state / 16-bit-word decode path (rANS32x32 16w, rANS32x64 16w; raw, block_, mt_ containers).
 *
 * The reference (rainerzufalldererste/hypersonic-rANS @ 2024_10_08) has no FFI layer: its boundary is the set of
 * free C++ functions main.cpp stores in codec_info_t (src/main.cpp:146-155):
 *     size_t encode(const uint8_t *in, size_t len, uint8_t *out, size_t outCap, const hist_t *hist);
 *     size_t decode(const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);      // bytes produced, 0 = failure
 * This header is the C form of exactly that boundary (container / state count / histogram bits become arguments
 * instead of name suffixes) plus what a GPU replacement has to add: a context, a device-pointer
 * entry for device-resident pipelines, and an optional decode plan ("index") that lets many wavefronts work on ONE
 * stream.  include/hsrans_dropin.hpp declares the same functionality under the reference's own C++ names, include/hsrans_names.h
 * under the same names with C linkage (hsrans_<reference name>).
 *
 * Conventions kept from the reference: plain pointers + sizes, caller owns all buffers, return = bytes produced,
 * 0 on any failure, no exceptions cross this boundary, all entry points are re-entrant (per context).
 * The GPU entries (hsrans_decode_host, hsrans_decode_device*, hsrans_hpipe_*, ...) never fall back to the host: without a usable
 * gfx950 device they fail (return 0 / an error code).  The library's own host SIMD decoder sits behind entries that say so in
 * their name — hsrans_decode_cpu, hsrans_index_build_host — and behind the runtime-dispatch names of hsrans_dropin.hpp /
 * hsrans_names.h (`*_decode_auto_N` and the reference's own decoder names), the counterpart of the reference's CPUID dispatch.
 */
#ifndef HSRANS_HIP_H
#define HSRANS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSRANS_RAW 0   /* one histogram, one stream:           src/rANS32x64_16w.cpp, src/rANS32x32_16w.cpp        */
#define HSRANS_BLOCK 1 /* histogram swapped per block, inline: src/block_rANS32x64_16w_{encode,decode}.cpp         */
#define HSRANS_MT 2    /* independent blocks:                  src/mt_rANS32x64_16w_{encode,decode}.cpp            */

/* opaque handles of the GPU side (declared up here so that the host-side prototypes below that take a context are valid C as well) */
typedef struct hsrans_ctx hsrans_ctx;     /* device id, staging buffers, HIP streams/events */
typedef struct hsrans_dplan hsrans_dplan; /* a plan resident in device memory + its status word */

/* = reference hist_t (src/hist.h:16-20) */
typedef struct hsrans_hist
{
  uint16_t symbolCount[256];
  uint16_t cumul[256];
} hsrans_hist;

/* ------------------------------------------------------------------------------------------------------------
 * Host-side format functions (no GPU involved).  The reference's encoders are scalar CPU code too
 * (README.md:19-27 "all encoders scalar"); these produce streams its decoders accept.
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces rANS32x64_16w_capacity (src/rANS32x64_16w.cpp:10), rANS32x32_16w_capacity (src/rANS32x32_16w.cpp:10),
 * block_rANS32x{32,64}_16w_capacity (src/block_rANS32x64_16w_encode.cpp:47), mt_…_capacity (src/mt_rANS32x64_16w_encode.cpp:50) */
size_t hsrans_capacity(int container, int states, size_t input_size);

/* replaces make_hist (src/hist.cpp:217): byte histogram normalised to sum 1 << bits */
void hsrans_make_hist(hsrans_hist *hist, const uint8_t *data, size_t size, uint32_t bits);

/* replaces rANS32x{32,64}_16w_encode_scalar_N (src/rANS32x64_16w.cpp:34), block_…_encode_N
 * (src/block_rANS32x64_16w_encode.cpp:137), mt_…_encode_N (src/mt_rANS32x64_16w_encode.cpp:140).
 * `hist` is used by HSRANS_RAW only (NULL = make_hist over the whole input, as src/main.cpp:746 does). */
size_t hsrans_encode(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t out_capacity,
                     const hsrans_hist *hist);

typedef struct hsrans_encode_opts
{
  uint32_t block_size;       /* block_/mt_: symbols per block (multiple of 64); 0 = the reference's adaptive block policy      */
  uint32_t index_interval;   /* 0 = no plan; else emit a checkpoint every `index_interval` groups of `states` symbols         */
  uint8_t *plan_out;         /* receives the decode plan (see hsrans_plan_*) when index_interval != 0                         */
  size_t plan_capacity;      /* bytes available at plan_out (hsrans_plan_capacity)                                            */
  size_t plan_size;          /* out: bytes written to plan_out                                                                */
  uint32_t flags;            /* HSRANS_ENC_INDEPENDENT_BLOCKS: mt_ only, needs block_size != 0                                 */
  uint32_t reserved;
  const uint64_t *index_groups; /* optional explicit checkpoints: ascending absolute group indices (multiples of 4), e.g. from    */
  size_t n_index_groups;        /* hsrans_index_boundaries; used instead of index_interval when n_index_groups != 0                */
} hsrans_encode_opts;

/* mt_: start every block from fresh states (2^15) instead of carrying the encoder's states across block boundaries as
 * the reference's encoder does (src/mt_rANS32x64_16w_encode.cpp:220-222 initialises them once).  The stream stays a
 * valid mt_ stream (every block header stores its start states anyway) and the blocks become encodable independently —
 * this is the layout hsrans_encode_device produces, so that the two can be compared byte for byte. */
#define HSRANS_ENC_INDEPENDENT_BLOCKS 1u

/* Checkpoint positions that give the decode kernel exactly ONE chain per resident wavefront, each sized by the wave's
 * scheduling class (what the uniform-interval launch otherwise works out per launch): the smallest sidecar that still
 * fills the GPU — 8,192 checkpoints (2.5 MB) for any stream size on an MI355X, against one per `index_interval` groups
 * (15 MB per 100 MB at 32).  `ctx` NULL = the MI355X defaults (256 CUs), so streams can be indexed where they are
 * encoded.  Writes ascending group indices (multiples of 4) to groups_out and returns their count (0 = one chain is all
 * the stream is good for, or capacity too small); pass them as hsrans_encode_opts::index_groups, to hsrans_index_build_at,
 * or to hsrans_plan_thin.  A plan built for another geometry still decodes correctly, only less evenly. */
size_t hsrans_index_boundaries(const hsrans_ctx *ctx, int states, uint32_t bits, size_t decoded_size, uint64_t *groups_out, size_t capacity);

/* same as hsrans_encode, additionally emitting the sidecar decode plan; the stream bytes are unchanged by it */
size_t hsrans_encode_ex(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t out_capacity,
                        const hsrans_hist *hist, hsrans_encode_opts *opts);

/* ------------------------------------------------------------------------------------------------------------
 * Decode plans.  A plan lists the independent chains (start states, read cursor, output range, histogram location)
 * one wavefront each will decode.  hsrans_plan_build derives it from the stream alone:
 *   raw    -> 1 chain (the format has no restart points: src/rANS32x64_16w.cpp:223-250 is one dependent chain);
 *   mt_    -> 1 chain per block, by following the header chain like src/mt_rANS32x64_16w_decode.cpp:166-227;
 *   block_ -> 1 chain that parses the inline headers on the device (src/block_rANS32x64_16w_decode.cpp:47-90).
 * A plan emitted by hsrans_encode_ex (or hsrans_index_build) additionally splits chains every `index_interval`
 * groups, which is what lets one stream fill the GPU.  Plans are position-independent byte blobs.
 * ---------------------------------------------------------------------------------------------------------- */
size_t hsrans_plan_capacity(int container, int states, size_t decoded_size, uint32_t index_interval, uint32_t block_size);
/* the same for a plan with `extra_chains` checkpoints (explicit index_groups) */
size_t hsrans_plan_capacity_chains(int container, int states, size_t decoded_size, size_t extra_chains, uint32_t block_size);
size_t hsrans_plan_build(int container, int states, uint32_t bits, const uint8_t *stream, size_t stream_length, size_t out_capacity,
                         uint8_t *plan_out, size_t plan_capacity);
uint32_t hsrans_plan_chain_count(const uint8_t *plan, size_t plan_size);
uint64_t hsrans_plan_decoded_length(const uint8_t *plan, size_t plan_size);
/* restrict a plan to chains [first, first+count): used to shard one stream over several GPUs (each rank decodes its
 * chains into the same output offsets of its own buffer).  Returns bytes written, 0 on error. */
size_t hsrans_plan_slice(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint8_t *out, size_t out_capacity);
/* output byte range [*begin, *end) covered by chains [first, first+count) */
int hsrans_plan_chain_range(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint64_t *begin, uint64_t *end);
/* Thin a raw-stream plan: keep only the chains that start at (or, when a boundary is not a chain start of `plan`, at the
 * last chain start before) the given ascending group indices, merging everything in between — a finer plan (e.g. one
 * checkpoint per 32 groups, good for any split over GPUs) becomes the one-chain-per-wave plan of one GPU.  Returns bytes
 * written, 0 on error (not a mergeable raw plan, capacity). */
size_t hsrans_plan_thin(const uint8_t *plan, size_t plan_size, const uint64_t *groups, size_t n_groups, uint8_t *out, size_t out_capacity);
/* stream byte ranges chains [first, first+count) can read: ranges = {head_begin, head_end, body_begin, body_end}; head is the
 * shared histogram of a raw stream (empty otherwise), body the chains' own headers and words up to the next chain's first
 * word.  A rank that decodes only these chains needs only these bytes of the stream in its HBM (at their stream offsets). */
int hsrans_plan_stream_ranges(const uint8_t *plan, size_t plan_size, uint32_t first_chain, uint32_t chain_count, uint64_t ranges[4]);

/* ------------------------------------------------------------------------------------------------------------
 * Host SIMD decoders with runtime dispatch (scalar / AVX2 / AVX-512) — the counterpart of the reference's CPU decoders and
 * of its dispatcher block_rANS32x64_decode_wrapper (src/block_rANS32x64_16w_decode.cpp:130-152), written from the format.
 * The GPU cannot speed up a stream that is one dependent chain (raw / block_ without an index: one wavefront, a third of a
 * CPU core); these serve that case in the `*_decode_auto_N` drop-in entries, build indexes of foreign streams faster than
 * one wavefront can, and are the in-run CPU comparator.  The GPU entries below never fall back to them.
 * `level`: 0 scalar, 1 AVX2, 2 AVX-512, -1 = the best this host supports.  `threads` >= 1 (independent chains / mt_ blocks
 * are spread over std::threads like the reference's thread pool, src/mt_rANS32x64_16w_decode.cpp:217-220).
 * ---------------------------------------------------------------------------------------------------------- */
int hsrans_cpu_level(void); /* 0 / 1 / 2: what hsrans_decode_cpu(level = -1) uses here */
/* replaces rANS32x{32,64}_16w_decode_{scalar,avx2_*,avx512_*}_N, block_…_decode_N, mt_…_decode[_mt]_N on the host */
size_t hsrans_decode_cpu(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out,
                         size_t out_capacity, const uint8_t *plan, size_t plan_size);
/* hsrans_index_build_at without a GPU: one (per mt_ block: parallel) host decode pass that records the checkpoints */
size_t hsrans_index_build_host(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length,
                               const uint64_t *groups, size_t n_groups, uint8_t *plan_out, size_t plan_capacity);

/* ------------------------------------------------------------------------------------------------------------
 * GPU side
 * ---------------------------------------------------------------------------------------------------------- */
#define HSRANS_OK 0
#define HSRANS_E_NO_DEVICE 1
#define HSRANS_E_ARG 2
#define HSRANS_E_FORMAT 3   /* malformed stream / plan (the reference's "return 0" cases) */
#define HSRANS_E_HIP 4
#define HSRANS_E_DEVICE 5   /* the kernel reported a malformed histogram / block header */

int hsrans_ctx_create(int device, hsrans_ctx **out_ctx);
void hsrans_ctx_destroy(hsrans_ctx *ctx);
const char *hsrans_ctx_device_name(const hsrans_ctx *ctx);
/* chains of the index hsrans_decode_host keeps from its last plan-less call (0 = none): see hsrans_decode_host */
uint32_t hsrans_ctx_host_index_chains(hsrans_ctx *ctx);

/* Host-pointer drop-in for decodeFunc (src/main.cpp:149): H2D, plan, launch, D2H.  Replaces
 * rANS32x{32,64}_16w_decode_*_N, block_rANS32x{32,64}_16w_decode_N, mt_rANS32x{32,64}_16w_decode[_mt]_N.
 * Returns the decoded length, 0 on failure. `plan` may be NULL (derived from the stream).
 * With plan == NULL, mt_ and raw streams of >= 1 MiB leave an index behind: the first call's decode records checkpoints (as
 * hsrans_decode_device_indexing) and the context keeps the plan; a later call with the same `in`, in_length and codec launches it
 * — but only counts when the stream's first 128 bytes (compared on the host) and a 64-bit fingerprint of ALL stream bytes, computed on
 * the device beside the decode, equal the first call's (otherwise the call starts over; other bytes at the same address cost one
 * wasted launch).  The fingerprint is a mixing hash, not a MAC: it guards against ACCIDENTAL reuse of a buffer (another file read into
 * the same allocation), where a false match needs a 2^-64 coincidence; bytes crafted to collide with it would decode with the previous
 * stream's checkpoints — memory-safe (every read stays inside in_length) but wrong.  Callers that decode hostile streams through this
 * entry switch the cache off (HSRANS_HOST_INDEX_CACHE_OFF=1) or pass their own plan.  A loop over one file (src/main.cpp:860-889) thus runs the indexed kernels from its second iteration:
 * 100 MB mt_: 0.24 -> 0.06 ms of kernel per call; raw: 125 ms -> 0.04 ms (a raw stream's index comes from the host SIMD decoder's
 * pass, ~30 ms for 100 MB, during the first call — the one host-side step of this entry; HSRANS_HIP_STRICT=1: from the wavefront that
 * decodes the stream instead, ~125 ms once).  HSRANS_HOST_INDEX_CACHE_OFF=1 disables the cache: every call decodes what the stream alone
 * allows (a raw stream: one wavefront, k_decode_single). */
size_t hsrans_decode_host(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out,
                          size_t out_capacity, const uint8_t *plan, size_t plan_size);

/* Device-resident entry: everything asynchronous on `hip_stream` (a hipStream_t of the context's device; NULL = default
 * stream), graph-capturable (no allocation, no synchronisation).  `d_stream` must be 16-byte aligned, `d_out` 4-byte aligned.
 * Overlapping launches of ONE device plan (several streams, double-buffered outputs) are fine: plans with one chain per
 * wave (hsrans_index_boundaries) and mt_ plans without checkpoints keep no per-launch state on the device; uniform-interval
 * raw plans and block_/mt_ plans with checkpoints draw work from atomic ticket counters and own 32 sets of them, used round
 * robin, so up to 32 of their launches may be in flight at once — a captured graph node keeps the set it was captured with,
 * so replays of one captured graph must not overlap each other (HIP does not allow a hipGraphExec to run concurrently with
 * itself anyway).  The status word of a plan is shared by all its launches (error bits are only ever OR-ed in;
 * hsrans_dplan_status clears them). */
int hsrans_dplan_create(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, hsrans_dplan **out_dplan);
void hsrans_dplan_destroy(hsrans_dplan *dplan);
int hsrans_decode_device(hsrans_ctx *ctx, hsrans_dplan *dplan, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                         void *hip_stream);
/* The same launch for a caller that holds only a WINDOW of the stream in device memory — the bytes
 * [window_offset, window_offset + window_length) at d_window — e.g. one GPU's share of a stream sharded with
 * hsrans_plan_slice: the window must cover the body range hsrans_plan_stream_ranges reports for the plan's chains
 * (window_offset a multiple of 16, at or below body_begin: HSRANS_E_FORMAT otherwise — the lowest stream byte the plan's
 * chains read is recorded when the device plan is made); a raw stream's shared histogram need not be in it when the plan
 * carries its copy.  Requests beyond the window's end are dropped by the kernel's bounds check, never issued. */
int hsrans_decode_device_window(hsrans_ctx *ctx, hsrans_dplan *dplan, const void *d_window, size_t window_offset, size_t window_length, void *d_out,
                                size_t out_capacity, void *hip_stream);
/* ... and only a WINDOW of the output: d_out_window receives the decoded bytes [out_offset, out_offset + out_length)
 * (out_offset a multiple of 4; the plan's chains must write inside it, hsrans_plan_chain_range: HSRANS_E_FORMAT
 * otherwise).  A rank of a sharded decode then needs neither the whole stream nor the whole output in its HBM: the
 * sender side of the gather in src/mt_rANS32x64_16w_decode.cpp:217-220's fan-out, one GPU per run of blocks. */
int hsrans_decode_device_ranges(hsrans_ctx *ctx, hsrans_dplan *dplan, const void *d_window, size_t window_offset, size_t window_length, void *d_out_window,
                                size_t out_offset, size_t out_length, void *hip_stream);
/* Plan an mt_ stream that only exists in device memory: the header chain (src/mt_rANS32x64_16w_decode.cpp:166-227) is
 * followed by a device kernel; synchronises `hip_stream` twice (chain count, then the finished plan). HSRANS_MT only. */
int hsrans_dplan_create_from_device_stream(hsrans_ctx *ctx, int container, int states, uint32_t bits, const void *d_stream, size_t stream_length,
                                           size_t out_capacity, void *hip_stream, hsrans_dplan **out_dplan);
/* copies the plan blob a dplan holds in device memory back to the host (inspection / tests); returns its size, 0 on error */
size_t hsrans_dplan_read_plan(hsrans_dplan *dplan, uint8_t *out, size_t capacity);
/* synchronises `hip_stream` and returns HSRANS_OK or HSRANS_E_DEVICE (kernel found a bad histogram/header) */
int hsrans_dplan_status(hsrans_ctx *ctx, hsrans_dplan *dplan, void *hip_stream);

/* ------------------------------------------------------------------------------------------------------------
 * K independent streams, ONE launch.  The reference decodes independent work from one pool — a task per mt_ block on its
 * thread pool (src/mt_rANS32x64_16w_decode.cpp:182-224), file after file in its benchmark loop (src/main.cpp:841-898); on the
 * GPU the pool is the device's resident wave slots.  One launch per stream pays the launch's prologue, tail and kernel boundary
 * (~11 of ~41 us for a 100 MB stream) K times, and launches put on several HIP streams cannot co-reside (every plan is shaped
 * to fill the device).  A batch deals the wave slots of ONE launch to its members — whole workgroups (a workgroup holds one
 * decode table), in proportion to the members' sizes, each member's chains cut into runs sized by the slots' scheduling
 * class — so the three are paid once.  Members keep their own device plans, status words and results.  block_/mt_ members with
 * checkpoints (64 states, <= 12 bits) share a launch of their own kind: all their blocks in one list, a workgroup per block and round —
 * many small streams then fill the device together instead of each launching a mostly empty one.  Raw members come in four kinds
 * with a shared kernel each (64 states <= 12 bits; 32 states <= 12 bits; 64 states 13 bits; 64 states 14-15 bits): a call makes one
 * launch per kind present.  A member no shared kernel takes (an un-indexed stream, a lone member of its kind) gets its own launch
 * behind them, in the same call.
 * Best served: raw streams of <= 12 bits (64 states, or 32: a launch per state count) with a uniform index
 * (hsrans_encode_opts::index_interval: any mix of sizes), with the index shaped for the batch (hsrans_index_boundaries_batch), or with
 * the one-chain-per-wavefront index of a launch of their own (hsrans_index_boundaries) when the members are 1, 2 or 4 of one size.
 * The dplans must outlive the batch and must not be refilled while it exists; a dplan belongs to at most one member.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct hsrans_batch hsrans_batch;
int hsrans_dplan_batch_create(hsrans_ctx *ctx, hsrans_dplan *const *dplans, uint32_t count, hsrans_batch **out_batch);
void hsrans_dplan_batch_destroy(hsrans_batch *batch);
/* member k: stream d_streams[k] (16-byte aligned, stream_lengths[k] bytes) -> d_outs[k] (4-byte aligned, out_capacities[k]).
 * Asynchronous on `hip_stream`, no allocation, no synchronisation; nothing is launched unless every member's arguments pass. */
int hsrans_decode_device_batch(hsrans_ctx *ctx, hsrans_batch *batch, const void *const *d_streams, const size_t *stream_lengths, void *const *d_outs,
                               const size_t *out_capacities, void *hip_stream);
/* synchronises `hip_stream`; member_status[k] (may be NULL) = hsrans_dplan_status of member k; returns the first failure or HSRANS_OK */
int hsrans_dplan_batch_status(hsrans_ctx *ctx, hsrans_batch *batch, void *hip_stream, int *member_status);
typedef struct hsrans_batch_info
{
  uint32_t members, launches;            /* kernel launches one hsrans_decode_device_batch call makes */
  uint32_t direct_members, solo_members; /* members in the shared one-chain-per-wave launch(es) / with a launch of their own */
  uint32_t grouped_members, reserved;    /* block_/mt_ members with checkpoints sharing a grouped launch (one per histogram width) */
  uint32_t grid, block, lds_bytes;       /* the (first) shared launch */
  uint32_t class_weights[8];             /* per-mille run lengths of the 8 wave scheduling classes it was dealt with */
  double imbalance;                      /* its most loaded wave slot (groups / class weight) over the mean: 1.0 = all waves end together */
} hsrans_batch_info;
int hsrans_dplan_batch_info(const hsrans_batch *batch, hsrans_batch_info *info);
/* hsrans_index_boundaries for a stream that will be decoded as member `member` of a batch of `count` streams of the given decoded
 * sizes: exactly one chain per wave slot the batch launch deals that member (8,192 / count each for streams of one size: the sidecar
 * shrinks with the batch), sized by the slots' scheduling classes at the batch's run length.  64 states, bits 10..15, or 32 states, bits
 * 10..12 (the sizes given are those of the members of the same kind: see hsrans_dplan_batch_create).  A batch of
 * plans made this way is dealt without rounding (hsrans_batch_info::imbalance ~ 1.00); the plans still decode alone, or in another
 * batch, only less evenly.  32 states: two chains per wave slot.  Returns the number of group indices written (0 = one chain, or
 * capacity too small). */
size_t hsrans_index_boundaries_batch(const hsrans_ctx *ctx, int states, uint32_t bits, const size_t *decoded_sizes, uint32_t count, uint32_t member,
                                     uint64_t *groups_out, size_t capacity);
/* The dealing alone, without a device (tests, planning): members' chain starts in groups (chain_starts[m][0 .. n_chains[m]], the last
 * entry = all the member's groups) -> slots_out[4 * (wg * waves + wave)] = {member, first chain, end chain, flags}; returns the
 * imbalance (see above), < 0 on bad arguments.  weights NULL = the MI355X defaults of the 64-state launch. */
double hsrans_batch_deal(const uint64_t *const *chain_starts, const uint32_t *n_chains, uint32_t members, uint32_t grid, uint32_t waves, const uint32_t *weights,
                         uint32_t *slots_out);
/* diagnostics: with HSRANS_BATCH_STAMPS=1 in the environment when the batch is made, every wave of its first shared launch leaves
 * its finish time (100 MHz) in slot wg * waves + wave, the launch's first wave its entry time in the slot after the last */
size_t hsrans_dplan_batch_read_finish(hsrans_batch *batch, uint64_t *out, size_t capacity_u64);

/* ------------------------------------------------------------------------------------------------------------
 * Open-ended submission: streams that arrive one by one.  The reference's pool takes tasks as they come (thread_pool_add,
 * src/thread_pool.cpp:124-133) and its benchmark loop hands it file after file (src/main.cpp:841-898, :163-170); hsrans_dplan_batch_create
 * wants all members at once.  A queue collects submissions on the host (a submit launches nothing, except that the `max_members`-th
 * pending one flushes) and hsrans_queue_flush decodes everything pending with ONE batch launch.  What a batch costs to make (about a
 * millisecond) is paid once per shape: the queue keeps the last 8 batches it made; the same device plans in the same order reuse theirs
 * as it is, and OTHER raw plans of the same shapes (same decoded size and index geometry: a loop over files of one size class) reuse
 * its dealing — only a 64-byte record per member is rewritten, asynchronously in front of the launch.  Use one `hip_stream` per queue
 * at a time.  The buffers must stay valid until the flush's launch has run; per-member results: hsrans_dplan_status on the plans.
 * A plan submitted again while it is still pending flushes first (a plan is one member of a launch).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct hsrans_queue hsrans_queue;
typedef struct hsrans_queue_stats_t
{
  uint64_t submitted, flushes, launches; /* kernel launches made by the flushes */
  uint64_t batches_made, batches_reused; /* flushes that had to make a batch / found theirs in the cache */
  uint64_t retargeted;                   /* ... of the reused ones: with other plans of the same shapes */
} hsrans_queue_stats_t;
int hsrans_queue_create(hsrans_ctx *ctx, uint32_t max_members /* 1..32 */, hsrans_queue **out_queue);
void hsrans_queue_destroy(hsrans_queue *queue); /* synchronises the device */
int hsrans_queue_submit(hsrans_queue *queue, hsrans_dplan *dplan, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity, void *hip_stream);
int hsrans_queue_flush(hsrans_queue *queue, void *hip_stream); /* asynchronous on hip_stream; nothing pending: HSRANS_OK, nothing launched */
uint32_t hsrans_queue_pending(const hsrans_queue *queue);
int hsrans_queue_stats(const hsrans_queue *queue, hsrans_queue_stats_t *out);

/* ------------------------------------------------------------------------------------------------------------
 * ONE stream over the GPUs of a node: one process per GPU, the plan's chains cut into one contiguous run per rank, the decoded
 * ranges exchanged point to point over xGMI (RCCL ncclSend / ncclRecv groups on the communicator's own HIP stream), pipelined behind
 * the decode in `parts` sub-runs.  This is the C form of the reference's thread-pool fan-out behind the `thread_pool *` argument of
 * mt_rANS32x64_16w_decode_mt_N (src/mt_rANS32x64_16w.h:23-28): blocks handed to workers (src/mt_rANS32x64_16w_decode.cpp:217-220),
 * joined by thread_pool_await (:262) — a worker is a GPU, a block a chain, the join the exchange.  RCCL is bound at run time
 * (the librccl the process already carries, else the system's); without one these entries return HSRANS_E_NO_DEVICE.
 * Rendezvous: rank 0 calls hsrans_comm_unique_id and ships the 128 bytes to the other ranks by whatever it has (MPI, a file, a
 * socket); every rank then calls hsrans_comm_create with the same bytes (collective: returns when all `world` ranks have called).
 * ---------------------------------------------------------------------------------------------------------- */
#define HSRANS_COMM_ID_BYTES 128
typedef struct hsrans_comm hsrans_comm;
int hsrans_comm_unique_id(uint8_t id[HSRANS_COMM_ID_BYTES]);
int hsrans_comm_create(hsrans_ctx *ctx, const uint8_t id[HSRANS_COMM_ID_BYTES], int rank, int world, hsrans_comm **out_comm);
void hsrans_comm_destroy(hsrans_comm *comm);
int hsrans_comm_rank(const hsrans_comm *comm);
int hsrans_comm_world(const hsrans_comm *comm);
int hsrans_comm_rccl_version(void); /* 0: no RCCL could be bound */

/* Which chains / output bytes / stream bytes each rank owns, and how each rank's run is cut into `parts` sub-runs: pure host
 * arithmetic, identical on every rank, no GPU and no RCCL involved.  The chains are cut into `world` contiguous runs whose decoded
 * bytes follow `weights` (NULL = equal shares; e.g. a larger share for the rank a gather goes to, which sends nothing), every run
 * into `parts` sub-runs of equal decoded bytes.  shards[rank * parts + k] = sub-run k of `rank`; windows (may be NULL) receives
 * {begin, end} of the stream bytes each rank's chains read (begin aligned down to 16: hsrans_decode_device_window). */
typedef struct hsrans_shard
{
  uint32_t first_chain, chain_count; /* chain_count 0: nothing (fewer chains than sub-runs) */
  uint64_t out_begin, out_end;       /* decoded bytes [out_begin, out_end) */
} hsrans_shard;
int hsrans_shard_layout(const uint8_t *plan, size_t plan_size, uint32_t world, uint32_t parts, const double *weights, hsrans_shard *shards /* [world * parts] */,
                        uint64_t *windows /* [2 * world] or NULL */);

/* Everything that does not change between decodes of one (plan, communicator) pair, prepared once: this rank's sub-runs as device
 * plans, its stream window, the output bytes it holds.  root < 0: every rank ends with the whole output; root >= 0: that rank only
 * (the others then hold just their own range: out_base / out_length below). */
typedef struct hsrans_sharded hsrans_sharded;
int hsrans_sharded_create(hsrans_ctx *ctx, hsrans_comm *comm, const uint8_t *plan, size_t plan_size, uint32_t parts, const double *weights, int root,
                          hsrans_sharded **out_sharded);
/* one rank's share WITHOUT a communicator (decode only: hsrans_decode_sharded with HSRANS_SHARD_DECODE_ONLY) — a consumer that is
 * sharded like the decode, or tests that run every rank's GPU side on one GPU */
int hsrans_sharded_create_rank(hsrans_ctx *ctx, int rank, int world, const uint8_t *plan, size_t plan_size, uint32_t parts, const double *weights, int root,
                               hsrans_sharded **out_sharded);
void hsrans_sharded_destroy(hsrans_sharded *sharded);
typedef struct hsrans_sharded_info_t
{
  uint32_t world, rank, parts;
  int32_t root;
  uint64_t window_begin, window_end; /* stream bytes this rank must hold at d_window (d_window[0] = stream byte window_begin) */
  uint64_t out_base, out_length;     /* output bytes this rank's d_out holds (d_out[0] = output byte out_base) */
  uint64_t decoded_length, stream_length;
  uint32_t one_launch, reserved;     /* 1: this rank's sub-runs are decoded by ONE launch that publishes a completion word per sub-run */
} hsrans_sharded_info_t;
int hsrans_sharded_info(const hsrans_sharded *sharded, hsrans_sharded_info_t *info, hsrans_shard *shards /* [world * parts] or NULL */, size_t shard_capacity);
hsrans_dplan *hsrans_sharded_part_plan(hsrans_sharded *sharded, uint32_t part); /* this rank's sub-run `part` (launch info, status); NULL: no chains, or one launch for all */
hsrans_dplan *hsrans_sharded_whole_plan(hsrans_sharded *sharded);               /* this rank's whole run when ONE launch decodes all its sub-runs; NULL otherwise */
/* One decode of the stream: this rank's sub-runs are queued on `hip_stream`; with gather != 0 sub-run k's ranges go onto the links
 * (the communicator's stream waits for exactly that sub-run) while sub-run k + 1 decodes, and `hip_stream` continues when
 * the transfers are done.  block_/mt_ plans with checkpoints (64 states, <= 16 sub-runs): the sub-runs are ONE launch — every block of
 * the rank's run handed to the device in one pass, as src/mt_rANS32x64_16w_decode.cpp:182-224 hands them to its pool — that publishes a
 * completion word per sub-run, and the communicator's stream waits for the word (hipStreamWaitValue32); other plans, or
 * HSRANS_SHARD_ONE_LAUNCH=0 in the environment at hsrans_sharded_create: a launch per sub-run, an event behind each.  A receiving root posts all its receives first.  Asynchronous; collective when gather != 0 (every rank
 * of the communicator must make the same call). */
#define HSRANS_SHARD_DECODE_ONLY 0         /* every rank keeps its range */
#define HSRANS_SHARD_DECODE_AND_EXCHANGE 1 /* the step: decode, ranges exchanged behind it */
#define HSRANS_SHARD_EXCHANGE_ONLY 2       /* the ranges of an earlier decode-only call (to time the two legs apart) */
int hsrans_decode_sharded(hsrans_sharded *sharded, const void *d_window, void *d_out, int gather, void *hip_stream);
/* Makes `hip_stream` (another stream than the decode's) wait until sub-run `part` of the LAST hsrans_decode_sharded call queued on this
 * object is decoded and its bytes are visible device-wide — what the exchange does internally, for a consumer that is sharded like the
 * decode and can start on a sub-run while the next one is still being decoded (a worker of the reference's pool picking up finished
 * blocks: src/thread_pool.cpp:124-133).  One launch for all sub-runs: waits for the sub-run's completion word; else for its event. */
int hsrans_sharded_wait_part(hsrans_sharded *sharded, uint32_t part, void *hip_stream);
int hsrans_sharded_status(hsrans_sharded *sharded, void *hip_stream); /* as hsrans_dplan_status, over this rank's sub-runs */

/* First decode of a stream that came WITHOUT an index (e.g. a reference-emitted mt_ stream: one chain per block,
 * src/mt_rANS32x64_16w_decode.cpp:137-265 decodes it with one thread per block): decodes like hsrans_decode_device with
 * `dplan` (from hsrans_plan_build + hsrans_dplan_create, or from hsrans_dplan_create_from_device_stream; HSRANS_RAW and
 * HSRANS_MT, no checkpoints yet) and records the coder states and read cursor every `index_interval` groups (multiple of 4)
 * on the way.  *indexed receives a device plan with those checkpoints (the blob hsrans_index_build would return for the
 * same stream and interval) for every later decode of the same stream.  Synchronises `hip_stream`.  d_out is complete on
 * return.  HSRANS_E_DEVICE: the pass found a malformed stream (status cleared), *indexed is NULL. */
int hsrans_decode_device_indexing(hsrans_ctx *ctx, hsrans_dplan *dplan, const void *d_stream, size_t stream_length, void *d_out, size_t out_capacity,
                                  uint32_t index_interval, void *hip_stream, hsrans_dplan **indexed);

/* GPU encoder: mt_ stream with fixed blocks of `block_size` symbols (multiple of 64), each block
 * encoded independently by one wavefront (HSRANS_ENC_INDEPENDENT_BLOCKS layout).  d_in / d_out are device pointers (16-byte
 * aligned); d_out needs hsrans_capacity(HSRANS_MT, states, length) bytes.  Synchronises `hip_stream`; returns the stream
 * size, 0 on failure.  Byte-identical to hsrans_encode_ex(HSRANS_MT, ..., {block_size, index_interval, flags =
 * HSRANS_ENC_INDEPENDENT_BLOCKS}), stream and plan.
 * out_dplan != NULL: the stream's decode plan is written on the device as well — one chain per block plus one per
 * checkpoint every `index_interval` groups inside the blocks (multiple of 4; 0 = block starts only), the encoder passes
 * through exactly those states — and returned as a device plan for hsrans_decode_device (hsrans_dplan_read_plan fetches
 * the blob).  Nothing leaves HBM between encode and decode. */
size_t hsrans_encode_device(hsrans_ctx *ctx, int container, int states, uint32_t bits, const void *d_in, size_t length, void *d_out, size_t out_capacity,
                            uint32_t block_size, uint32_t index_interval, void *hip_stream, hsrans_dplan **out_dplan);

/* The raw format on the GPU (replaces src/rANS32x64_16w.cpp:34-166 `rANS32x64_16w_encode_scalar_N` / src/rANS32x32_16w.cpp for
 * data that lives in HBM): the format carries every coder state from the last symbol to the first, so it
 * is ONE wavefront's work — about 3.5x one host core, not a throughput kernel; mt_ (hsrans_encode_device) is the format to encode
 * at HBM rates.  `hist` NULL: the histogram of the input is counted and normalised on the device (what make_hist gives); else the
 * caller's normalised counts are used (they must sum to 1 << bits), as the reference's encode signature passes them.
 * The sidecar index: a checkpoint every `index_interval` groups, or at `index_groups` (hsrans_index_boundaries) when
 * n_index_groups != 0; the plan goes to plan_out (host; *plan_size receives its size; hsrans_plan_capacity[_chains]) and/or is
 * returned as a device plan.  Stream and plan are byte-identical to hsrans_encode_ex's.  length <= 2^31 - 2^16.
 * hsrans_encode_device(HSRANS_RAW, ...) is this call with hist = NULL and a uniform interval.  Returns the stream length, 0 on failure. */
size_t hsrans_encode_device_raw(hsrans_ctx *ctx, int states, uint32_t bits, const void *d_in, size_t length, void *d_out, size_t out_capacity, const hsrans_hist *hist,
                                uint32_t index_interval, const uint64_t *index_groups, size_t n_index_groups, uint8_t *plan_out, size_t plan_capacity,
                                size_t *plan_size, void *hip_stream, hsrans_dplan **out_dplan);

/* Build a plan with checkpoints every `index_interval` groups for an EXISTING stream (e.g. one written by the
 * reference's encoder) by one decode pass on the GPU that records the states at the checkpoints: HSRANS_RAW (one
 * sequential wavefront), HSRANS_MT (one wavefront per block) and HSRANS_BLOCK (one sequential wavefront that also reports
 * the inline block headers it meets: the plan then has a chain per block and per checkpoint, which is what makes a block_
 * stream chunk-parallel; blocks must average >= 4 KiB).  Returns plan bytes written to plan_out (host memory), 0 on
 * failure. */
size_t hsrans_index_build(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length,
                          uint32_t index_interval, uint8_t *plan_out, size_t plan_capacity);

/* as hsrans_index_build, with checkpoints at the given ascending group indices (raw streams; hsrans_index_boundaries) */
size_t hsrans_index_build_at(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length,
                             const uint64_t *groups, size_t n_groups, uint8_t *plan_out, size_t plan_capacity);

/* ------------------------------------------------------------------------------------------------------------
 * Host buffers, PCIe legs overlapped (BASELINE config 5 shape; the GPU counterpart of the reference's thread-pool fan-out,
 * src/mt_rANS32x64_16w_decode.cpp:137-265): the plan's chains are cut into `n_slices` runs; slice k's compressed bytes go
 * up on one HIP stream while slice k-1 decodes on a second and slice k-2's output comes down on a third (page-lock `in` and
 * `out` — hipHostMalloc / hipHostRegister / hsrans_host_register — or the runtime stages the copies and the legs serialise).
 * 2^30 bytes: 22.6 ms end to end = 47.4 GB/s decoded; 100 MB: 36-39 k MiB/s for every codec.  HSRANS_HPIPE_DIRECT=1 in the
 * environment makes the decode kernels store STRAIGHT into a page-locked `out` instead (no device-side output buffer, no
 * download copies): the same rate at 2^30 bytes, 29-34 k MiB/s at 100 MB.  The three streams belong to the context.
 * The slice plans live on the device for the lifetime of the pipeline object; a pipe serves one decode at a time (calls
 * serialise on it).  hsrans_decode_host_pipelined is the one-call form: it keeps the pipeline of the plan it saw last inside
 * the context (keyed by the plan's address, size and a sampled checksum).  On any failure every entry returns only after all work it queued has drained.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct hsrans_hpipe hsrans_hpipe;
int hsrans_hpipe_create(hsrans_ctx *ctx, const uint8_t *plan, size_t plan_size, uint32_t n_slices /* 0 = by size: 2..16 slices of >= 16 MiB of output */, hsrans_hpipe **out_pipe);
size_t hsrans_hpipe_decode(hsrans_hpipe *pipe, const uint8_t *in, size_t in_length, uint8_t *out, size_t out_capacity);
void hsrans_hpipe_destroy(hsrans_hpipe *pipe);
size_t hsrans_decode_host_pipelined(hsrans_ctx *ctx, int container, int states, uint32_t bits, const uint8_t *in, size_t in_length, uint8_t *out,
                                    size_t out_capacity, const uint8_t *plan, size_t plan_size, uint32_t n_slices);
/* page-lock / release a caller-owned host buffer (hipHostRegister on the context's device); 0 on success */
int hsrans_host_register(hsrans_ctx *ctx, void *ptr, size_t bytes);
int hsrans_host_unregister(hsrans_ctx *ctx, void *ptr);

/* Per-device fit of the one-chain-per-wave index (hsrans_index_boundaries): the SIMDs serve their oldest wave first, so the 8
 * scheduling classes of waves decode at different rates and get chains of different lengths; the lengths compiled in were fitted
 * on one box and are a few per cent off on others.  This measures the classes' finish times on the context's own device (48 MiB
 * of synthetic enwik8-shaped bytes, encoded once on the host; `iterations` (0 = 4) rounds of index -> launches -> adjust; about
 * half a second) and keeps the best lengths in the context: hsrans_index_boundaries(ctx, ...) and the launch info use them from
 * then on.  64 states, bits 10..12 (the 8-byte-table kernel).  Returns HSRANS_OK; `report` may be NULL.  Synchronises the
 * device; call it before the context is shared between threads (it rewrites the context's geometry while it runs).  Indexes made
 * before and after stay valid on any device: the lengths only decide how evenly a launch finishes. */
typedef struct hsrans_calibration
{
  uint32_t class_weights[8];                 /* per-mille chain lengths kept */
  double class_finish_us_last_iteration[8];  /* mean finish time of each class in the last iteration measured */
  double last_wave_us_before, last_wave_us_after;       /* when the launch's last wave was done: first iteration / best iteration */
  double class_spread_us_before, class_spread_us_after; /* latest minus earliest class mean */
  uint32_t iterations, reserved;
  uint64_t bytes;
} hsrans_calibration;
int hsrans_ctx_calibrate(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, hsrans_calibration *report);
/* The same fit at another RUN LENGTH.  A wave's time is its prologue plus its groups at its class's rate, so the chain lengths that
 * make the classes finish together depend on how long the runs are (an old wave's head start counts for less in a long run):
 * hsrans_ctx_calibrate fits runs of ~96 groups (its 48 MiB stream over 8,192 waves); this call decodes `copies` (1..16) members
 * of that stream in ONE batch launch — runs of copies x 96 groups — and keeps the fitted lengths as the set of that run length.
 * hsrans_index_boundaries[_batch] and the batch dealing interpolate between the fitted run lengths (log scale; the nearest set
 * outside them); up to 4 sets per context.  E.g. copies = 2 for 100 MB streams decoded alone, 8 for four of them in one launch. */
int hsrans_ctx_calibrate_runs(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, uint32_t copies, hsrans_calibration *report);

/* kernel launch geometry of the last hsrans_decode_device call on this plan (for benchmarks / DESIGN.md tables) */
typedef struct hsrans_launch_info
{
  uint32_t grid, block, lds_bytes, waves_per_block, chains, shared_table, walk, two_level;
  uint32_t table_mode; /* decode-table layout: 0/1 packed u32, 2 two-level, 3 8-byte per slot, 4 rank byte per slot + 256 entries, 5 8-byte in global memory */
  uint32_t chains_per_wave; /* 2: the two-chains-per-wave kernel (13..15 bits with a one-chain-per-wave index) */
  uint32_t class_weights[8]; /* per-mille chain / run lengths of the 8 wave scheduling classes the launch was shaped with (fitted
                              * constants, HSRANS_*_WEIGHTS override them): which table a measurement used */
  uint32_t dynamic_groups;   /* block_/mt_ plans with checkpoints: blocks handed to workgroups by a ticket counter (1) or statically (0) */
  uint32_t spread;           /* such plans with few, large blocks: all chains dealt out evenly over the resident workgroups (1) */
} hsrans_launch_info;
int hsrans_dplan_launch_info(const hsrans_dplan *dplan, hsrans_launch_info *info);

/* The dealing of the one-round launch of block_/mt_ plans with checkpoints (k_decode_dealt), without a device (tests, planning): block k of the
 * plan = chains [block_begin[k], block_begin[k + 1]) (n_blocks + 1 entries, the last = n_chains); total_groups = the groups of S symbols the
 * plan decodes.  Every one of the launch's 512 workgroups gets chains [begin_out[b], begin_out[b + 1]) — sized by the workgroups' scheduling
 * classes (the MI355X defaults; ctx != NULL: that context's fitted ones), never reaching into a third block — and split_out[b] = how many of
 * them belong to the share's first block (0xFFFF: all).  Returns 1 when the plan suits the launch, 0 when it keeps the grouped / spread
 * launch (too few chains per wave, a block and more per workgroup slot, shares bent too far by the cuts), < 0 on bad arguments. */
int hsrans_dealt_shares(const hsrans_ctx *ctx, uint32_t bits /* the class lengths differ between the 8-byte-table loop (<= 11) and the rank-table loop (13, 14) */,
                        const uint32_t *block_begin, uint32_t n_blocks, uint32_t n_chains, uint64_t total_groups, uint32_t *begin_out /* [513] */,
                        uint16_t *split_out /* [512] */);

/* diagnostics: with HSRANS_DEBUG_STAMPS=1 in the environment every wavefront of a persistent / direct launch records
 * s_memtime stamps {entry, table built, stream ready, done, static share done} in 8 slots; copies up to capacity_u64 values to
 * `out`, returns the count */
size_t hsrans_debug_read_stamps(hsrans_dplan *dplan, uint64_t *out, size_t capacity_u64);

const char *hsrans_version(void);

#ifdef __cplusplus
}
#endif

#endif /* HSRANS_HIP_H */
