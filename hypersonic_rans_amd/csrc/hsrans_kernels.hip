// gfx950 (MI355X, CDNA4) kernels for the hypersonic-rANS 32-bit-state / 16-bit-word decode path.
//
// What is computed (reference: /root/reference/src, scalar form rANS32x64_16w.cpp:223-250 = block_codec64.h:182-213):
//   for every group of S symbols, for state j = 0..S-1 in order:
//       slot = x_j & (2^bits - 1);  sym = cumulInv[slot];  out[i + idx2idx[j]] = sym;
//       x_j  = (x_j >> bits) * freq[sym] + slot - cumul[sym];
//       if (x_j < 2^15) x_j = (x_j << 16) | *readHead++;
//
// How it is mapped to a wave64 (one wavefront = one chain of the decode plan, hsrans_plan.h):
//   * lane j owns state j.  All S table steps happen at once; the only cross-lane dependency is the read cursor:
//     `readHead++` in ascending j  ==  lane j reads word[cur + popcount(renorm_mask & lanes_below_j)], i.e.
//     v_cmp -> SGPR-pair mask, v_mbcnt_lo/hi for the lane's rank, s_bcnt1 to advance the (scalar) cursor.
//   * decode table lives in LDS.  bits <= 12: one uint32 per slot = sym | (freq-1) << 8 | (slot-cumul) << 20
//     (freq-1 so that freq == 4096 fits — the reference's own packed table cannot, hist.cpp:304).
//     bits >= 13: uint8 sym[2^bits] + uint32 {freq | cumul << 16}[256] (two dependent LDS reads, as the reference's
//     hist_dec2_t path, hist.h:42-47).  The table is built in-kernel from the 256 uint16 counts in the stream.
//   * the uint16 stream is staged through a 2 KiB LDS ring per wave, refilled 1 KiB at a time by one coalesced,
//     bounds-checked buffer_load_dwordx4 per lane that is issued ~25 groups before it is needed.
//   * idx2idx maps 4 consecutive lanes to 4 consecutive output bytes (it is the bit permutation
//     j -> (j&0x23)|((j&4)<<2)|((j&0x18)>>1)), so a quad assembles one output dword with two DPP quad_perm ORs;
//     four groups are accumulated so that every lane stores one dword and the wave writes 4*S contiguous bytes.
//   * no MFMA: this is integer gather work; the roofline that bounds it is HBM (compressed bytes in + decoded bytes out).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "hsrans_kernels.h"
#include "hsrans_plan.h"

namespace hsrans
{

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kRingSlots = 4;
constexpr uint32_t kChunkBytes = 512; // 32 lanes x 16 B
constexpr uint32_t kRingBytes = kRingSlots * kChunkBytes; // 2 KiB
// (the mirror is 128 bytes: the first 64 words of the ring, copied behind its end)
constexpr uint32_t kWaveRingBytes = kRingBytes + 256;     // per wave (ring + mirror, 256-byte granular)
// the hand-scheduled loop of k_decode_direct (run_groups_fast) keeps its read cursor as a plain LDS address that is only
// re-based every 4 groups, so it can run up to one chunk past the ring's end: the mirror there is a whole chunk
constexpr uint32_t kFastRingBytes = kRingBytes + kChunkBytes;
constexpr uint32_t kConsume = 1u << 15; // rans.h:8 DecodeConsumePoint16
// Per-wave time stamps (tools/stamps.py, tools/stamps_grouped.py, tools/tune_weights.py) exist only in the diagnostic build
// (`make stamps` -> lib/libhsrans_hip_stamps.so: -DHSRANS_HAVE_STAMPS=1 -DHSRANS_GROUP_STAMPS=1; the Python layer loads it when
// HSRANS_DEBUG_STAMPS=1).  Compiled in but switched off, their bookkeeping (five 64-bit time values kept across the decode loop)
// cost the shipped kernels 3-4 %: 39.9 -> 38.1 us for the replayed 100 MB decode, 61.9 -> 59.4 us at 15 bits, 8 % in run_grouped.
#ifndef HSRANS_PERSIST_STRICT // A/B builds: 0 = the exact wait in the uniform-interval launches
#define HSRANS_PERSIST_STRICT 1
#endif
#ifndef HSRANS_FORCE_STRICT // A/B builds: the constant wait at chunk crossings in every hand-scheduled single-chain loop
#define HSRANS_FORCE_STRICT 0
#endif
#ifndef HSRANS_HAVE_STAMPS
#define HSRANS_HAVE_STAMPS 0
#endif
#define HSRANS_STAMPS(kp) (HSRANS_HAVE_STAMPS && (kp).stamps != nullptr)
constexpr uint32_t kSingleMirror = 256;  // k_decode_single: ring entries mirrored behind the ring's end (4 groups x 64 words)

// decode-table layouts
constexpr int kModePack = 0;     // bits <= 11: uint32 per slot = sym | freq << 8 | (slot - cumul) << 20
constexpr int kModePackM1 = 1;   // bits == 12: same with freq - 1 (freq == 4096 must fit 12 bits)
constexpr int kModeTwoLevel = 2; // bits >= 13: uint8 sym[2^bits] + uint32 {freq | cumul << 16}[256]
constexpr int kModePack64 = 3;   // bits <= 14, table shared by a workgroup: uint2 per slot = {freq | sym << 24, slot - cumul}:
                                 // v_mad_u32_u24 takes freq (low 24 bits) and the bias operand as they are, v_perm takes byte 3

// bits >= 14 with a host-built table (persistent 64-state launches): uint8 rank[2^bits] — the slot's symbol as its RANK by
// frequency — followed by 256 x uint2 {freq | sym << 24, -cumul} ordered by rank.  A byte gather (the slot is the LDS address in
// k_decode_dual), then an 8-byte gather from a 2 KiB table in which the 32 most frequent symbols — nearly every lane of a group —
// sit in 32 different bank pairs; x' = freq * (x >> bits) + slot - cumul.  18 / 34 KiB at 14 / 15 bits instead of 128 / 256 KiB.
// (Round 2's layout for these widths was a coarse table of 4096 granules + a fine table for the granules that straddle a symbol
// boundary: 16.4 vector instructions and 11.2 LDS cycles per group against 12.3 and 13.3 here — 62.2 -> 56.7 us at 15 bits.)
constexpr int kModeRank = 4;
// The MODE 3 entries left in global memory ("spilled" table: L1/L2-resident, gathered with global_load_dwordx2): the
// comparison point BASELINE config 3 asks for next to the LDS-resident tables (HSRANS_TABLE_SPILL=1, host-built tables only)
constexpr int kModeSpill = 5;

__host__ __device__ constexpr uint32_t table_bytes_for(int mode, uint32_t bits)
{
  return mode == kModeSpill ? 0u
         : mode == kModeTwoLevel ? (1u << bits) + 1024
         : mode == kModePack64 ? 8u << bits
         : mode == kModeRank ? (1u << bits) + 2048u
                               : 4u << bits;
}

// Modes whose 64-state loop is hand-scheduled: their rings carry a whole-chunk mirror (kFastRingBytes per wave) ...
__host__ __device__ constexpr bool fast_ring_mode(int mode) { return mode == kModePack64 || mode == kModeRank; }
// ... and the one whose table sits at the START of the workgroup's LDS (address 0: the slot is the address of its rank byte)
__host__ __device__ constexpr bool table_first_mode(int mode) { return mode == kModeRank; }

__device__ __forceinline__ uint32_t lds_address(const void *p)
{
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)p;
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
  return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
}

// The decoded bytes are written once and never read again by this kernel, so the stores carry cache-policy bits.  Which ones is
// a measurement, made in ONE process on the same buffers with the variants' launches alternating (tools/ab_probe.py; between
// processes the same binary moves by +-3 us), sustained, rotated over 4 (stream, output) pairs / one pair replayed, on three boxes
// (profiles/r04_store_policy_ab.jsonl), us per 100 MB decode unless noted, "sc0 sc1" against "nt":
//     one chain per wave, 64 states, 11 bit (k_decode_direct, the headline)   38.7-41.0 against 43.0-44.0 rotated, 32.8-33.5 against 33.6-34.1 replayed
//     the same at 12 bits 39.1 / 42.2; 32-state pairs (k_decode_direct) 44.5 / 45.8-48.2
//     checkpoint every 32 groups (k_decode_persist) 42.5 / 40.9-41.2; 14 / 15 bits (k_decode_dual) 47.8-48.4 / 46.0; 2^30-byte mt_ stream (k_decode_grouped) 376-384 / 371-373
//     (plain stores 41.7 / 35.9, sc1 41.1 / 33.0, sc1 nt 44.0 / 33.1 for the headline; no stores at all, a diagnostic: 31.5 / 31.2)
// sc1 / sc0 sc1 write through and DROP the line from the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"); nt keeps it.
// So: the one-chain-per-wave launches write through, everything else keeps the streaming stores of rounds 1-3.  The stores are
// what a rotated launch loses its time to: per-wave clocks around the store instruction (diagnostic build, -DHSRANS_DIAG_STORE_TIME)
// show the slowest tenth of the waves blocked for 10 us at store issue, the median wave for 2.7 us.
// -DHSRANS_STORE_POLICY='" nt"' / -DHSRANS_STORE_POLICY_DIRECT='" nt"' build other combinations.
#ifndef HSRANS_STORE_POLICY
#define HSRANS_STORE_POLICY " nt"
#endif
#ifndef HSRANS_STORE_POLICY_DIRECT
#define HSRANS_STORE_POLICY_DIRECT " sc0 sc1"
#endif
#define HSRANS_STORE_U32(ptr, v) asm volatile("global_store_dword %0, %1, off" HSRANS_STORE_POLICY : : "v"(ptr), "v"(v) : "memory")
#define HSRANS_STORE_U32_SADDR(base, voff, v) asm volatile("global_store_dword %0, %1, %2" HSRANS_STORE_POLICY : : "v"(voff), "v"(v), "s"(base) : "memory")
// WT: the write-through policy of the one-chain-per-wave launches
template <bool WT>
__device__ __forceinline__ void store_u32(uint8_t *ptr, uint32_t v)
{
  if (WT)
    asm volatile("global_store_dword %0, %1, off" HSRANS_STORE_POLICY_DIRECT : : "v"(ptr), "v"(v) : "memory");
  else
    asm volatile("global_store_dword %0, %1, off" HSRANS_STORE_POLICY : : "v"(ptr), "v"(v) : "memory");
}
template <bool WT>
__device__ __forceinline__ void store_u32_saddr(uint8_t *base, uint32_t voff, uint32_t v)
{
  if (WT)
    asm volatile("global_store_dword %0, %1, %2" HSRANS_STORE_POLICY_DIRECT : : "v"(voff), "v"(v), "s"(base) : "memory");
  else
    asm volatile("global_store_dword %0, %1, %2" HSRANS_STORE_POLICY : : "v"(voff), "v"(v), "s"(base) : "memory");
}

// are the 512 bytes of a histogram at stream offset `off` there to be read?
#define HSRANS_HIST_IN_RANGE(c, off) ((off) >= (c).stream_lo && (off) <= (c).stream_len && (c).stream_len - (off) >= 512)

// idx2idx as arithmetic (rANS32x64_16w.cpp:210-216; the 32-state table rANS32x32_16w.cpp:203 is its first half)
__device__ __forceinline__ uint32_t lane_to_byte(uint32_t j) { return (j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1); }

struct WaveCtx
{
  const uint8_t *stream;
  uint64_t stream_len;
  uint64_t stream_lo; // first stream byte that exists behind `stream` (0 unless the caller holds only a window of the stream, hsrans_decode_device_window)
  uint8_t *out;
  uint64_t out_cap;
  uint32_t *status;
  uint32_t bits, S, lane;
  uint32_t v_mask, v_bits; // 2^bits - 1 and bits, each held in a VGPR: a VALU op with an SGPR operand issues at half rate
  uint8_t *rings;        // LDS, kWaveRingBytes: this wave's stream ring + mirror
  uint8_t *table;        // LDS
  uint8_t *table_b;      // LDS: the table lanes 32..63 use in the paired 32-state modes (== table unless the halves decode different blocks)
  const uint2 *gtable;   // kModeSpill: the table in global memory
  uint16_t *scratch_cnt; // LDS, 512 B each, only live during table builds: they alias a ring that has no request in
  uint16_t *scratch_cum; // flight (build_table is always called before the ring is begun)
};

// ---------------------------------------------------------------------------------------------------------------
// stream ring: 4 slots x 512 B per wave, filled by LDS-DMA (buffer_load_dwordx4 ... lds under EXEC = lanes 0..31:
// 32 lanes x 16 B land linearly at M0, no VGPR staging, hardware bounds check against the descriptor).  The first 128
// bytes of the ring are mirrored behind its end (a second, 8-lane request whenever slot 0 is filled), so the up to 64
// words one group reads never wrap: a lane's address is ring + (cursor mod ring) + 2 * rank, one v_lshl_add_u32.
//
// Invariant: whenever the cursor is in chunk c (256 words), chunks c .. c+2 have been requested and c, c+1 have landed.
// ring_advance() runs at least once per 256 consumed words (4 groups of 64), so the cursor crosses at most one chunk
// boundary between two calls and never needs more than chunks c, c+1 before the next call.  On entering chunk c it
// requests chunk c+2 into the slot of the dead chunk c-2 and then waits with vmcnt(2) for chunk c+1.  Why 2 is enough:
// vmcnt(N) waits until all but the N youngest vector-memory operations are done, in issue order.  Younger than chunk
// c+1's request (and than its mirror request, if it has one) are (a) the request for c+2 just issued and (b) at least
// one output store: chunk c+1 was requested at an earlier ring_advance(), the cursor has moved since, every decoded
// group is followed by its store before the next ring_advance() (run_groups_impl: 4 groups, store, advance; or group,
// store, ..., advance), so a store sits between the two requests.  More young operations only make the wait stricter,
// never weaker.  The loads are issued from asm, so the compiler never tracks them and never parks the decode loop on
// vmcnt(0).  (Requesting one chunk further ahead, which makes the bound independent of the stores, measured 7 % slower.)
// ---------------------------------------------------------------------------------------------------------------
// cache-policy bits of the stream requests (experiments: -DHSRANS_STREAM_LOAD_POLICY=1 nt, 2 sc1, 3 sc0 sc1).  Measured on the
// 100 MB headline decode: sc1 / sc0 sc1 change nothing; nt makes the requests bypass the Infinity Cache, i.e. even a replayed
// stream comes from HBM every time (58.6 us against 44.0 us) — the default (no bits) is right.
#if !defined(HSRANS_STREAM_LOAD_POLICY) || HSRANS_STREAM_LOAD_POLICY == 0
#define HSRANS_STREAM_LOAD_FLAGS ""
#elif HSRANS_STREAM_LOAD_POLICY == 1
#define HSRANS_STREAM_LOAD_FLAGS " nt"
#elif HSRANS_STREAM_LOAD_POLICY == 2
#define HSRANS_STREAM_LOAD_FLAGS " sc1"
#else
#define HSRANS_STREAM_LOAD_FLAGS " sc0 sc1"
#endif
struct StreamWin // the stream as the ring's requests see it
{
  u32x4 rs;      // buffer descriptor (SGPRs): base = stream + `base`, num_records = bytes up to `limit`
  uint64_t base; // absolute byte offset in the stream of descriptor offset 0 (16-byte aligned)
};

struct Ring
{
  uint32_t voff0; // descriptor offset of this chain's word index 0 (16-byte aligned)
  uint32_t k;     // chunk the cursor was in at the last ring_advance()
  uint32_t cur;   // next word to read, counted from voff0 (wave-uniform)
  uint32_t lds;   // LDS byte address of the ring (what M0 / ds_read take)
  uint32_t clog;  // log2 of the chunk size in bytes: 9 (32 lanes x 16 B; 64-state chains) or 8 (16 lanes; paired 32-state chains)
  uint32_t mirror_lanes; // EXEC mask of the mirror request that goes with slot 0: 0xFF (128 B: a group reads <= 64 words) or all 32 lanes (kFastRingBytes)
  // Exact waits (ring_advance_exact): `vm` counts the vector-memory instructions this wave has issued through this file's asm
  // (stream requests, the counted output stores); seqN = its value right after the request for chunk k+N.  Vector-memory
  // operations of a wave complete in issue order, so "chunk k+1 has landed" == at most (vm - seq1) operations outstanding.
  // Operations the compiler issues on its own are not counted: that only makes a wait stricter than needed, never weaker.
  uint32_t vm, seq1, seq2, seq3;
  // the hand-scheduled loop's own bookkeeping, kept across its calls on ONE chain (run_direct decodes a chain in segments): output
  // stores issued since the last / the last but one chunk crossing when the loop was left (run_groups_fast; zero at a chain's start)
  uint32_t st1, st2;
#if HSRANS_HAVE_STAMPS
  uint32_t diag_wait = 0, diag_store = 0; // diagnostic build: shader clocks spent in the crossing waits / issuing the output stores (run_groups_fast)
#endif
};

// clog = 9: 2 KiB ring + 128 B mirror (a group reads <= 64 words); clog = 8: 1 KiB ring + 64 B mirror (<= 32 words)
__device__ __forceinline__ void ring_bind(Ring &r, const uint8_t *lds_ring, uint32_t clog = 9, bool whole_chunk_mirror = false)
{
  r.lds = uni(lds_address(lds_ring));
  r.clog = clog;
  r.mirror_lanes = whole_chunk_mirror ? (clog == 9 ? 0xFFFFFFFFu : 0xFFFFu) : clog == 9 ? 0xFFu : 0xFu;
}
__device__ __forceinline__ uint32_t ring_bytes(const Ring &r) { return kRingSlots << r.clog; }

__device__ __forceinline__ void ring_request(const StreamWin &sw, const Ring &r, const WaveCtx &c, uint32_t chunk, bool with_mirror = true)
{
  const uint32_t voff = r.voff0 + (chunk << r.clog) + c.lane * 16;
  const uint32_t slot = chunk & (kRingSlots - 1);
  const uint32_t dst = uni(r.lds + (slot << r.clog));
  const uint32_t lanes = r.clog == 9 ? 0xFFFFFFFFu : 0xFFFFu; // 32 or 16 lanes x 16 B
  // (EXEC in one move: the 64-bit move zero-extends its 32-bit source, and these masks never reach the upper half)
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %0, %2, 0 offen" HSRANS_STREAM_LOAD_FLAGS " lds\n\ts_mov_b64 exec, -1"
               :
               : "v"(voff), "s"(dst), "s"(sw.rs), "s"((uint64_t)lanes)
               : "memory");
  if (slot == 0 && with_mirror) // wave-uniform: the ring's first 128 (64) bytes once more, behind its end (lanes 0..7 / 0..3)
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %0, %2, 0 offen" HSRANS_STREAM_LOAD_FLAGS " lds\n\ts_mov_b64 exec, -1"
                 :
                 : "v"(voff), "s"(uni(r.lds + ring_bytes(r))), "s"(sw.rs), "s"((uint64_t)r.mirror_lanes)
                 : "memory");
}

// chunk 0's mirror alone (ring_begin with `later`: the mirror is first read when the cursor nears the ring's end, three chunks on)
__device__ __forceinline__ void ring_request_mirror0(const StreamWin &sw, const Ring &r, const WaveCtx &c)
{
  const uint32_t voff = r.voff0 + c.lane * 16;
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %0, %2, 0 offen" HSRANS_STREAM_LOAD_FLAGS " lds\n\ts_mov_b64 exec, -1"
               :
               : "v"(voff), "s"(uni(r.lds + ring_bytes(r))), "s"(sw.rs), "s"((uint64_t)r.mirror_lanes)
               : "memory");
}

// `pos` = first stream byte the descriptor must reach, `limit` = first stream byte the chain(s) can NOT need (the next
// chain's cursor, or the stream length): requests past it are dropped by the range check instead of fetching a
// neighbour's words
__device__ __forceinline__ void win_open(StreamWin &sw, const WaveCtx &c, uint64_t pos, uint64_t limit)
{
  pos = uni64(pos);
  limit = uni64(limit);
  if (limit > c.stream_len)
    limit = c.stream_len;
  // (starting the requests on a 128-byte line instead was measured: no difference, warm or cold)
  const uint64_t a0 = pos & ~(uint64_t)15;
  // range in whole 16-byte lanes: a dwordx4 that straddles num_records is dropped as a whole, and a0 is 16-aligned
  // inside a 16-aligned allocation, so rounding up never leaves the page the last stream byte is on
  // (a window launch holds nothing below stream_lo: the host entry refuses plans that read there, and a descriptor that would
  // start below it is left empty, so every request through it is dropped)
  const uint64_t left = a0 < limit && a0 >= c.stream_lo ? (limit - a0 + 15) & ~(uint64_t)15 : 0;
  const uint64_t addr = (uint64_t)(uintptr_t)c.stream + a0;
  sw.rs.x = uni((uint32_t)addr);
  sw.rs.y = uni((uint32_t)(addr >> 32) & 0xFFFF); // stride 0
  sw.rs.z = uni((uint32_t)(left > 0xFFFFFFFFull ? 0xFFFFFFFFull : left));
  sw.rs.w = 0x00020000;
  sw.base = a0;
}

// How far ahead of the chunk the cursor is in the ring requests stream bytes: 2 (round 1) keeps one slot spare; 3 uses all four
// slots.  A request has (HSRANS_RING_AHEAD - 1) chunks of decoding (12.4 groups each = ~2 us at 8 waves per SIMD) to land.
// Measured after the loop-header wait was removed (ring_ready): 2 and 3 are within noise of each other, replayed or rotated
// (39.3 / 45.6 us at 3, 40.4 / 46.7 us at 2); before that fix neither mattered, because the loop drained the queue anyway.
#ifndef HSRANS_RING_AHEAD
#define HSRANS_RING_AHEAD 3
#endif
static_assert(HSRANS_RING_AHEAD == 2 || HSRANS_RING_AHEAD == 3, "the ring has 4 slots: the cursor's chunk + 2 or 3 requested ones");

// start streaming a chain whose first word is at absolute stream byte `pos` (>= sw.base, < sw.base + 4 GiB)
// (`issue` false: the requests of exactly this call were issued earlier — run_grouped asks for a round's first chunks before
// the round's table build — and only the ring's bookkeeping is set up)
// (`later` true: only chunks 0 and 1 are asked for now, the caller asks for the others with ring_begin_rest — the one-chain-per-
// wave launch, in which every wave of the device is in its prologue at once and a CU takes in about 11 bytes per clock: the
// bytes a wave needs before its first group come first)
__device__ __forceinline__ void ring_begin(const StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t pos, bool issue = true, bool later = false)
{
  pos = uni64(pos);
  const uint32_t rel = (uint32_t)(pos - sw.base);
  r.voff0 = rel & ~15u;
  r.cur = (rel - r.voff0) >> 1;
  r.k = 0;
  r.st1 = r.st2 = 0;
  // every lane is done with the ring's previous contents (its ds_reads returned before their results were used)
  r.vm = 0;
  if (issue)
  {
    ring_request(sw, r, c, 0, !later);
    ring_request(sw, r, c, 1);
  }
  r.vm += 3; // chunk 0, its mirror, chunk 1
  r.seq1 = r.vm;
  if (issue && !later)
    ring_request(sw, r, c, 2);
  r.seq2 = ++r.vm;
  if (HSRANS_RING_AHEAD == 3)
  {
    if (issue && !later)
      ring_request(sw, r, c, 3);
    r.vm++;
  }
  r.seq3 = r.vm;
}
__device__ __forceinline__ void ring_begin_rest(const StreamWin &sw, Ring &r, const WaveCtx &c)
{
  ring_request_mirror0(sw, r, c);
  ring_request(sw, r, c, 2);
  if (HSRANS_RING_AHEAD == 3)
    ring_request(sw, r, c, 3);
}

// chunks 0 and 1 (and the mirror) have landed: ring_begin issues {chunk 0, mirror, chunk 1, chunk 2 [, chunk 3]} and anything
// issued after it only makes this wait stricter.
// The chain's state register(s) pass through the wait as asm operands.  Reason: the states are fetched by an ordinary load and
// are first USED inside the decode loop; the compiler then places its "s_waitcnt vmcnt(0)" for that load at the loop header,
// where it runs on EVERY iteration and drains the whole vector-memory queue (the previous iteration's store, the stream
// requests in flight) — the loop never had more than one request outstanding.  With the register as an operand here the
// compiler's wait lands in front of this statement, once per chain.  (tests/test_kernel_resources.py checks the ISA for it.)
__device__ __forceinline__ void ring_ready()
{
  if (HSRANS_RING_AHEAD == 3)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
}
__device__ __forceinline__ void ring_ready(uint32_t &x)
{
  if (HSRANS_RING_AHEAD == 3)
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(x)::"memory");
  else
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(x)::"memory");
}
__device__ __forceinline__ void ring_ready(uint32_t &xa, uint32_t &xb)
{
  if (HSRANS_RING_AHEAD == 3)
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(xa), "+v"(xb)::"memory");
  else
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(xa), "+v"(xb)::"memory");
}

__device__ __forceinline__ void ring_init(StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t pos, uint32_t &x)
{
  win_open(sw, c, pos, c.stream_len);
  ring_begin(sw, r, c, pos);
  ring_ready(x);
}

// call at least once per 256 consumed words
__device__ __forceinline__ void ring_advance(const StreamWin &sw, Ring &r, const WaveCtx &c)
{
  if ((r.cur >> (r.clog - 1)) > r.k)
  {
    r.k++;
    ring_request(sw, r, c, r.k + HSRANS_RING_AHEAD);
    r.vm += ((r.k + HSRANS_RING_AHEAD) & (kRingSlots - 1)) == 0 ? 2 : 1;
    r.seq1 = r.seq2;
    r.seq2 = r.seq3;
    r.seq3 = r.vm;
    if (HSRANS_RING_AHEAD == 2)
      r.seq2 = r.vm;
    // chunk k+1 has landed: AHEAD 2: see the invariant above (the request just issued and a store are younger); AHEAD 3: the
    // requests for k+2 and k+3 are both younger than the one for k+1, whatever the stores do
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
}

// wait until at most n vector-memory operations are outstanding, n rounded DOWN to one of a few immediates (s_waitcnt takes no
// register operand; waiting for fewer outstanding operations than allowed is only stricter)
#ifndef HSRANS_WAIT_MAX
#define HSRANS_WAIT_MAX 8
#endif
__device__ __forceinline__ void wait_vm_at_most(uint32_t n)
{
  if (HSRANS_WAIT_MAX >= 24 && n >= 24)
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (HSRANS_WAIT_MAX >= 16 && n >= 16)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (HSRANS_WAIT_MAX >= 12 && n >= 12)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (HSRANS_WAIT_MAX >= 10 && n >= 10)
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 8)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 6)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n >= 4)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1)
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The same with the exact wait (for paths whose stores are counted in r.vm: run_groups_impl<FULL>): with vmcnt(2) the wave also
// waits for the request of chunk k+2 — issued one chunk ago — and for every store in between, which makes a third chunk in
// flight worthless; vmcnt(vm - seq1) waits for chunk k+1 and nothing younger.
__device__ __forceinline__ void ring_advance_exact(const StreamWin &sw, Ring &r, const WaveCtx &c)
{
  if ((r.cur >> (r.clog - 1)) > r.k)
  {
    r.k++;
    ring_request(sw, r, c, r.k + HSRANS_RING_AHEAD);
    r.vm += ((r.k + HSRANS_RING_AHEAD) & (kRingSlots - 1)) == 0 ? 2 : 1;
    r.seq1 = r.seq2;
    r.seq2 = r.seq3;
    r.seq3 = r.vm;
    if (HSRANS_RING_AHEAD == 2)
      r.seq2 = r.vm;
    wait_vm_at_most(r.vm - r.seq1);
  }
}

__device__ __forceinline__ uint64_t ring_pos(const StreamWin &sw, const Ring &r) { return sw.base + r.voff0 + (uint64_t)r.cur * 2; }

// ---------------------------------------------------------------------------------------------------------------
// decode table build (hist.cpp:291-306 make_dec_pack_hist, :356-384 inplace_make_hist_dec2, :308-324 the sum check)
// `tid`/`nthreads` = the threads that share this table (one wave, or the whole workgroup); SYNC() orders their LDS traffic.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool BLOCK_SYNC>
__device__ bool build_table(const WaveCtx &c, uint64_t hist_off, uint32_t tid, uint32_t nthreads)
{
  auto sync = [&]() {
    if (BLOCK_SYNC)
      __syncthreads();
    else
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  if (MODE == kModeSpill) // the spilled table only exists host-built; the launcher never pairs this mode with a plan that needs a build
  {
    if (tid == 0)
      atomicOr(c.status, kStatusBadHist);
    return false;
  }
  uint16_t *cnt = c.scratch_cnt; // [256]
  uint16_t *cum = c.scratch_cum; // [256] exclusive prefix sums
  const uint32_t total = 1u << c.bits;
  bool good = true;

  if (!BLOCK_SYNC)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // a stream request of the previous piece may still be landing in the scratch slot
  sync(); // scratch aliases a ring slot: everyone must be done with it
  const bool in_range = HSRANS_HIST_IN_RANGE(c, hist_off);
  for (uint32_t s = tid; s < 256; s += nthreads)
    cnt[s] = in_range ? *(const uint16_t *)(c.stream + hist_off + 2 * s) : (uint16_t)0;
  sync();
  if (tid < 64)
  {
    const uint32_t c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    const uint32_t mine = c0 + c1 + c2 + c3;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t up = __shfl_up(incl, d, 64);
      if (tid >= (uint32_t)d)
        incl += up;
    }
    const uint32_t excl = incl - mine;
    cum[4 * tid] = (uint16_t)excl;
    cum[4 * tid + 1] = (uint16_t)(excl + c0);
    cum[4 * tid + 2] = (uint16_t)(excl + c0 + c1);
    cum[4 * tid + 3] = (uint16_t)(excl + c0 + c1 + c2);
    // uint32 sum must be exactly 2^bits, as inplace_complete_hist (hist.cpp:310); the decoder then returns 0.
    // A workgroup-shared build only raises the status bit and decodes on with the bogus table (every index stays
    // masked, so that is memory-safe; the host discards the output); a single-wave build stops its chain.
    good = (uint32_t)__shfl(incl, 63, 64) == total;
    if (!good && tid == 0)
      atomicOr(c.status, kStatusBadHist);
  }
  sync();
  if (!BLOCK_SYNC && !good)
    return false;

  // slot -> symbol: the largest s with cum[s] <= slot (zero-count symbols share cum with their successor and lose
  // the tie; trailing zero-count symbols sit at cum == total and are never hit) == hist.cpp:343-351
  // (the 8-byte table written by runs like the byte tables below — one search per thread — measured the same: 0.3715 against 0.370)
  if (MODE == kModePack64)
  {
    uint2 *tab = (uint2 *)c.table;
    for (uint32_t slot = tid; slot < total; slot += nthreads)
    {
      uint32_t s = 0;
#pragma unroll
      for (uint32_t step = 128; step >= 1; step >>= 1)
        s += ((uint32_t)cum[s + step] <= slot) ? step : 0;
      tab[slot] = make_uint2((uint32_t)cnt[s] | (s << 24), slot - (uint32_t)cum[s]);
    }
  }
  else if (MODE != kModeTwoLevel && MODE != kModeRank)
  {
    uint32_t *tab = (uint32_t *)c.table;
    for (uint32_t slot = tid; slot < total; slot += nthreads)
    {
      uint32_t s = 0;
#pragma unroll
      for (uint32_t step = 128; step >= 1; step >>= 1)
        s += ((uint32_t)cum[s + step] <= slot) ? step : 0;
      tab[slot] = s | (((uint32_t)cnt[s] - (MODE == kModePackM1 ? 1 : 0)) << 8) | ((slot - (uint32_t)cum[s]) << 20);
    }
  }
  else
  {
    uint32_t *sym4 = (uint32_t *)c.table;                  // uint8 sym[total], written 4 slots per store
    uint32_t *symtab = (uint32_t *)(c.table + total);      // freq | cumul << 16
    // every thread a contiguous run of dwords: ONE search for its first slot, then the symbol only moves forward (a search per
    // slot — 8 dependent LDS reads each — made the build of a 15-bit table the longest part of a grouped launch's round)
    const uint32_t dwords = total / 4;
    const uint32_t per = (dwords + nthreads - 1) / nthreads;
    const uint32_t q0 = tid * per, q1 = q0 + per < dwords ? q0 + per : dwords;
    if (q0 < q1)
    {
      uint32_t s = 0;
#pragma unroll
      for (uint32_t step = 128; step >= 1; step >>= 1)
        s += ((uint32_t)cum[s + step] <= 4 * q0) ? step : 0;
      uint32_t next = s < 255 ? (uint32_t)cum[s + 1] : 0x10000u; // first slot of the next symbol (zero-count symbols share theirs and are stepped over)
      for (uint32_t q = q0; q < q1; q++)
      {
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++)
        {
          const uint32_t slot = 4 * q + b;
          while (next <= slot)
          {
            s++;
            next = s < 255 ? (uint32_t)cum[s + 1] : 0x10000u;
          }
          packed |= s << (8 * b);
        }
        sym4[q] = packed;
      }
    }
    // kModeRank built on the device (the grouped launches: a table per block): the byte is the symbol itself — ranking 256
    // counts per block would cost more than the 0.7 conflict cycles per group it saves — and the entries are the 8-byte ones
    if (MODE == kModeRank)
    {
      uint2 *ent = (uint2 *)(c.table + total);
      for (uint32_t s = tid; s < 256; s += nthreads)
        ent[s] = make_uint2((uint32_t)cnt[s] | (s << 24), 0u - (uint32_t)cum[s]);
    }
    else
      for (uint32_t s = tid; s < 256; s += nthreads)
        symtab[s] = (uint32_t)cnt[s] | ((uint32_t)cum[s] << 16);
  }
  sync();
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// one group of S symbols.  Returns the table word whose low byte is this lane's symbol.
// `act_mask` = lanes that take part (lane < S; inside the final partial group only lanes whose byte exists,
// rANS32x64_16w.cpp:256).  Lanes outside it run the arithmetic on junk: they never enter the renormalisation
// ballot, never store, and their state is never used again.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool FULL>
__device__ __forceinline__ uint32_t group_step(uint32_t &x, Ring &r, const WaveCtx &c, unsigned long long act_mask)
{
  const uint32_t mask = (1u << c.bits) - 1;
  const uint32_t slot = x & c.v_mask;
  const uint32_t q = x >> c.v_bits; // < 2^21: the 24-bit multiplier applies (x < 2^31, bits >= 10)
  uint32_t e, nx;
  if (MODE == kModePack64)
  {
    const uint2 e2 = ((const uint2 *)c.table)[slot];
    e = e2.x;                      // symbol in byte 3: the output v_perm selects it from there
    nx = __umul24(q, e2.x) + e2.y; // the 24-bit multiplier ignores the symbol in bits 24..31
  }
  else if (MODE == kModeSpill)
  {
    const uint2 e2 = c.gtable[slot]; // per-lane gather through L1 / L2
    e = e2.x;
    nx = __umul24(q, e2.x) + e2.y;
  }
  else if (MODE == kModeRank)
  {
    const uint32_t rank = c.table[slot];
    const uint2 e2 = ((const uint2 *)(c.table + mask + 1))[rank];
    e = e2.x;
    nx = __umul24(q, e2.x) + e2.y + slot;
  }
  else if (MODE == kModePack)
  {
    e = ((const uint32_t *)c.table)[slot]; // sym | freq << 8 | (slot - cumul) << 20, freq <= 2048
    nx = __umul24(q, (e >> 8) & 0xFFF) + (e >> 20);
  }
  else if (MODE == kModePackM1)
  {
    e = ((const uint32_t *)c.table)[slot]; // sym | (freq - 1) << 8 | (slot - cumul) << 20
    nx = __umul24(q, (e >> 8) & 0xFFF) + q + (e >> 20);
  }
  else
  {
    e = c.table[slot];
    const uint32_t fc = ((const uint32_t *)(c.table + mask + 1))[e]; // freq | cumul << 16
    nx = __umul24(q, fc & 0xFFFF) + slot - (fc >> 16);
  }
  const bool low = nx < kConsume;
  const unsigned long long m_all = __builtin_amdgcn_ballot_w64(low);
  const unsigned long long m = FULL ? m_all : (m_all & act_mask);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
  // this lane's word: ring + (cursor mod ring) + 2 * rank; the mirror behind the ring's end makes the wrap invisible
  uint32_t waddr;
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(waddr) : "v"(rank), "s"(r.lds + ((r.cur << 1) & (ring_bytes(r) - 1))));
  uint32_t w = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)waddr;
  // x = low ? (nx << 16 | w) : nx, as one VALU op under EXEC = renormalising lanes (EXEC is all ones here: every
  // caller is in wave-uniform control flow of a full 64-lane wave); the two EXEC writes go to the scalar unit
  x = nx;
  asm volatile("s_mov_b64 exec, %2\n\tv_lshl_or_b32 %0, %0, 16, %1\n\ts_mov_b64 exec, -1" : "+v"(x) : "v"(w), "s"(m_all));
  r.cur += (uint32_t)__popcll(m);
#if defined(HSRANS_EXTRA_SALU) // experiment: is the scalar unit (one instruction per cycle per CU, shared by the four SIMDs) a limiter?
  uint32_t scratch_s = uni(c.bits);
  for (int k = 0; k < 2 * HSRANS_EXTRA_SALU; k++)
    asm volatile("s_add_u32 %0, %0, 1" : "+s"(scratch_s));
#endif
  return e;
}

// 4x4 byte transpose inside every quad of lanes: in = this lane's symbols of 4 consecutive groups (byte t = group t);
// out = the 4 symbols of group (lane & 3) for the quad's 4 lanes = one aligned dword of the output row.
__device__ __forceinline__ uint32_t quad_transpose(uint32_t v, uint32_t sel_a, uint32_t sel_b)
{
  const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
  v = __builtin_amdgcn_perm(p1, v, sel_a);
  const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
  return __builtin_amdgcn_perm(p2, v, sel_b);
}

// per-lane constants of the output path
struct OutLanes
{
  uint32_t store_off, sel_a, sel_b;
};

__device__ __forceinline__ OutLanes out_lanes(uint32_t lane, uint32_t S)
{
  OutLanes ol;
  const uint32_t row = lane & 3, quad = lane >> 2;
  const uint32_t dcol = (quad & 8) | ((quad & 1) << 2) | ((quad & 6) >> 1); // dword column of this quad = lane_to_byte(lane) >> 2
  ol.store_off = row * S + dcol * 4;
#if defined(HSRANS_DIAG_LINEAR_STORES) // diagnostic build (wrong output order!): what would the launch cost if lane L stored dword L of the 256-byte row?
  ol.store_off = lane * 4;
#endif
  ol.sel_a = (lane & 1) ? 0x03070105u : 0x06020400u;
  ol.sel_b = (lane & 2) ? 0x03020706u : 0x05040100u;
  return ol;
}

// this lane's symbols of 4 consecutive groups (byte SYM_BYTE of each table word) -> the dword it stores
template <uint32_t SYM_BYTE>
__device__ __forceinline__ uint32_t pack4(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, const OutLanes &ol)
{
  const uint32_t lo = __builtin_amdgcn_perm(e1, e0, 0x0c0c0400u + SYM_BYTE * 0x0101u);
  const uint32_t hi = __builtin_amdgcn_perm(e3, e2, 0x0c0c0400u + SYM_BYTE * 0x0101u);
  return quad_transpose(__builtin_amdgcn_perm(hi, lo, 0x05040100u), ol.sel_a, ol.sel_b);
}

// decode `steps` whole groups starting at output offset `o` (block_codec64.h:173-217)
template <int MODE, bool FULL>
__device__ __forceinline__ void run_groups_impl(uint32_t &x, const StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t &o_ref, uint32_t steps)
{
  uint64_t o = uni64(o_ref); // wave-uniform by construction; pinned to SGPRs
  const uint32_t S = FULL ? 64 : c.S;
  const bool act = FULL || c.lane < S;
  const unsigned long long act_mask = FULL ? ~0ull : __builtin_amdgcn_ballot_w64(act);
  constexpr uint32_t kSymByte = (MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) ? 3 : 0; // where group_step's return value holds the symbol
  const OutLanes ol = out_lanes(c.lane, S);

  for (; steps >= 4; steps -= 4)
  {
    const uint32_t e0 = group_step<MODE, FULL>(x, r, c, act_mask);
    const uint32_t e1 = group_step<MODE, FULL>(x, r, c, act_mask);
    const uint32_t e2 = group_step<MODE, FULL>(x, r, c, act_mask);
    const uint32_t e3 = group_step<MODE, FULL>(x, r, c, act_mask);
    const uint32_t acc = pack4<kSymByte>(e0, e1, e2, e3, ol);
    uint8_t *row_base = c.out + o; // wave-uniform
    if (FULL) // scalar base + 32-bit lane offset: no 64-bit address arithmetic per store (the compiler's form adds one v_lshl_add_u64 per 4 groups)
    {
      HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)row_base), ol.store_off, acc); // (uni64: the asm needs the base in an SGPR pair whatever the compiler thinks of its uniformity)
      r.vm++;
    }
    else if (act)
      HSRANS_STORE_U32((uint32_t *)(row_base + ol.store_off), acc);
    o += 4 * S;
    if (FULL)
      ring_advance_exact(sw, r, c);
    else
      ring_advance(sw, r, c);
  }
  const uint32_t p = lane_to_byte(c.lane);
  for (; steps > 0; steps--)
  {
    const uint32_t e = group_step<MODE, FULL>(x, r, c, act_mask);
    if (act)
      c.out[o + p] = (uint8_t)(e >> (8 * kSymByte));
    o += S;
  }
  ring_advance(sw, r, c);
  o_ref = o;
}

// ---------------------------------------------------------------------------------------------------------------
// The headline loop, hand-scheduled (64 states, 8-byte table entries; k_decode_direct).  Measured on MI355X: the SCALAR unit is
// a limiter of this kernel — one scalar instruction per cycle per CU, shared by the four SIMDs: two more s_add per group cost
// +10 us per 100 MB (tools/build_variants.sh salu2 / salu4) — and the compiler's version of the group spends 11 scalar
// instructions (cursor arithmetic with wrap, two EXEC writes, loop control).  Here a group costs 3:
//   * the read cursor is a plain LDS byte address (s_bcnt1 + s_lshl1_add per group); it is re-based only when run_groups_fast
//     looks at it every 4 groups, which is why the ring's mirror is a whole chunk (kFastRingBytes);
//   * v_cmpx writes the renormalisation mask to VCC and to EXEC in one VALU instruction: rank, address, word read and merge then
//     run under EXEC = renormalising lanes (the word read touches only those lanes' banks) and ONE s_mov restores EXEC.
// Per group: 9 vector, 2 LDS, 3 scalar instructions (+ 2 s_waitcnt); the packing of the 4 symbols is inside the block as well.
// ---------------------------------------------------------------------------------------------------------------
#define HSRANS_FAST_GROUP(P0, P1)                                                                                                                    \
  "v_and_b32 %[t], %[x], %[vmask]\n\t"                                                                                                               \
  "v_lshl_add_u32 %[t], %[t], 3, %[stab]\n\t"                                                                                                        \
  "ds_read_b64 v[" #P0 ":" #P1 "], %[t]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[x], %[vbits], %[x]\n\t"                                                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_mad_u32_u24 %[x], v" #P0 ", %[x], v" #P1 "\n\t"                                                                                                 \
  "v_cmpx_gt_u32 vcc, %[lim], %[x]\n\t"                                                                                                              \
  "s_nop 1\n\t"                                                                                                                                      \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[t], %[t], 1, %[sa]\n\t"                                                                                                          \
  "ds_read_u16 %[t], %[t]\n\t"                                                                                                                       \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                   \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_or_b32 %[x], %[x], 16, %[t]\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"

// four groups from state x; returns the dword of this lane's four symbols (byte t = group t), before the quad transpose
__device__ __forceinline__ uint32_t fast_groups4(uint32_t &x, uint32_t &s_addr, const WaveCtx &c, uint32_t s_table)
{
  uint32_t acc, t, st;
  asm volatile(HSRANS_FAST_GROUP(52, 53) HSRANS_FAST_GROUP(54, 55) HSRANS_FAST_GROUP(56, 57) HSRANS_FAST_GROUP(58, 59)
               "v_perm_b32 %[acc], v54, v52, %[selp]\n\t"
               "v_perm_b32 %[t], v58, v56, %[selp]\n\t"
               "v_perm_b32 %[acc], %[t], %[acc], %[selq]"
               : [x] "+v"(x), [sa] "+s"(s_addr), [acc] "=&v"(acc), [t] "=&v"(t), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [stab] "s"(s_table), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "vcc", "scc", "memory");
  return acc;
}

// The same group for the rank table (kModeRank; one chain per wave): three dependent LDS reads — the rank byte at LDS address
// `slot` (the table starts at address 0), the symbol's 8-byte entry behind the bytes (%[sent] = 2^bits), the stream word.
// 10 vector, 3 LDS, 3 scalar instructions.
#define HSRANS_FAST_GROUP_RANK(P0, P1)                                                                                                               \
  "v_and_b32 %[g], %[x], %[vmask]\n\t"                                                                                                               \
  "ds_read_u8 v" #P0 ", %[g]\n\t"                                                                                                                    \
  "v_lshrrev_b32 %[x], %[vbits], %[x]\n\t"                                                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_add_u32 %[t], v" #P0 ", 3, %[sent]\n\t"                                                                                                    \
  "ds_read_b64 v[" #P0 ":" #P1 "], %[t]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_mad_u32_u24 %[x], v" #P0 ", %[x], v" #P1 "\n\t"                                                                                                 \
  "v_add_u32 %[x], %[x], %[g]\n\t"                                                                                                                   \
  "v_cmpx_gt_u32 vcc, %[lim], %[x]\n\t"                                                                                                              \
  "s_nop 1\n\t"                                                                                                                                      \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[t], %[t], 1, %[sa]\n\t"                                                                                                          \
  "ds_read_u16 %[t], %[t]\n\t"                                                                                                                       \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                   \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_or_b32 %[x], %[x], 16, %[t]\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"

__device__ __forceinline__ uint32_t fast_groups4_rank(uint32_t &x, uint32_t &s_addr, const WaveCtx &c, uint32_t s_entries)
{
  uint32_t acc, t, g, st;
  asm volatile(HSRANS_FAST_GROUP_RANK(52, 53) HSRANS_FAST_GROUP_RANK(54, 55) HSRANS_FAST_GROUP_RANK(56, 57) HSRANS_FAST_GROUP_RANK(58, 59)
               "v_perm_b32 %[acc], v54, v52, %[selp]\n\t"
               "v_perm_b32 %[t], v58, v56, %[selp]\n\t"
               "v_perm_b32 %[acc], %[t], %[acc], %[selq]"
               : [x] "+v"(x), [sa] "+s"(s_addr), [acc] "=&v"(acc), [t] "=&v"(t), [g] "=&v"(g), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [sent] "s"(s_entries), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "vcc", "scc", "memory");
  return acc;
}

// (Measured and dropped: the four groups' symbols as four byte stores — global_store_byte / _d16_hi, lane j writing byte idx2idx(j)
// of its group — instead of pack + quad transpose + one dword store: 1.25 vector instructions per group fewer, the replayed
// decode unchanged (38.2-38.9 us against 38.8-39.4), the rotated one 4-6 us slower: the loop is not bound by vector issue alone
// — the table gather keeps the LDS busy two thirds of the time — and four times as many store instructions crowd vmcnt.)
// `steps` whole groups (64 states, kModePack64) with the loop above; what is left over (< 4 groups) goes to the ordinary path
// The exact wait at a chunk crossing of the hand-scheduled loops, from the loop's iteration counter alone (it runs DOWN; one
// counted store per iteration, issued before the crossing is looked for).  The wave must know that chunk k + 1 has landed; its
// request was made at the crossing before the previous one, when the counter stood at t2.  Issued after it: the stores of the
// iterations since, the requests for k + 2 and k + 3 = `k3` (the latter just now) and their mirrors — and whatever else the wave
// issued (another ring's requests), which only makes "at most n outstanding" stricter than needed.  One asm statement with
// t1 / t2 tied to their registers: left to the compiler, the count became an induction variable of its own (a v_add and a
// v_readfirstlane per iteration) and the rotation of the marks put register moves on the path WITHOUT a crossing.
// n >= 3 always (a store precedes every crossing); 3 only at a chain's first crossings, whose requests ring_begin made in one go.
__device__ __forceinline__ void wait_after_crossing(uint32_t &t1, uint32_t &t2, uint32_t iters, uint32_t k3)
{
  static_assert(HSRANS_RING_AHEAD == 3 || HSRANS_RING_AHEAD == 2, "");
  static_assert(kRingSlots == 4, "");
  uint32_t n, extra; // extra = 2 requests + a mirror if k3 or k3 - 1 went to slot 0, i.e. slot(k3) < 2
  asm volatile("s_and_b32 %[extra], %[k3], 3\n\t"
               "s_cmp_lt_u32 %[extra], 2\n\t"
               "s_cselect_b32 %[extra], 3, 2\n\t"
               "s_sub_u32 %[n], %[t2], %[it]\n\t"
               "s_add_u32 %[n], %[n], %[extra]\n\t"
               "s_mov_b32 %[t2], %[t1]\n\t"
               "s_mov_b32 %[t1], %[it]\n\t"
               "s_cmp_ge_u32 %[n], 8\n\t"
               "s_cbranch_scc1 8f\n\t"
               "s_cmp_ge_u32 %[n], 6\n\t"
               "s_cbranch_scc1 6f\n\t"
               "s_cmp_ge_u32 %[n], 4\n\t"
               "s_cbranch_scc1 4f\n\t"
               "s_waitcnt vmcnt(3)\n\t"
               "s_branch 9f\n"
               "4:\n\t"
               "s_waitcnt vmcnt(4)\n\t"
               "s_branch 9f\n"
               "6:\n\t"
               "s_waitcnt vmcnt(6)\n\t"
               "s_branch 9f\n"
               "8:\n\t"
               "s_waitcnt vmcnt(8)\n"
               "9:"
               : [n] "=&s"(n), [extra] "=&s"(extra), [t1] "+s"(t1), [t2] "+s"(t2)
               : [it] "s"(iters), [k3] "s"(k3)
               : "scc", "memory");
}

template <bool STRICT, int MODE = kModePack64, bool WT = false>
__device__ __forceinline__ void run_groups_fast(uint32_t &x, const StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t &o_ref, uint32_t &steps)
{
  const OutLanes ol = out_lanes(c.lane, 64);
  const uint32_t s_table = uni(lds_address(c.table));
  // the cursor as an LDS address, and the address at which it enters the next chunk
  uint32_t s_addr = uni(r.lds + ((r.cur << 1) & (kRingBytes - 1)));
  uint32_t next_cross = uni(r.lds + (((r.k + 1) & (kRingSlots - 1)) << 9));
  if (next_cross == r.lds)
    next_cross += kRingBytes; // the chunk in slot 3 ends at the ring's end, not at its start
  uint32_t words0 = r.cur; // to rebuild r.cur afterwards: words consumed = (bytes the address moved) / 2
  uint32_t moved = 0;      // bytes the address has moved, including the re-basings
  const uint32_t s_addr0 = s_addr;
  // The loop's own bookkeeping is scalar work too, and the scalar unit is shared by the CU's four SIMDs (see above).  As the
  // compiler had it, an iteration WITHOUT a chunk crossing spent 15 scalar instructions outside the four groups (base + offset of
  // the output added up twice, a count of vector-memory instructions, register moves for the three request marks it rotates);
  // now 7: the output position is ONE pointer, the iteration counter is the only count, and everything about the waits happens
  // at the crossings.  At a crossing into chunk k the wave must know that chunk k + 1 has landed.  Issued after that chunk's
  // request: the requests for k + 2 and k + 3 (and their mirrors) and the stores of the iterations since.
  //   exact  (!STRICT): that number, from the iteration counter at the crossing before the previous one (t2);
  //   STRICT: "at most 4 outstanding" — there is at least one store between any two crossings (a crossing is looked for once per
  //           iteration, after the iteration's store), so 4 always implies it; stricter than exact by a store or two issued two
  //           chunks ago.
  // Measured (100 MB raw 11 bit / 2^30-byte mt_ stream in 256 KiB blocks, against the loop as it was): one pair replayed
  // 39.1 -> 37.8 us exact, 37.5 strict; the grouped launch 484-497 -> 494-497 us exact, 475-477 strict; a checkpoint every 32 groups
  // replayed 0.479 -> 0.499 strict, rotated 45.5 -> 44.8 us.  Four pairs rotated, one chain per wave — what the bench reports —
  // strict against exact, alternating runs: 45.9 / 44.4 us on one box (three runs each), 42.1 / 43.5 on another (six each): inside
  // the run-to-run spread (39-45 us).  Strict everywhere but in that launch, which keeps the exact wait.
  uint8_t *outp = (uint8_t *)uni64((uint64_t)(uintptr_t)(c.out + uni64(o_ref)));
  uint32_t iters = steps >> 2;
  steps &= 3;
  o_ref += (uint64_t)iters * 256;
  // iteration counts (they run down) at the last two crossings; at entry: where the previous call on this chain left off (r.st1 /
  // r.st2 stores ago), or, on a fresh chain, as if both had just happened.  Whatever else the wave issued in between is younger
  // than the requests these counts are about: stricter, never weaker.
  uint32_t t1 = iters + r.st1, t2 = iters + r.st2;
  // (The constant wait leans on the steady state: two crossings behind the current one, each with a store in front of it.  A
  // chain's first crossings have no such past — ring_begin asked for chunks 0..3 in one go — and wait for one operation more.)
  {
    for (; iters != 0; iters--)
    {
      const uint32_t acc = quad_transpose(MODE == kModeRank ? fast_groups4_rank(x, s_addr, c, 1u << c.bits) : fast_groups4(x, s_addr, c, s_table), ol.sel_a, ol.sel_b);
#if HSRANS_HAVE_STAMPS && defined(HSRANS_DIAG_STORE_TIME)
      const uint64_t ds0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if !defined(HSRANS_DIAG_NO_STORES)
      store_u32_saddr<WT>((uint8_t *)uni64((uint64_t)(uintptr_t)outp), ol.store_off, acc);
#else
      asm volatile("" ::"v"(acc)); // (diagnostic build: what does the launch cost without its output stores?  The waits below then count one operation too many: stricter)
#endif
#if HSRANS_HAVE_STAMPS && defined(HSRANS_DIAG_STORE_TIME)
      r.diag_store += (uint32_t)(__builtin_amdgcn_s_memtime() - ds0);
#endif
      outp += 256;
      if (s_addr >= next_cross) // entered the next chunk (at most one per 4 groups: they take <= 512 bytes)
      {
#if HSRANS_HAVE_STAMPS
        const uint64_t dwt0 = __builtin_amdgcn_s_memtime();
#endif
        r.k++;
        next_cross += kChunkBytes;
        if (s_addr >= r.lds + kRingBytes) // ... which was slot 0, read through the mirror so far: back to the ring proper
        {
          s_addr -= kRingBytes;
          next_cross -= kRingBytes;
          moved += kRingBytes;
        }
        ring_request(sw, r, c, r.k + HSRANS_RING_AHEAD);
        if (!STRICT && HSRANS_RING_AHEAD == 3)
          wait_after_crossing(t1, t2, iters, r.k + HSRANS_RING_AHEAD);
        else
        {
          if (HSRANS_RING_AHEAD == 3 && r.k <= 2)
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // (the chain's first crossings: behind the request in question only the next one, one store, this one)
          else if (HSRANS_RING_AHEAD == 3)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // (request k + 2 and one store)
        }
#if HSRANS_HAVE_STAMPS
        r.diag_wait += (uint32_t)(__builtin_amdgcn_s_memtime() - dwt0);
#endif
      }
    }
  }
  r.cur = words0 + ((s_addr + moved - s_addr0) >> 1);
  r.st1 = t1, r.st2 = t2; // (iters == 0 here: the counts are "stores since")
  r.vm = r.seq1 = r.seq2 = r.seq3 = 0; // (not kept here; zero only makes the waits of the few groups behind this loop stricter)
}

// FAST: the call sites that carry the bulk of a launch's groups (every inlined copy of the hand-scheduled loop pins v52-v59 and
// costs the big multi-path kernel registers: with it at every call site k_decode<3, true> went to 97 VGPRs and spilled)
template <int MODE, bool FAST = false, bool STRICT = false, bool WT = false> // STRICT: the constant wait of run_groups_fast (the grouped launches); WT: write-through stores
__device__ __forceinline__ void run_groups(uint32_t &x, const StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t &o, uint32_t steps)
{
  if (FAST && MODE == kModePack64 && c.S == 64 && r.mirror_lanes == 0xFFFFFFFFu)
    run_groups_fast<STRICT || HSRANS_FORCE_STRICT, kModePack64, WT>(x, sw, r, c, o, steps); // the hand-scheduled loop; leaves < 4 groups
  if (FAST && MODE == kModeRank && c.S == 64 && r.mirror_lanes == 0xFFFFFFFFu && uni(lds_address(c.table)) == 0)
    run_groups_fast<true, kModeRank, WT>(x, sw, r, c, o, steps); // (its rank byte's address is the slot itself: the table at LDS address 0)
  if (c.S == 64)
    run_groups_impl<MODE, true>(x, sw, r, c, o, steps);
  else
    run_groups_impl<MODE, false>(x, sw, r, c, o, steps);
}

// ---------------------------------------------------------------------------------------------------------------
// Paired 32-state chains (rANS32x32, persistent launches): lanes 0..31 decode chain A, lanes 32..63 chain B, each with
// its own ring (1 KiB + 64 B mirror) and cursor, so a wave64 is fully used.  One ballot serves both: its low half is
// A's renormalisation mask, its high half B's; v_mbcnt over the whole mask gives lanes >= 32 rank_B + popcount(A),
// and that popcount is folded into B's scalar cursor base.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ uint32_t group_step_pair(uint32_t &x, Ring &ra, Ring &rb, const WaveCtx &c)
{
  const uint32_t mask = (1u << c.bits) - 1;
  const uint32_t slot = x & c.v_mask;
  const uint32_t q = x >> c.v_bits;
  const uint8_t *tab = c.lane < 32 ? c.table : c.table_b; // per half (loop-invariant)
  uint32_t e, nx;
  if (MODE == kModePack64)
  {
    const uint2 e2 = ((const uint2 *)tab)[slot];
    e = e2.x;
    nx = __umul24(q, e2.x) + e2.y;
  }
  else if (MODE == kModeSpill)
  {
    const uint2 e2 = c.gtable[slot];
    e = e2.x;
    nx = __umul24(q, e2.x) + e2.y;
  }
  else if (MODE == kModeRank)
  {
    const uint2 e2 = ((const uint2 *)(tab + mask + 1))[tab[slot]];
    e = e2.x;
    nx = __umul24(q, e2.x) + e2.y + slot;
  }
  else if (MODE == kModePack)
  {
    e = ((const uint32_t *)tab)[slot];
    nx = __umul24(q, (e >> 8) & 0xFFF) + (e >> 20);
  }
  else if (MODE == kModePackM1)
  {
    e = ((const uint32_t *)tab)[slot];
    nx = __umul24(q, (e >> 8) & 0xFFF) + q + (e >> 20);
  }
  else
  {
    e = tab[slot];
    const uint32_t fc = ((const uint32_t *)(tab + mask + 1))[e];
    nx = __umul24(q, fc & 0xFFFF) + slot - (fc >> 16);
  }
  const bool low = nx < kConsume;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(low);
  const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0));
  const uint32_t cnt_a = (uint32_t)__popc(m_lo), cnt_b = (uint32_t)__popc(m_hi);
  // LDS address of this half's cursor (lanes >= 32 carry A's count in their rank, so it comes off B's base): the two
  // scalars are spread to their halves with one DPP move restricted to rows 2..3 (lanes 32..63) — no EXEC writes
  const uint32_t base_a = ra.lds + ((ra.cur << 1) & (ring_bytes(ra) - 1));
  const uint32_t base_b = rb.lds + ((rb.cur << 1) & (ring_bytes(rb) - 1)) - 2 * cnt_a;
  uint32_t va, vb;
  asm("v_mov_b32 %0, %1" : "=v"(va) : "s"(base_a));
  asm("v_mov_b32 %0, %1" : "=v"(vb) : "s"(base_b));
  const uint32_t vbase = (uint32_t)__builtin_amdgcn_update_dpp((int)va, (int)vb, 0xE4, 0xC, 0xF, false);
  uint32_t waddr;
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(waddr) : "v"(rank), "v"(vbase));
  uint32_t w = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)waddr;
  x = nx;
  asm volatile("s_mov_b64 exec, %2\n\tv_lshl_or_b32 %0, %0, 16, %1\n\ts_mov_b64 exec, -1" : "+v"(x) : "v"(w), "s"(m));
  ra.cur += cnt_a;
  rb.cur += cnt_b;
  return e;
}

// the cursor of one ring as an LDS address, for the loops that re-base it only every 4 groups (they need a whole chunk mirrored
// behind the ring's end: ring_bind(..., whole_chunk_mirror = true))
struct FastCursor
{
  uint32_t addr, next_cross, addr0, words0, moved;
};
__device__ __forceinline__ FastCursor fast_cursor_open(const Ring &r)
{
  FastCursor f;
  f.addr = uni(r.lds + ((r.cur << 1) & (ring_bytes(r) - 1)));
  f.next_cross = uni(r.lds + (((r.k + 1) & (kRingSlots - 1)) << r.clog));
  if (f.next_cross == r.lds)
    f.next_cross += ring_bytes(r); // the chunk in the last slot ends at the ring's end, not at its start
  f.addr0 = f.addr;
  f.words0 = r.cur;
  f.moved = 0;
  return f;
}
// the cursor has entered the next chunk (at most one per 4 groups): bookkeeping only, the caller requests and waits
__device__ __forceinline__ void fast_cursor_cross(FastCursor &f, Ring &r)
{
  r.k++;
  f.next_cross += 1u << r.clog;
  if (f.addr >= r.lds + ring_bytes(r)) // ... which was slot 0, read through the mirror so far: back to the ring proper
  {
    f.addr -= ring_bytes(r);
    f.next_cross -= ring_bytes(r);
    f.moved += ring_bytes(r);
  }
}
__device__ __forceinline__ void fast_cursor_close(const FastCursor &f, Ring &r) { r.cur = f.words0 + ((f.addr + f.moved - f.addr0) >> 1); }

// Two 32-state chains per wave: ring A and ring B of one wave (4 x 256 B each).  With the 8-byte table the pair loop is
// hand-scheduled and wants whole-chunk mirrors: 2 x (1 KiB + 256 B) = the 2.5 KiB every wave of a kModePack64 launch owns.
template <int MODE>
__device__ __forceinline__ void pair_bind(Ring &ra, Ring &rb, const WaveCtx &c)
{
  ring_bind(ra, c.rings, 8, fast_ring_mode(MODE));
  ring_bind(rb, c.rings + (fast_ring_mode(MODE) ? 1280 : 1152), 8, fast_ring_mode(MODE));
}

// One group of both chains (lanes 0..31 chain A, 32..63 chain B; 8-byte table entries), hand-scheduled like HSRANS_FAST_GROUP:
// v_cmpx puts the renormalisation mask of both halves in VCC and EXEC; chain A's ranks come from v_mbcnt_lo, chain B's from
// v_mbcnt_hi alone (issued under EXEC = mask & upper half, like B's word address), so each half counts from its own scalar
// cursor.  10 vector, 2 LDS, 7 scalar instructions (the compiler's version: 14 + 2 + 19); since round 3 both halves' word addresses
// are formed on all lanes and picked by a v_cndmask instead: 11 vector, 5 scalar (a checkpoint every 32 groups 0.388 -> 0.401,
// one chain per wave the same, rotated 51.0 -> 50.4 us: the CU's one scalar unit is this loop's contended resource).
#ifndef HSRANS_PAIR_SELECT // 1: both halves' word addresses formed on all lanes and picked with a v_cndmask (5 vector, 0 scalar); 0: chain B's under EXEC = upper half (4 vector, 2 scalar)
#define HSRANS_PAIR_SELECT 1
#endif
#if HSRANS_PAIR_SELECT
#define HSRANS_PAIR_ADDR                                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_mbcnt_hi_u32_b32 %[w], vcc_hi, 0\n\t"                                                                                                           \
  "v_lshl_add_u32 %[t], %[t], 1, %[sa]\n\t"                                                                                                          \
  "v_lshl_add_u32 %[w], %[w], 1, %[sb]\n\t"                                                                                                          \
  "v_cndmask_b32 %[w], %[t], %[w], %[up]\n\t"
#else
#define HSRANS_PAIR_ADDR                                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_lshl_add_u32 %[w], %[t], 1, %[sa]\n\t"                                                                                                          \
  "s_mov_b32 exec_lo, 0\n\t"                                                                                                                         \
  "v_mbcnt_hi_u32_b32 %[t], vcc_hi, 0\n\t"                                                                                                           \
  "v_lshl_add_u32 %[w], %[t], 1, %[sb]\n\t"                                                                                                          \
  "s_mov_b32 exec_lo, vcc_lo\n\t"
#endif
#define HSRANS_PAIR_GROUP(P0, P1)                                                                                                                    \
  "v_and_b32 %[t], %[x], %[vmask]\n\t"                                                                                                               \
  "v_lshl_add_u32 %[t], %[t], 3, %[stab]\n\t"                                                                                                        \
  "ds_read_b64 v[" #P0 ":" #P1 "], %[t]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[x], %[vbits], %[x]\n\t"                                                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_mad_u32_u24 %[x], v" #P0 ", %[x], v" #P1 "\n\t"                                                                                                 \
  "v_cmpx_gt_u32 vcc, %[lim], %[x]\n\t"                                                                                                              \
  "s_nop 1\n\t"                                                                                                                                      \
  HSRANS_PAIR_ADDR                                                                                                                                   \
  "ds_read_u16 %[w], %[w]\n\t"                                                                                                                       \
  "s_bcnt1_i32_b32 %[st], vcc_lo\n\t"                                                                                                                \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                          \
  "s_bcnt1_i32_b32 %[st], vcc_hi\n\t"                                                                                                                \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_or_b32 %[x], %[x], 16, %[w]\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"

__device__ __forceinline__ uint32_t pair_groups4(uint32_t &x, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_table)
{
  uint32_t acc, t, w, st;
  asm volatile(HSRANS_PAIR_GROUP(52, 53) HSRANS_PAIR_GROUP(54, 55) HSRANS_PAIR_GROUP(56, 57) HSRANS_PAIR_GROUP(58, 59)
               "v_perm_b32 %[acc], v54, v52, %[selp]\n\t"
               "v_perm_b32 %[t], v58, v56, %[selp]\n\t"
               "v_perm_b32 %[acc], %[t], %[acc], %[selq]"
               : [x] "+v"(x), [sa] "+s"(s_a), [sb] "+s"(s_b), [acc] "=&v"(acc), [t] "=&v"(t), [w] "=&v"(w), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [stab] "s"(s_table), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u),
                 [up] "s"(0xFFFFFFFF00000000ull)
               : "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "vcc", "scc", "memory");
  return acc;
}

// The pair group for the rank table (14 / 15 bits; the table at LDS address 0, both halves use the one table): rank byte, entry,
// stream word — 12 vector, 3 LDS, 5 scalar instructions
#define HSRANS_PAIR_GROUP_RANK(P0, P1)                                                                                                               \
  "v_and_b32 %[g], %[x], %[vmask]\n\t"                                                                                                               \
  "ds_read_u8 v" #P0 ", %[g]\n\t"                                                                                                                    \
  "v_lshrrev_b32 %[x], %[vbits], %[x]\n\t"                                                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_add_u32 %[t], v" #P0 ", 3, %[sent]\n\t"                                                                                                    \
  "ds_read_b64 v[" #P0 ":" #P1 "], %[t]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_mad_u32_u24 %[x], v" #P0 ", %[x], v" #P1 "\n\t"                                                                                                 \
  "v_add_u32 %[x], %[x], %[g]\n\t"                                                                                                                   \
  "v_cmpx_gt_u32 vcc, %[lim], %[x]\n\t"                                                                                                              \
  "s_nop 1\n\t"                                                                                                                                      \
  HSRANS_PAIR_ADDR                                                                                                                                   \
  "ds_read_u16 %[w], %[w]\n\t"                                                                                                                       \
  "s_bcnt1_i32_b32 %[st], vcc_lo\n\t"                                                                                                                \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                          \
  "s_bcnt1_i32_b32 %[st], vcc_hi\n\t"                                                                                                                \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_or_b32 %[x], %[x], 16, %[w]\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"

__device__ __forceinline__ uint32_t pair_groups4_rank(uint32_t &x, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_entries)
{
  uint32_t acc, t, w, g, st;
  asm volatile(HSRANS_PAIR_GROUP_RANK(52, 53) HSRANS_PAIR_GROUP_RANK(54, 55) HSRANS_PAIR_GROUP_RANK(56, 57) HSRANS_PAIR_GROUP_RANK(58, 59)
               "v_perm_b32 %[acc], v54, v52, %[selp]\n\t"
               "v_perm_b32 %[t], v58, v56, %[selp]\n\t"
               "v_perm_b32 %[acc], %[t], %[acc], %[selq]"
               : [x] "+v"(x), [sa] "+s"(s_a), [sb] "+s"(s_b), [acc] "=&v"(acc), [t] "=&v"(t), [w] "=&v"(w), [g] "=&v"(g), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [sent] "s"(s_entries), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u),
                 [up] "s"(0xFFFFFFFF00000000ull)
               : "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "vcc", "scc", "memory");
  return acc;
}

// `steps` whole 32-symbol groups of chain A (lanes 0..31, output at oa) and of chain B (lanes 32..63, output at ob)
template <int MODE, bool FAST = false, bool WT = false> // FAST, WT: see run_groups
__device__ __forceinline__ void run_pair_groups(uint32_t &x, const StreamWin &sw, Ring &ra, Ring &rb, const WaveCtx &c, uint64_t &oa_ref, uint64_t &ob_ref,
                                                uint32_t steps)
{
  const uint64_t oa = uni64(oa_ref), ob = uni64(ob_ref);
  constexpr uint32_t kSymByte = (MODE == kModePack64 || MODE == kModeSpill || MODE == kModeRank) ? 3 : 0;
  const uint32_t l32 = c.lane & 31, row = l32 & 3, quad = l32 >> 2;
  const uint32_t dcol = ((quad & 1) << 2) | ((quad & 6) >> 1);
  const uint32_t sel_a = (c.lane & 1) ? 0x03070105u : 0x06020400u;
  const uint32_t sel_b = (c.lane & 2) ? 0x03020706u : 0x05040100u;
  uint8_t *vout = c.out + (c.lane < 32 ? oa : ob) + row * 32 + dcol * 4; // per-lane: this half's output row
  uint32_t done = 0;
  if (FAST && ra.mirror_lanes == 0xFFFFu && (MODE == kModePack64 || (MODE == kModeRank && c.table_b == c.table && uni(lds_address(c.table)) == 0)))
  {
    // the hand-scheduled loop; its waits are counted from here on (everything issued before is older than anything it waits for)
    const uint32_t s_table = uni(lds_address(c.table));
    FastCursor fa = fast_cursor_open(ra), fb = fast_cursor_open(rb);
    // (the loop's bookkeeping as in run_groups_fast: the iteration counter is the only count, the waits are made up at the crossings)
    uint32_t iters = (steps - done) >> 2;
    done += iters * 4;
    uint32_t ta1 = iters, ta2 = iters, tb1 = iters, tb2 = iters;
    auto crossed = [&](FastCursor &f, Ring &r, uint32_t &t1, uint32_t &t2) {
      fast_cursor_cross(f, r);
      ring_request(sw, r, c, r.k + HSRANS_RING_AHEAD);
      // (exact here: the constant wait — at most 4 outstanding — gives 0.407 -> 0.417 replayed and takes 4 % rotated: 53.5 -> 55.5 us)
      if (HSRANS_RING_AHEAD == 3)
        wait_after_crossing(t1, t2, iters, r.k + HSRANS_RING_AHEAD); // (the other ring's requests are not counted: stricter, never weaker)
      else
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // (this ring's request for k + 2 and one store)
    };
    for (; iters != 0; iters--)
    {
      const uint32_t acc = quad_transpose(MODE == kModeRank ? pair_groups4_rank(x, fa.addr, fb.addr, c, 1u << c.bits) : pair_groups4(x, fa.addr, fb.addr, c, s_table), sel_a, sel_b);
      store_u32<WT>(vout, acc);
      vout += 128;
      if (fa.addr >= fa.next_cross)
        crossed(fa, ra, ta1, ta2);
      if (fb.addr >= fb.next_cross)
        crossed(fb, rb, tb1, tb2);
    }
    fast_cursor_close(fa, ra);
    fast_cursor_close(fb, rb);
    ra.vm = ra.seq1 = ra.seq2 = ra.seq3 = rb.vm = rb.seq1 = rb.seq2 = rb.seq3 = 0; // (stale otherwise; zero only makes later waits stricter)
  }
  for (; steps - done >= 4; done += 4)
  {
    const uint32_t e0 = group_step_pair<MODE>(x, ra, rb, c);
    const uint32_t e1 = group_step_pair<MODE>(x, ra, rb, c);
    const uint32_t lo = __builtin_amdgcn_perm(e1, e0, 0x0c0c0400u + kSymByte * 0x0101u);
    const uint32_t e2 = group_step_pair<MODE>(x, ra, rb, c);
    const uint32_t e3 = group_step_pair<MODE>(x, ra, rb, c);
    const uint32_t hi = __builtin_amdgcn_perm(e3, e2, 0x0c0c0400u + kSymByte * 0x0101u);
    HSRANS_STORE_U32((uint32_t *)vout, quad_transpose(__builtin_amdgcn_perm(hi, lo, 0x05040100u), sel_a, sel_b));
    vout += 128;
    ring_advance(sw, ra, c);
    ring_advance(sw, rb, c);
  }
  oa_ref = oa + (uint64_t)done * 32;
  ob_ref = ob + (uint64_t)done * 32;
}

// final partial group (rANS32x64_16w.cpp:252-280): only lanes whose output byte exists take part, in lane order
template <int MODE>
__device__ __forceinline__ void run_tail(uint32_t &x, Ring &r, const WaveCtx &c, uint64_t o, uint32_t tail)
{
  if (tail == 0)
    return;
  const uint32_t p = lane_to_byte(c.lane);
  const bool act = c.lane < c.S && p < tail;
  const uint32_t e = group_step<MODE, false>(x, r, c, __builtin_amdgcn_ballot_w64(act));
  if (act)
    c.out[o + p] = (uint8_t)(e >> ((MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) ? 24 : 0));
}

// single-symbol block (block_rANS32x64_16w_decode.cpp:52-60): wave-wide fill
__device__ void wave_fill(const WaveCtx &c, uint64_t o, uint64_t len, uint32_t symbol)
{
  uint8_t *p = c.out + o;
  uint64_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head > len)
    head = len;
  if (c.lane < head)
    p[c.lane] = (uint8_t)symbol;
  p += head;
  len -= head;
  const uint32_t s4 = symbol * 0x01010101u;
  const u32x4 v = {s4, s4, s4, s4};
  const uint64_t vecs = len / 16;
  for (uint64_t i = c.lane; i < vecs; i += 64)
    ((u32x4 *)p)[i] = v;
  const uint64_t done = vecs * 16;
  if (c.lane < len - done)
    p[done + c.lane] = (uint8_t)symbol;
}

// ---------------------------------------------------------------------------------------------------------------
// chain runners
// ---------------------------------------------------------------------------------------------------------------
struct PlanView
{
  const PlanHeader *hdr;
  const uint32_t *chain_first;
  const Piece *pieces;
  const uint32_t *states;
};

__device__ __forceinline__ PlanView plan_view(const uint8_t *plan)
{
  PlanView v;
  v.hdr = (const PlanHeader *)plan;
  v.chain_first = (const uint32_t *)(plan + plan_chain_first_off());
  v.pieces = (const Piece *)(plan + plan_pieces_off(v.hdr->n_chains));
  v.states = (const uint32_t *)(plan + plan_states_off(v.hdr->n_chains, v.hdr->n_pieces));
  return v;
}

// planned chain: pieces [first, last) with absolute offsets.  SHARED: the table was built by the workgroup already.
template <int MODE, bool SHARED>
__device__ void run_planned_chain(const WaveCtx &c, const PlanView &pv, uint32_t chain, const KParams &kp)
{
  const uint32_t first = uni(pv.chain_first[chain]);
  const uint32_t last = uni(pv.chain_first[chain + 1]);
  uint32_t x = 0;
  uint64_t have_hist = ~(uint64_t)0;
  StreamWin sw;
  Ring r;
  ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
  for (uint32_t pi = first; pi < last; pi++)
  {
    const Piece *pc = pv.pieces + pi;
    const uint32_t flags = uni(pc->flags);
    if (flags & kPieceChainStart)
      x = c.lane < c.S ? pv.states[(uint64_t)uni(pc->state_idx) * c.S + c.lane] : 0;
    if (flags & kPieceFill)
    {
      wave_fill(c, uni64(pc->out_off), uni64(pc->fill_len), (uint32_t)uni64(pc->hist_off) & 0xFF);
      continue;
    }
    const uint64_t hist_off = uni64(pc->hist_off);
    if (!SHARED && hist_off != have_hist)
    {
      if (!build_table<MODE, false>(c, hist_off, c.lane, 64))
        return;
      have_hist = hist_off;
    }
    ring_init(sw, r, c, uni64(pc->words_off), x);
    uint64_t o = uni64(pc->out_off);
    uint32_t steps = uni(pc->steps);

    if (kp.ckpt_groups != nullptr)
    {
      // index-build pass with explicit checkpoints (hsrans_index_build_at): `ckpt_groups` is an ascending list of absolute
      // group indices; boundary k that falls strictly inside this piece gets {states, cursor} recorded in slot k
      uint64_t g_abs = o / c.S;
      uint32_t lo = 0, hi = kp.n_ckpt_groups; // first boundary > g_abs
      while (lo < hi)
      {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (uni64(kp.ckpt_groups[mid]) <= g_abs)
          lo = mid + 1;
        else
          hi = mid;
      }
      uint32_t bi = lo;
      while (steps > 0)
      {
        const uint64_t next = bi < kp.n_ckpt_groups ? uni64(kp.ckpt_groups[bi]) : ~(uint64_t)0;
        const uint32_t n = next - g_abs < steps ? (uint32_t)(next - g_abs) : steps;
        run_groups<MODE>(x, sw, r, c, o, n);
        steps -= n;
        g_abs += n;
        if (steps > 0)
        {
          if (c.lane < c.S)
            kp.ckpt_states[(uint64_t)bi * c.S + c.lane] = x;
          if (c.lane == 0)
            kp.ckpt_words[bi] = ring_pos(sw, r);
          bi++;
        }
      }
    }
    else if (kp.ckpt_interval != 0)
    {
      // index-build pass (hsrans_index_build): record {states, cursor} at every `ckpt_interval`-th group boundary of the
      // piece.  Slot = absolute group index / interval: checkpoints of one piece are an interval apart and pieces do
      // not overlap in the output, so slots are unique across all chains of a stream.
      uint32_t g = 0;
      const uint64_t g_abs0 = o / c.S;
      while (steps > 0)
      {
        if (g != 0)
        {
          const uint64_t slot = (g_abs0 + g) / kp.ckpt_interval;
          if (c.lane < c.S)
            kp.ckpt_states[slot * c.S + c.lane] = x;
          if (c.lane == 0)
            kp.ckpt_words[slot] = ring_pos(sw, r);
        }
        const uint32_t n = steps < kp.ckpt_interval ? steps : kp.ckpt_interval;
        run_groups<MODE>(x, sw, r, c, o, n);
        steps -= n;
        g += n;
      }
    }
    else
      run_groups<MODE>(x, sw, r, c, o, steps);
    run_tail<MODE>(x, r, c, o, uni(pc->tail));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent launch (kPlanMergeable plans): the grid is sized to the machine (2 workgroups per CU).  Every wave first
// decodes `static_per_wave` consecutive chains as ONE chain (stream requests and state load in flight while the
// workgroup gets its table), then pulls single chains from one of kDynQueues atomic heads until the stream is done.
// Why dynamic: the SIMD arbiter favours its oldest wave, so equal static shares finish 2x apart (measured 23..52 us)
// and the tail runs at one-wave latency; the queues keep every SIMD full until the end.
// (Keeping two chains in flight per wave was tried and is not faster: the loop is bound by VALU/LDS throughput, not by
// the latency of the dependent LDS round trips.)
// ---------------------------------------------------------------------------------------------------------------
struct RunGeom
{
  uint64_t o;
  uint32_t steps, tail;
};

// geometry of the run of chains [c0, c1) and the start of its stream / state loads
template <int MODE>
__device__ __forceinline__ RunGeom run_begin(const WaveCtx &c, const PersistentArgs &pa, StreamWin &sw, uint32_t c0, uint32_t c1, uint32_t &x, Ring &r)
{
  const uint64_t words = uni64(pa.pieces[c0].words_off);
  win_open(sw, c, words, c1 < pa.n_chains ? uni64(pa.pieces[c1].words_off) : c.stream_len);
  ring_begin(sw, r, c, words);
  x = c.lane < c.S ? pa.states[(uint64_t)c0 * c.S + c.lane] : 0;
  RunGeom g;
  const uint64_t g0 = (uint64_t)c0 * pa.interval;
  const uint64_t g1 = (uint64_t)c1 * pa.interval < pa.steps_total ? (uint64_t)c1 * pa.interval : pa.steps_total;
  g.o = pa.out_base + g0 * c.S;
  g.steps = (uint32_t)(g1 - g0);
  g.tail = c1 == pa.n_chains ? pa.tail : 0;
  return g;
}

template <int MODE>
__device__ void run_persistent(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w)
{
  const PersistentArgs &pa = kp.pa;
  const uint32_t W = gridDim.x * waves;
  const uint64_t t_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t t_table = 0, t_ready = 0;

  uint32_t x = 0;
  StreamWin sw;
  Ring r;
  ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
  RunGeom g{};
  // static run of this wave: run_len[class] chains (host guarantees static_total <= n_chains)
  const uint32_t wave_in_wg = w % waves, blk = w / waves;
  const uint32_t first_half = (gridDim.x + 1) / 2;
  const uint32_t half = blk >= first_half ? 1 : 0;
  const uint32_t per_class = waves >= 4 ? waves / 4 : 1; // waves of one class in a workgroup
  const uint32_t cls = half * 4 + wave_in_wg / per_class;
  const uint32_t q0 = pa.run_len[cls];
  const uint32_t c_first = pa.half_base[half] + (blk - half * first_half) * pa.wg_chains[half] + pa.class_off[cls] + (wave_in_wg % per_class) * q0;
  const bool host_table = (MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) && pa.table != nullptr; // kModeRank / kModeSpill are host-built only
  if (!host_table) // the in-kernel build borrows ring space: it has to come before the first stream request
    build_table<MODE, true>(c, pa.hist_off, threadIdx.x, blockDim.x);
  if (q0 != 0)
    g = run_begin<MODE>(c, pa, sw, c_first, c_first + q0, x, r);
  if (host_table)
  {
    // the table was built on the host from the plan's histogram copy: one coalesced 16 B load + LDS store per thread,
    // while the first wave checks that the stream really carries that histogram (else: status, as a failed sum check)
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8; // (0 for the spilled table: it stays in global memory)
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
      bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo; // (a window launch may lack the histogram: nothing to compare)
      if (same && pa.hist_off >= c.stream_lo)
      {
        const uint64_t mine = *(const uint64_t *)(pa.hist_copy + 4 * c.lane);
        uint64_t theirs = 0;
        for (int b = 3; b >= 0; b--) // stream offsets are only 2-byte aligned
          theirs = (theirs << 16) | *(const uint16_t *)(c.stream + pa.hist_off + 8 * c.lane + 2 * b);
        same = mine == theirs;
      }
      if (__builtin_amdgcn_ballot_w64(!same) != 0 && c.lane == 0)
        atomicOr(c.status, kStatusBadHist);
    }
    __syncthreads();
  }
  if (HSRANS_STAMPS(kp))
    t_table = __builtin_amdgcn_s_memrealtime();
  if (q0 != 0)
  {
    ring_ready(x);
    if (HSRANS_STAMPS(kp))
      t_ready = __builtin_amdgcn_s_memrealtime();
    run_groups<MODE, true, HSRANS_PERSIST_STRICT>(x, sw, r, c, g.o, g.steps); // (strict wait: 0.479 -> 0.499 replayed with a checkpoint every 32 groups)
    run_tail<MODE>(x, r, c, g.o, g.tail);
  }
  const uint64_t t_static = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;

  // Dynamic part: queue k hands out chains [lo, hi) in order.  Its 64-bit head is never reset: every launch draws
  // exactly H = (hi - lo) + (waves on this queue) tickets from it (each wave fails exactly once), and launches on one
  // plan are serialised by their stream, so ticket mod H is this launch's ticket.  No memset node, no exit protocol.
  const uint32_t dyn0 = pa.static_total;
  const uint64_t D = pa.n_chains - dyn0;
  const uint32_t nq = W < kDynQueues ? W : kDynQueues; // queues in use: every one of them needs a wave
  const uint32_t k = w % nq;
  const uint32_t lo = dyn0 + (uint32_t)(k * D / nq), hi = dyn0 + (uint32_t)((k + 1) * D / nq);
  const uint64_t H = (uint64_t)(hi - lo) + (W - k + nq - 1) / nq;
  while (true)
  {
    unsigned long long t = 0;
    if (c.lane == 0)
      t = atomicAdd(pa.counters + k * kDynQueueStride, 1ull);
    t = uni64(t) % H;
    if (t >= hi - lo)
      break;
    const uint32_t ch = lo + (uint32_t)t;
    g = run_begin<MODE>(c, pa, sw, ch, ch + 1, x, r);
    ring_ready(x);
    run_groups<MODE, true, HSRANS_PERSIST_STRICT>(x, sw, r, c, g.o, g.steps); // (strict wait: 0.479 -> 0.499 replayed with a checkpoint every 32 groups)
    run_tail<MODE>(x, r, c, g.o, g.tail);
  }

  if (HSRANS_STAMPS(kp) && c.lane == 0)
  {
    uint64_t *st = kp.stamps + (uint64_t)w * 8;
    st[0] = t_entry;
    st[1] = t_table;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = t_static;
  }
}

// Direct launch (kPlanMergeable plans with chains of any length, PlanHeader::interval == 0: hsrans_index_boundaries /
// hsrans_plan_thin): the plan's chains are dealt to the launch's W waves as W RUNS of R = ceil(n_chains / W) consecutive chains.
// Chains of such a plan are back to back in stream and output, so a wave decodes its run as ONE chain from the first one's start
// states; everything it needs is in two Piece records, fetched with scalar loads; no queues, no atomics, nothing per launch on the
// device, so launches of one plan may overlap freely.  R == 1 — hsrans_index_boundaries makes exactly one chain per resident
// wavefront, sized by the wave's scheduling class — is the headline's shape; an index made for a smaller launch gives R > 1.
//
// (Round 4 built late-phase rebalancing on top of this and took it out again — branch r4-tail-stealing-experiment,
// profiles/r04_tail_stealing_ab.jsonl.  The last fifth of every wave's share was cut into 1-3 "tail" chains with a claim word
// each; the owner walked into them seamlessly, asking for the claim word 16 groups ahead by LDS-DMA; waves that were done sampled
// 128 claim words (requested before their own last groups) and took unclaimed tail chains with an atomic exchange.  Every variant
// was slower than none, 41-50 us against 40-42 rotated: a steal costs the exchange, the chain's own prologue (two dependent round
// trips) and its decode by a lone wave, 4-7 us in all, while the launch's tail is 3-7 us — and "unclaimed" does not tell a late
// owner from one that is on time, because the young wave classes run slowly first and fast at the end.  What the experiment left
// behind: the loop's crossing bookkeeping survives across calls (Ring::st1 / st2), and its finding about where a rotated launch
// loses its time — the stores, not the stream — is why the stores of this launch write through now.)
typedef const __attribute__((address_space(4))) uint64_t *kptr64; // constant address space: s_load through the scalar cache
typedef const __attribute__((address_space(4))) uint32_t *kptr32;

struct DirectPiece
{
  uint64_t words, out, limit;
  uint32_t steps, tail;
};

__device__ __forceinline__ DirectPiece direct_piece(const WaveCtx &c, const PersistentArgs &pa, uint32_t ch)
{
  const kptr64 p = (kptr64)(uintptr_t)(pa.pieces + ch);
  DirectPiece d;
  d.words = p[0];
  d.out = p[1];
  const uint32_t st = ((kptr32)p)[8]; // steps
  const uint32_t tf = ((kptr32)p)[9]; // tail | flags << 16
  d.steps = st;
  d.tail = tf & 0xFFFFu;
  d.limit = ch + 1 < pa.n_chains ? p[6] : c.stream_len; // the next piece's words_off (Piece is 48 bytes)
  return d;
}

template <int MODE>
__device__ void run_direct(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w)
{
  const PersistentArgs &pa = kp.pa;
  const uint32_t W = gridDim.x * waves;
  const uint64_t t_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  const uint64_t c_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memtime() : 0; // shader clock (diagnostics: what does the chip run at under this load?)
  uint64_t t_table = 0, t_ready = 0, t_static = 0;
#if HSRANS_HAVE_STAMPS
  uint32_t diag_wait = 0, diag_store = 0;
#endif
  const bool host_table = (MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) && pa.table != nullptr;
  if (kp.finish != nullptr && w == 0 && c.lane == 0) // calibration launches: the launch's time zero
    kp.finish[W] = __builtin_amdgcn_s_memrealtime();
  if (!host_table) // the in-kernel build borrows ring space: it has to come before the first stream request
    build_table<MODE, true>(c, pa.hist_off, threadIdx.x, blockDim.x);
  // the host-built table: one coalesced 16 B load + LDS store per thread (while the wave's first stream chunks and its
  // states are in flight); the first wave also checks that the stream really carries the histogram the table was built
  // from (else: status, as a failed sum check)
  // (requesting the table BEFORE the piece record, so that its fetch overlaps that round trip, was measured twice: no gain)
  auto fetch_table = [&]() {
    if (MODE != kModeSpill)
    {
      const uint32_t entries = table_bytes_for(MODE, c.bits) / 8;
      for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
        *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
      bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo; // (a window launch may lack the histogram: nothing to compare)
      if (same && pa.hist_off >= c.stream_lo)
      {
        const uint64_t mine = *(const uint64_t *)(pa.hist_copy + 4 * c.lane);
        uint64_t theirs = 0;
        for (int b = 3; b >= 0; b--) // stream offsets are only 2-byte aligned
          theirs = (theirs << 16) | *(const uint16_t *)(c.stream + pa.hist_off + 8 * c.lane + 2 * b);
        same = mine == theirs;
      }
      if (__builtin_amdgcn_ballot_w64(!same) != 0 && c.lane == 0)
        atomicOr(c.status, kStatusBadHist);
    }
    if (MODE != kModeSpill)
      __syncthreads();
    if (HSRANS_STAMPS(kp))
      t_table = __builtin_amdgcn_s_memrealtime();
  };
  const uint32_t n = pa.n_chains;
  const uint32_t R = pa.run_chains ? pa.run_chains : 1; // launch_decode: W * R >= n
  const uint32_t ch = w * R;                            // this wave's run: chains [ch, end)
  const uint32_t end = ch + R < n ? ch + R : n;
  if (ch < n)
  {
    StreamWin sw;
    Ring r;
    ring_bind(r, c.rings, 9, fast_ring_mode(MODE) && c.S == 64);
    uint32_t x = c.lane < c.S ? pa.states[(uint64_t)ch * c.S + c.lane] : 0; // address known up front: in flight beside the piece records
    const DirectPiece d = direct_piece(c, pa, ch);
    // the run's last record (the first one again when R == 1): where the run's words and its output end, the stream's final partial group
    const kptr64 plast = (kptr64)(uintptr_t)(pa.pieces + (end - 1));
    const uint64_t run_limit = end < n ? plast[6] : c.stream_len;
    const uint32_t last_steps = ((kptr32)plast)[8], last_tail = ((kptr32)plast)[9] & 0xFFFFu;
    const uint64_t run_end_out = plast[1] + (uint64_t)last_steps * c.S;
    win_open(sw, c, d.words, run_limit);
    // Every wave of the device is in its prologue at the same time, and a CU takes in about 11 bytes per clock then
    // (MI355X_MICROARCH.md, "prologue HBM burst"): what the first ~25 groups read — states, chunks 0 and 1 — is asked for first,
    // then the table; chunk 0's mirror and the chunks the ring keeps ahead come after that.  Rotated 39.3 -> 38.7 us, replayed
    // 32.65 -> 32.4 (profiles/r04_prologue_ab.jsonl; -DHSRANS_PROLOGUE_SPLIT=0: all five requests up front, as in rounds 1-3).
#if !defined(HSRANS_PROLOGUE_SPLIT) || HSRANS_PROLOGUE_SPLIT
    ring_begin(sw, r, c, d.words, true, true);
    if (host_table)
      fetch_table();
    ring_begin_rest(sw, r, c);
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(x)::"memory"); // (the three requests just made are the only younger ones: chunks 0, 1 and the states have landed)
#else
    ring_begin(sw, r, c, d.words);
    if (host_table)
      fetch_table();
    ring_ready(x);
#endif
    if (HSRANS_STAMPS(kp))
      t_ready = __builtin_amdgcn_s_memrealtime();
    uint64_t o = d.out;
    run_groups<MODE, true, false, true>(x, sw, r, c, o, (uint32_t)((run_end_out - o) / c.S));
    run_tail<MODE>(x, r, c, o, end == n ? last_tail : 0);
#if HSRANS_HAVE_STAMPS
    diag_wait += r.diag_wait, diag_store += r.diag_store;
#endif
  }
  else if (host_table) // a wave without a chain still takes part in the workgroup's table copy
    fetch_table();
  if (HSRANS_STAMPS(kp))
    t_static = __builtin_amdgcn_s_memrealtime();
  if (kp.finish != nullptr && c.lane == 0) // calibration launches (hsrans_ctx_calibrate): when this wave was done
    kp.finish[w] = __builtin_amdgcn_s_memrealtime();
  if (HSRANS_STAMPS(kp) && c.lane == 0)
  {
    uint64_t *st = kp.stamps + (uint64_t)w * 8;
    st[0] = t_entry;
    st[1] = t_table;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = t_static;
    st[5] = __builtin_amdgcn_s_memtime() - c_entry;
    // where the wave really ran (HW_ID: wave/SIMD/CU/SH/SE fields; XCC_ID): tools/stamps.py groups the finish times by it
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
    st[6] = (uint64_t)hw_id | ((uint64_t)xcc_id << 32);
#if HSRANS_HAVE_STAMPS
    st[7] = (uint64_t)diag_wait | ((uint64_t)diag_store << 32); // shader clocks waiting at chunk crossings | issuing stores (-DHSRANS_DIAG_STORE_TIME)
#endif
  }
}

// Direct launch of a 32-state plan: wave w decodes chains 2w and 2w + 1 side by side (lanes 0..31 / 32..63, group_step_pair);
// what the pair loop leaves (unequal lengths, < 4 groups, the stream's final partial group) is finished one chain at a time.
template <int MODE>
__device__ void run_direct_pair(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w)
{
  const PersistentArgs &pa = kp.pa;
  const uint32_t W = gridDim.x * waves;
  const uint64_t t_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t t_ready = 0, t_static = 0;
  StreamWin sw;
  Ring ra, rb;
  pair_bind<MODE>(ra, rb, c);
  const bool host_table = (MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) && pa.table != nullptr;
  if (!host_table)
    build_table<MODE, true>(c, pa.hist_off, threadIdx.x, blockDim.x);
  bool table_pending = host_table && MODE != kModeSpill;
  if (host_table && blockIdx.x == 0 && threadIdx.x < 64)
  {
    bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo; // (a window launch may lack the histogram: nothing to compare)
    if (same && pa.hist_off >= c.stream_lo)
    {
      const uint64_t mine = *(const uint64_t *)(pa.hist_copy + 4 * c.lane);
      uint64_t theirs = 0;
      for (int b = 3; b >= 0; b--)
        theirs = (theirs << 16) | *(const uint16_t *)(c.stream + pa.hist_off + 8 * c.lane + 2 * b);
      same = mine == theirs;
    }
    if (__builtin_amdgcn_ballot_w64(!same) != 0 && c.lane == 0)
      atomicOr(c.status, kStatusBadHist);
  }
  auto copy_table = [&]() {
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8;
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    __syncthreads();
  };
  // pair (2w, 2w + 1); a plan with more chain pairs than waves: wave w goes on with chains 2 (w + W), ...  One loop, one call site of the decode body
  uint32_t a = 2 * w, end = pa.n_chains; // chains [a, min(a + 2, end)) are this round's
  bool have = a < pa.n_chains;
  while (true)
  {
    if (have)
    {
      const bool have_b = a + 1 < end;
      const DirectPiece da = direct_piece(c, pa, a);
      const DirectPiece db = have_b ? direct_piece(c, pa, a + 1) : da;
      win_open(sw, c, da.words, have_b ? db.limit : da.limit);
      ring_begin(sw, ra, c, da.words);
      if (have_b)
        ring_begin(sw, rb, c, db.words);
      uint32_t x = pa.states[(uint64_t)((c.lane < 32 || !have_b) ? a : a + 1) * 32 + (c.lane & 31)];
      if (table_pending)
      {
        copy_table();
        table_pending = false;
      }
      ring_ready(x);
      if (HSRANS_STAMPS(kp) && t_ready == 0)
        t_ready = __builtin_amdgcn_s_memrealtime();
      uint64_t oa = da.out, ob = db.out;
      uint32_t sa = da.steps, sb = have_b ? db.steps : 0;
      const uint32_t both = have_b ? (sa < sb ? sa : sb) & ~3u : 0;
      run_pair_groups<MODE, true, true>(x, sw, ra, rb, c, oa, ob, both);
      sa -= both;
      sb -= both;
      uint32_t xb = __shfl(x, (c.lane & 31) + 32, 64); // B's states move down to lanes 0..31 and B is finished alone
      run_groups<MODE>(xb, sw, rb, c, ob, sb);
      run_tail<MODE>(xb, rb, c, ob, have_b ? db.tail : 0);
      run_groups<MODE>(x, sw, ra, c, oa, sa);
      run_tail<MODE>(x, ra, c, oa, da.tail);
    }
    if (HSRANS_STAMPS(kp) && t_static == 0)
      t_static = __builtin_amdgcn_s_memrealtime();
    a += 2 * W;
    have = a < pa.n_chains;
    if (!have)
      break;
  }
  if (table_pending) // a wave without chains still takes part in the workgroup's table copy
    copy_table();
  if (HSRANS_STAMPS(kp) && c.lane == 0) // (tools/stamps.py, tools/tune_weights.py --states 32)
  {
    uint64_t *st = kp.stamps + (uint64_t)w * 8;
    st[0] = t_entry;
    st[1] = t_ready;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = t_static;
    st[5] = 0;
    st[6] = 0;
  }
}

// Persistent launch for 32-state streams: every wave runs TWO runs of chains side by side (group_step_pair), A = run 2u,
// B = run 2u+1 of a numbering in which run j of the static part is chains [j*q0, (j+1)*q0) and a dynamic ticket t of
// queue k is the pair of adjacent chains lo+2t, lo+2t+1.  Whatever the pair loop leaves (unequal lengths, < 4 groups,
// the stream's final partial group) is finished one chain at a time on lanes 0..31.
template <int MODE, bool FAST = false> // FAST: the hand-scheduled pair loop (k_decode_persist only: it spills k_decode<3, true>)
__device__ void run_persistent_pair(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w)
{
  const PersistentArgs &pa = kp.pa;
  const uint32_t W = gridDim.x * waves;
  StreamWin sw;
  Ring ra, rb;
  pair_bind<MODE>(ra, rb, c);
  // static runs of this wave: two of run_len[class] chains each (PersistentArgs::run_len; host guarantees static_total <= n_chains)
  const uint32_t wave_in_wg = w % waves, blk = w / waves;
  const uint32_t first_half = (gridDim.x + 1) / 2;
  const uint32_t half = blk >= first_half ? 1 : 0;
  const uint32_t per_class = waves >= 4 ? waves / 4 : 1;
  const uint32_t cls = half * 4 + wave_in_wg / per_class;
  const uint32_t q0 = pa.run_len[cls];
  const uint32_t c_first = pa.half_base[half] + (blk - half * first_half) * pa.wg_chains[half] + pa.class_off[cls] + (wave_in_wg % per_class) * 2 * q0;
  const bool host_table = (MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) && pa.table != nullptr;
  if (!host_table)
    build_table<MODE, true>(c, pa.hist_off, threadIdx.x, blockDim.x);

  // decode chains [a0, a1) on lanes 0..31 and [a1, b1) on lanes 32..63 (b1 == a1: only A)
  auto run = [&](uint32_t a0, uint32_t a1, uint32_t b1, bool table_pending) {
    const bool have_b = b1 > a1;
    win_open(sw, c, uni64(pa.pieces[a0].words_off), b1 < pa.n_chains ? uni64(pa.pieces[b1].words_off) : c.stream_len);
    ring_begin(sw, ra, c, uni64(pa.pieces[a0].words_off));
    if (have_b)
      ring_begin(sw, rb, c, uni64(pa.pieces[a1].words_off));
    // lanes 0..31: state j of chain a0; lanes 32..63: state j of chain a1
    const uint32_t src_chain = (c.lane < 32 || !have_b) ? a0 : a1;
    uint32_t x = pa.states[(uint64_t)src_chain * 32 + (c.lane & 31)];
    auto geom = [&](uint32_t c0, uint32_t c1, uint64_t &o, uint32_t &steps, uint32_t &tail) {
      const uint64_t g0 = (uint64_t)c0 * pa.interval;
      const uint64_t g1 = (uint64_t)c1 * pa.interval < pa.steps_total ? (uint64_t)c1 * pa.interval : pa.steps_total;
      o = pa.out_base + g0 * 32;
      steps = (uint32_t)(g1 - g0);
      tail = c1 == pa.n_chains ? pa.tail : 0;
    };
    uint64_t oa, ob = 0;
    uint32_t sa, sb = 0, ta, tb = 0;
    geom(a0, a1, oa, sa, ta);
    if (have_b)
      geom(a1, b1, ob, sb, tb);
    if (table_pending)
    {
      // the table was built on the host from the plan's histogram copy: one coalesced 16 B load + LDS store per thread
      // (see run_persistent for the check of the copy against the stream)
      const uint32_t entries = table_bytes_for(MODE, c.bits) / 8; // (0 for the spilled table)
      for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
        *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
      __syncthreads();
    }
    ring_ready(x);
    if (have_b)
    {
      const uint32_t both = (sa < sb ? sa : sb) & ~3u;
      run_pair_groups<MODE, FAST>(x, sw, ra, rb, c, oa, ob, both);
      sa -= both;
      sb -= both;
      // chain B's states move down to lanes 0..31 and B is finished alone
      uint32_t xb = __shfl(x, (c.lane & 31) + 32, 64);
      run_groups<MODE>(xb, sw, rb, c, ob, sb);
      run_tail<MODE>(xb, rb, c, ob, tb);
    }
    run_groups<MODE>(x, sw, ra, c, oa, sa);
    run_tail<MODE>(x, ra, c, oa, ta);
  };

  if (host_table && blockIdx.x == 0 && threadIdx.x < 64)
  {
    bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo; // (a window launch may lack the histogram: nothing to compare)
    if (same && pa.hist_off >= c.stream_lo)
    {
      const uint64_t mine = *(const uint64_t *)(pa.hist_copy + 4 * c.lane);
      uint64_t theirs = 0;
      for (int b = 3; b >= 0; b--)
        theirs = (theirs << 16) | *(const uint16_t *)(c.stream + pa.hist_off + 8 * c.lane + 2 * b);
      same = mine == theirs;
    }
    if (__builtin_amdgcn_ballot_w64(!same) != 0 && c.lane == 0)
      atomicOr(c.status, kStatusBadHist);
  }
  if (q0 != 0)
    run(c_first, c_first + q0, c_first + 2 * q0, host_table);
  else if (host_table)
  {
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8; // (0 for the spilled table)
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    __syncthreads();
  }

  const uint32_t dyn0 = pa.static_total;
  const uint64_t D = pa.n_chains - dyn0;
  const uint32_t nq = W < kDynQueues ? W : kDynQueues; // queues in use: every one of them needs a wave
  const uint32_t k = w % nq;
  const uint32_t lo = dyn0 + (uint32_t)(k * D / nq), hi = dyn0 + (uint32_t)((k + 1) * D / nq);
  const uint32_t pairs = (hi - lo + 1) / 2;
  const uint64_t H = (uint64_t)pairs + (W - k + nq - 1) / nq; // tickets per launch, see run_persistent
  while (true)
  {
    unsigned long long t = 0;
    if (c.lane == 0)
      t = atomicAdd(pa.counters + k * kDynQueueStride, 1ull);
    t = uni64(t) % H;
    if (t >= pairs)
      break;
    const uint32_t a0 = lo + 2 * (uint32_t)t;
    run(a0, a0 + 1, a0 + 2 <= hi ? a0 + 2 : a0 + 1, false);
  }
}

// Grouped launch: workgroup b walks groups b, b + gridDim.x, ...; per group one table build, then every wave decodes an
// equal contiguous share of the group's chains.
// LEAN: the host promises a 64-state plan whose groups are all mergeable runs or fills (every block_/mt_ stream with checkpoints
// this library's encoders or index builders make): the 32-state pair path and the general chain runner are left out of the
// kernel, which is what keeps it at 8 waves per SIMD
template <int MODE, bool LEAN = false, bool FAST = false> // FAST: the hand-scheduled 32-state pair loop too (measured in a kernel of its own: 71 VGPRs, 7 waves per SIMD — not used)
__device__ void run_grouped(const WaveCtx &c, const PlanView &pv, const KParams &kp, uint32_t waves, uint32_t wave)
{
  // -DHSRANS_GROUP_STAMPS=1 builds (tools/stamps_grouped.py; needs HSRANS_DEBUG_STAMPS=1 at run time): where a wave's time goes,
  // summed over its rounds: [0] first entry, [1] waiting at the round's barrier, [2] table build, [3] plan records + first chunks
  // (until the decode loop starts), [4] decode, [5] rounds, [6] last exit.  Compile-time because even switched off the extra
  // bookkeeping cost this kernel 8 % (0.405 -> 0.44 of 8 TB/s on the 1 GiB mt_ workload without it).
#if defined(HSRANS_GROUP_STAMPS) && HSRANS_GROUP_STAMPS
#define HSRANS_GS(...) __VA_ARGS__
#else
#define HSRANS_GS(...)
#endif
  HSRANS_GS(uint64_t t_first = 0, acc_wait = 0, acc_build = 0, acc_meta = 0, acc_dec = 0, rounds = 0; if (HSRANS_STAMPS(kp)) t_first = __builtin_amdgcn_s_memrealtime();)
  // (Measured and not kept, round 3: wave 0 pulling the NEXT group's record, piece records and start states through the caches at the
  // end of its round — LDS-DMA into the build scratch, so that the three dependent round trips behind the barrier hit L2: 64 KiB
  // blocks 537 / 542 / 534 us against 537 / 540 / 542 with it, 100 MB 0.368 against 0.369.  The other workgroups of the CU hide them.)
  // (Round 2 had measured a ticket counter — drawn by everyone after the round's barrier — and checkpoints placed by wave class
  // inside the blocks, and found neither worth it; what changed the picture in round 3 is below: the draw hidden in wave 0's
  // barrier wait, four 8-wave workgroups per CU (launch_shape) and the younger waves' raised priority.)
  // Which group next.  Static: b, b + gridDim.x, ...  Dynamic (kp.group_tickets): round 0 is static, every later group comes from
  // a ticket counter.  The draw is made by WAVE 0 alone, at the end of its share of the round, and waited for on the spot: wave 0
  // is the oldest wave of the workgroup, the SIMDs serve it first, it finishes first and would spend the round trip (and several
  // microseconds more) at the round's barrier anyway — so the draw costs the workgroup nothing, the decision is made as late as
  // it can be, and no register carries a ticket across the decode loop.  (Drawn after the barrier by everyone it cost ~2 us per
  // round and made the launch slower than the static order it was meant to beat.)  The group goes through one of two LDS words,
  // alternating by round: round r's word is written before barrier r and read behind it; the next write to it comes behind
  // barrier r + 1.  Every workgroup draws once at the end of every round it runs, the launch as a whole exactly n_groups times:
  // ticket mod n_groups is the launch-local order whatever the counter has seen before (it is never reset).
  const bool dynamic = kp.group_tickets != nullptr && kp.n_groups > gridDim.x;
  volatile uint32_t *lds_next = (volatile uint32_t *)(c.table + table_bytes_for(MODE, c.bits)); // 2 words: launch_shape reserves 64 bytes behind the table
  uint32_t gi = blockIdx.x;
  for (uint32_t round = 0;; round++)
  {
    if (!(dynamic && round >= 1) && gi >= kp.n_groups) // (dynamic rounds: decided below, from the published group)
      break;
    // `advance` runs at the end of the round (every path of the loop body ends in it)
    auto advance = [&]() {
      if (!dynamic)
        gi += gridDim.x;
      else if (wave == 0 && c.lane == 0)
      {
        const uint32_t j = (uint32_t)(atomicAdd(kp.group_tickets, 1ull) % kp.n_groups);
        lds_next[(round + 1) & 1] = j < kp.n_groups - gridDim.x ? gridDim.x + j : 0xFFFFFFFFu;
      }
    };
    HSRANS_GS(const uint64_t t0 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no stream request of the previous group may still land in the scratch slot
    __syncthreads();                                    // every wave is done with the previous group's table and rings
    if (dynamic && round >= 1)
    {
      gi = uni(lds_next[round & 1]);
      if (gi >= kp.n_groups) // (the same in every wave)
        break;
    }
    const Group *G = kp.groups + gi;
    const uint32_t begin = uni(G->begin), count = uni(G->count), flags = uni(G->flags);
    // mergeable groups: chain `begin + i` is piece `piece0 + i` and its start states are states[begin + i] (the host checks this
    // when it marks a group mergeable), so a wave's records come straight from the group record: one level of loads, not three
    const uint32_t piece0 = uni(G->piece0);
    HSRANS_GS(const uint64_t t1 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
    // kp.group_overlap (64-state mergeable groups): the wave's piece records, start states and first stream chunks are requested
    // BEFORE the table build and land while it runs (the build's scratch then has an LDS area of its own: launch_shape) — the
    // ~4.5 us of dependent round trips a round used to spend after the build overlap its ~3 us instead
    const bool overlap = LEAN && kp.group_overlap != 0 && (flags & kGroupMergeable); // (the general instantiation has no registers to spare for it)
#if defined(HSRANS_GROUP_STAMPS) && HSRANS_GROUP_STAMPS
    const uint64_t t2 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
    acc_wait += t1 - t0;
    acc_build += t2 - t1;
    rounds++;
    auto stamp_out = [&]() {
      if (HSRANS_STAMPS(kp) && c.lane == 0)
      {
        uint64_t *st = kp.stamps + (uint64_t)(blockIdx.x * waves + wave) * 8;
        st[0] = t_first;
        st[1] = acc_wait;
        st[2] = acc_build;
        st[3] = acc_meta;
        st[4] = acc_dec;
        st[5] = rounds;
        st[6] = __builtin_amdgcn_s_memrealtime();
      }
    };
#endif
    const uint32_t half = blockIdx.x >= (gridDim.x + 1) / 2 ? 1 : 0;
    // age-class weights only where a wave gets enough chains for them to mean something (else an even split)
    const bool weighted = count >= 8 * waves;
    const uint32_t cum_all = weighted ? kp.group_cum[half][waves] : waves;
    const uint32_t first = begin + (uint32_t)((uint64_t)(weighted ? kp.group_cum[half][wave] : wave) * count / cum_all);
    const uint32_t last = begin + (uint32_t)((uint64_t)(weighted ? kp.group_cum[half][wave + 1] : wave + 1) * count / cum_all);
    // the wave's run of a mergeable 64-state group: chains [first, last) as one chain.  With `early` it is opened twice: before
    // the table build for the sake of its requests (start states, first stream chunks: in flight during the build), and again
    // after it without them — the records come from the caches then — so that only the state register lives across the build
    // (the window, the ring and the run's geometry are 20 scalar registers the builder has no room for: they spilled).
    StreamWin sw;
    Ring r;
    uint32_t x = 0, run_tail_syms = 0;
    uint64_t o = 0, run_steps = 0;
    auto open_run = [&](bool issue) {
      const Piece *p0 = pv.pieces + (piece0 + (first - begin));
      const Piece *p1 = pv.pieces + (piece0 + (last - 1 - begin));
      const uint64_t limit = last < begin + count ? uni64(pv.pieces[piece0 + (last - begin)].words_off) : uni64(G->words_end);
      if (issue)
        x = c.lane < c.S ? pv.states[(uint64_t)first * c.S + c.lane] : 0;
      ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
      win_open(sw, c, uni64(p0->words_off), limit);
      ring_begin(sw, r, c, uni64(p0->words_off), issue);
      o = uni64(p0->out_off);
      run_steps = (uni64(p1->out_off) - o) / c.S + uni(p1->steps);
      run_tail_syms = uni(p1->tail);
    };
    const bool early = overlap && first < last;
    if (early)
      open_run(true);
    if (!(flags & kGroupFill)) // (one call site: every inlined copy of the builder costs the kernel registers)
      build_table<MODE, true>(c, uni64(G->hist_off), threadIdx.x, blockDim.x);
    if (first >= last)
    {
      advance();
      continue;
    }
    if (!LEAN && (flags & kGroupMergeable) && c.S == 32)
    {
      // 32-state chains: the wave's share is cut in two runs that are decoded side by side, A on lanes 0..31 and B on
      // lanes 32..63 (group_step_pair), like run_persistent_pair; whatever the pair loop leaves is finished one run at a time
      const uint32_t mid = first + (last - first + 1) / 2;
      const bool have_b = mid < last;
      const Piece *a0 = pv.pieces + (piece0 + (first - begin));
      const Piece *a1 = pv.pieces + (piece0 + (mid - 1 - begin));
      const Piece *b0 = pv.pieces + (piece0 + ((have_b ? mid : first) - begin));
      const Piece *b1 = pv.pieces + (piece0 + (last - 1 - begin));
      const uint64_t limit = last < begin + count ? uni64(pv.pieces[piece0 + (last - begin)].words_off) : uni64(G->words_end);
      StreamWin sw;
      Ring ra, rb;
      pair_bind<MODE>(ra, rb, c);
      win_open(sw, c, uni64(a0->words_off), limit);
      ring_begin(sw, ra, c, uni64(a0->words_off));
      if (have_b)
        ring_begin(sw, rb, c, uni64(b0->words_off));
      const uint32_t src = (c.lane < 32 || !have_b) ? first : mid;
      uint32_t x = pv.states[(uint64_t)src * 32 + (c.lane & 31)];
      uint64_t oa = uni64(a0->out_off), ob = have_b ? uni64(b0->out_off) : 0;
      uint32_t sa = (uint32_t)((uni64(a1->out_off) - oa) / 32) + uni(a1->steps);
      uint32_t sb = have_b ? (uint32_t)((uni64(b1->out_off) - ob) / 32) + uni(b1->steps) : 0;
      ring_ready(x);
      if (have_b)
      {
        const uint32_t both = (sa < sb ? sa : sb) & ~3u;
        run_pair_groups<MODE, FAST>(x, sw, ra, rb, c, oa, ob, both);
        sa -= both;
        sb -= both;
        uint32_t xb = __shfl(x, (c.lane & 31) + 32, 64); // B's states move down to lanes 0..31 and B is finished alone
        run_groups<MODE>(xb, sw, rb, c, ob, sb);
        run_tail<MODE>(xb, rb, c, ob, uni(b1->tail));
      }
      run_groups<MODE>(x, sw, ra, c, oa, sa);
      run_tail<MODE>(x, ra, c, oa, uni(a1->tail));
    }
    else if (flags & kGroupMergeable)
    {
      open_run(!early);
      ring_ready(x);
      HSRANS_GS(const uint64_t t3 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
      // kp.group_prio (per mille; 350 by default): the younger half of the workgroup's waves decodes that share of its run at
      // raised instruction priority (s_setprio) — the SIMD otherwise serves its oldest wave first, the older half of the waves is
      // done 8 us before the younger one and waits at the round's barrier.  Unlike the one-chain-per-wave launch, whose index
      // gives the classes chains of different lengths, a block's checkpoints are where the encoder put them.
      // (not where the wave's share was already sized by its age class: 100 MB in 256 KiB blocks + G=32: 0.359 -> 0.340 with both)
      const uint32_t prio_steps = kp.group_prio != 0 && !weighted && wave >= waves / 2 ? (uint32_t)(run_steps * kp.group_prio / 1000) & ~3u : 0;
      if (prio_steps != 0)
      {
        __builtin_amdgcn_s_setprio(1);
        run_groups<MODE, true, true>(x, sw, r, c, o, prio_steps);
        __builtin_amdgcn_s_setprio(0);
      }
      run_groups<MODE, true, true>(x, sw, r, c, o, (uint32_t)run_steps - prio_steps);
      run_tail<MODE>(x, r, c, o, run_tail_syms);
      HSRANS_GS(if (HSRANS_STAMPS(kp)) {
        acc_meta += t3 - t2;
        acc_dec += __builtin_amdgcn_s_memrealtime() - t3;
      })
    }
    else if (LEAN) // fill chains (single-symbol blocks): one fill piece each
      for (uint32_t ch = first; ch < last; ch++)
      {
        const Piece *pc = pv.pieces + uni(pv.chain_first[ch]);
        wave_fill(c, uni64(pc->out_off), uni64(pc->fill_len), (uint32_t)uni64(pc->hist_off) & 0xFF);
      }
    else
      for (uint32_t ch = first; ch < last; ch++)
        run_planned_chain<MODE, true>(c, pv, ch, kp);
    HSRANS_GS(stamp_out();)
    advance();
  }
}

// Private-table launch of a 32-state plan: the wave decodes chains `ca` and `ca + 1` side by side, lanes 0..31 with the
// first chain's table, lanes 32..63 with the second's (mt_ blocks without a sidecar: every block has its own histogram).
// Anything but two plain single-piece rANS chains is done one chain after the other.
template <int MODE>
__device__ void run_private_pair(WaveCtx &c, const PlanView &pv, uint32_t ca, const KParams &kp)
{
  const uint32_t cb = ca + 1;
  const bool have_b = cb < pv.hdr->n_chains;
  const uint32_t fa = uni(pv.chain_first[ca]);
  const Piece *pa = pv.pieces + fa;
  const Piece *pb = pv.pieces + (have_b ? uni(pv.chain_first[cb]) : fa);
  bool plain = have_b && uni(pv.chain_first[ca + 1]) - fa == 1 && uni(pv.chain_first[cb + 1]) - uni(pv.chain_first[cb]) == 1;
  plain = plain && uni(pa->flags) == kPieceChainStart && uni(pb->flags) == kPieceChainStart;
  if (plain)
  {
    // both tables first (the builds borrow ring space), A's in c.table, B's in c.table_b
    WaveCtx cb_ctx = c;
    cb_ctx.table = c.table_b;
    plain = build_table<MODE, false>(c, uni64(pa->hist_off), c.lane, 64);
    plain = build_table<MODE, false>(cb_ctx, uni64(pb->hist_off), c.lane, 64) && plain;
    if (plain)
    {
      StreamWin sw;
      Ring ra, rb;
      pair_bind<MODE>(ra, rb, c);
      const uint64_t wa = uni64(pa->words_off), wb = uni64(pb->words_off);
      win_open(sw, c, wa < wb ? wa : wb, c.stream_len);
      ring_begin(sw, ra, c, wa);
      ring_begin(sw, rb, c, wb);
      uint32_t x = pv.states[(uint64_t)(c.lane < 32 ? uni(pa->state_idx) : uni(pb->state_idx)) * 32 + (c.lane & 31)];
      uint64_t oa = uni64(pa->out_off), ob = uni64(pb->out_off);
      uint32_t sa = uni(pa->steps), sb = uni(pb->steps);
      ring_ready(x);
      const uint32_t both = (sa < sb ? sa : sb) & ~3u;
      run_pair_groups<MODE>(x, sw, ra, rb, c, oa, ob, both);
      sa -= both;
      sb -= both;
      uint32_t xb = __shfl(x, (c.lane & 31) + 32, 64); // B's states move down to lanes 0..31; B is finished alone with its table
      run_groups<MODE>(xb, sw, rb, cb_ctx, ob, sb);
      run_tail<MODE>(xb, rb, cb_ctx, ob, uni(pb->tail));
      run_groups<MODE>(x, sw, ra, c, oa, sa);
      run_tail<MODE>(x, ra, c, oa, uni(pa->tail));
      return;
    }
    // a histogram did not sum up: the status bit is set; decode what can be decoded the ordinary way
  }
  run_planned_chain<MODE, false>(c, pv, ca, kp);
  if (have_b)
    run_planned_chain<MODE, false>(c, pv, cb, kp);
}

// block_ container without checkpoints: one wave follows the inline headers exactly like
// block_rANS32x64_16w_decode.cpp:47-123 (states carry over, histogram swapped per block).
template <int MODE>
__device__ void run_block_walk(const WaveCtx &c, const PlanView &pv, const KParams &kp)
{
  const uint32_t S = c.S;
  const uint64_t out_len = pv.hdr->decoded_len;
  const uint64_t whole = out_len - S + 1; // host guarantees out_len >= S - 1
  uint32_t x = c.lane < S ? pv.states[c.lane] : 0;
  uint64_t pos = pv.hdr->aux_off;
  uint64_t i = 0;
  bool have_table = false;
  uint32_t n_blocks = 0;
  StreamWin sw;
  Ring r;
  ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
  do
  {
    if (pos + 8 > c.stream_len)
    {
      if (c.lane == 0)
        atomicOr(c.status, kStatusOutOfRange);
      return;
    }
    uint64_t hdr = 0;
    for (int b = 3; b >= 0; b--) // stream offsets are only 2-byte aligned
      hdr = (hdr << 16) | *(const uint16_t *)(c.stream + pos + 2 * b);
    hdr = uni64(hdr);
    if (kp.ckpt_interval != 0) // index-build pass: where this block starts and the states the decoder enters it with
    {
      if (n_blocks >= kp.walk_max_blocks)
      {
        if (c.lane == 0)
          atomicOr(c.status, kStatusOutOfRange);
        return;
      }
      if (c.lane == 0)
      {
        kp.walk_blocks[3 * (uint64_t)n_blocks] = pos;
        kp.walk_blocks[3 * (uint64_t)n_blocks + 1] = i;
        kp.walk_blocks[3 * (uint64_t)n_blocks + 2] = hdr;
        kp.walk_count[0] = n_blocks + 1;
      }
      if (c.lane < S)
        kp.walk_states[(uint64_t)n_blocks * S + c.lane] = x;
      n_blocks++;
    }
    pos += 8;
    if (hdr >> 63)
    {
      const uint64_t len = hdr & (((uint64_t)1 << 54) - 1);
      if (len == 0 || len > c.out_cap - i) // len == 0 would never terminate
      {
        if (c.lane == 0)
          atomicOr(c.status, kStatusOutOfRange);
        return;
      }
      wave_fill(c, i, len, (uint32_t)(hdr >> 54) & 0xFF);
      i += len;
    }
    else
    {
      if (hdr == 0) // empty block: the walk would never terminate
      {
        if (c.lane == 0)
          atomicOr(c.status, kStatusBadBlock);
        return;
      }
      if (!build_table<MODE, false>(c, pos, c.lane, 64))
        return;
      have_table = true;
      pos += 512;
      uint64_t end = i + hdr;
      if (end > whole)
        end = whole;
      else if (end & (S - 1))
      {
        if (c.lane == 0)
          atomicOr(c.status, kStatusBadBlock);
        return;
      }
      ring_init(sw, r, c, pos, x);
      uint64_t steps = end > i ? (end - i + S - 1) / S : 0;
      if (kp.ckpt_interval != 0)
      {
        // checkpoints every ckpt_interval groups of the block, slot = absolute group / interval (unique: see run_planned_chain)
        uint64_t g = 0;
        const uint64_t g_abs0 = i / S;
        while (steps > 0)
        {
          if (g != 0)
          {
            const uint64_t slot = (g_abs0 + g) / kp.ckpt_interval;
            if (c.lane < S)
              kp.ckpt_states[slot * S + c.lane] = x;
            if (c.lane == 0)
              kp.ckpt_words[slot] = ring_pos(sw, r);
          }
          const uint32_t n = steps < kp.ckpt_interval ? (uint32_t)steps : kp.ckpt_interval;
          run_groups<MODE>(x, sw, r, c, i, n);
          steps -= n;
          g += n;
        }
      }
      else
        run_groups<MODE>(x, sw, r, c, i, (uint32_t)steps);
      pos = ring_pos(sw, r);
    }
    if (i > whole)
    {
      if (i >= out_len)
        return;
      break;
    }
  } while (i < whole);

  if (i < out_len)
  {
    if (!have_table) // tail without any histogram read: inplace_make_hist_dec of all-zero counts fails (decode.cpp:97-98)
    {
      if (c.lane == 0)
        atomicOr(c.status, kStatusBadHist);
      return;
    }
    ring_init(sw, r, c, pos, x);
    run_tail<MODE>(x, r, c, i, (uint32_t)(out_len - i));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel: blockDim.x = 64 * waves; wave w of block b runs chain b * waves + w
// LDS: SHARED  -> [waves x ring][table];   otherwise -> per wave [ring][table]
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool SHARED>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;

  const PlanView pv = plan_view(kp.plan);
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t bits = pv.hdr->bits;
  const uint32_t table_bytes = table_bytes_for(MODE, bits);

  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = bits;
  c.S = pv.hdr->states;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(bits));

  const uint32_t chain = blockIdx.x * waves + wave;

  if (SHARED)
  {
    const uint32_t ring_stride = fast_ring_mode(MODE) ? kFastRingBytes : kWaveRingBytes; // (launch_shape sizes the LDS the same way)
    uint8_t *ring0 = table_first_mode(MODE) ? smem + table_bytes : smem;
    c.rings = ring0 + wave * ring_stride;
    c.table = table_first_mode(MODE) ? smem : smem + waves * ring_stride;
    c.table_b = c.table;
    c.scratch_cnt = (uint16_t *)ring0;         // wave 0's ring (no request in flight while a table is built)
    c.scratch_cum = (uint16_t *)(ring0 + 512);
    const uint64_t hist_off = pv.hdr->aux_off; // shared plans: the one histogram every chain uses
    c.gtable = kp.pa.table;
    if (kp.pa.pieces != nullptr)
    {
      // (one-chain-per-wave plans, interval == 0, have a kernel of their own: k_decode_direct)
      if (c.S == 32)
        run_persistent_pair<MODE>(c, kp, waves, chain);
      else
        run_persistent<MODE>(c, kp, waves, chain);
      return;
    }
    // (grouped launches have a kernel of their own: k_decode_grouped)
    build_table<MODE, true>(c, hist_off, threadIdx.x, blockDim.x);
    if (chain < pv.hdr->n_chains)
      run_planned_chain<MODE, true>(c, pv, chain, kp);
  }
  else
  {
    const uint32_t table_stride = (table_bytes + 15) & ~15u;
    c.rings = smem + wave * kWaveRingBytes; // all rings first: they stay kRingBytes-aligned
    c.table = smem + waves * kWaveRingBytes + wave * table_stride * (kp.private_pair ? 2 : 1);
    c.table_b = kp.private_pair ? c.table + table_stride : c.table;
    c.gtable = nullptr;
    c.scratch_cnt = (uint16_t *)c.rings;
    c.scratch_cum = (uint16_t *)(c.rings + 512);
    if (pv.hdr->flags & kPlanWalk)
    {
      if (chain == 0)
        run_block_walk<MODE>(c, pv, kp);
    }
    else if (kp.private_pair)
    {
      if (2 * chain < pv.hdr->n_chains)
        run_private_pair<MODE>(c, pv, 2 * chain, kp);
    }
    else if (chain < pv.hdr->n_chains)
      run_planned_chain<MODE, false>(c, pv, chain, kp);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The kernel of the one-chain-per-wave launches (run_direct / run_direct_pair): a kernel of its own so that the headline path
// gets its own register allocation and inlining budget instead of sharing k_decode's with five other launch shapes.
// LDS: [waves x ring][table] as k_decode<MODE, true>.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_direct(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = kp.pa.bits;
  c.S = kp.pa.S;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  const uint32_t ring_stride = fast_ring_mode(MODE) ? kFastRingBytes : kWaveRingBytes; // (launch_shape sizes the LDS the same way)
  uint8_t *ring0 = table_first_mode(MODE) ? smem + table_bytes_for(MODE, c.bits) : smem;
  c.rings = ring0 + wave * ring_stride;
  c.table = table_first_mode(MODE) ? smem : smem + waves * ring_stride;
  c.table_b = c.table;
  c.gtable = kp.pa.table;
  c.scratch_cnt = (uint16_t *)ring0; // wave 0's ring (no request in flight while a table is built)
  c.scratch_cum = (uint16_t *)(ring0 + 512);
  const uint32_t chain = blockIdx.x * waves + wave;
  if (c.S == 32)
    run_direct_pair<MODE>(c, kp, waves, chain);
  else
    run_direct<MODE>(c, kp, waves, chain);
}

// hsrans_ctx_calibrate's launches: k_decode_direct<kModePack64> under a name of its own, so that a profile of a run that
// calibrates first (bench.py does) lists the calibration's launches — a 48 MiB stream, finish stamps on — apart from the decodes
// it is there to measure (rocprofv3 --stats averages per kernel name).
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_calibrate(KParams kp)
{
  constexpr int MODE = kModePack64;
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = kp.pa.bits;
  c.S = kp.pa.S;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  c.rings = smem + wave * kFastRingBytes;
  c.table = smem + waves * kFastRingBytes;
  c.table_b = c.table;
  c.gtable = kp.pa.table;
  c.scratch_cnt = (uint16_t *)smem;
  c.scratch_cum = (uint16_t *)(smem + 512);
  run_direct<MODE>(c, kp, waves, blockIdx.x * waves + wave);
}

// The kernel of the grouped launches (block_/mt_ plans with checkpoints: one workgroup per block, run_grouped) — BASELINE config 4's
// kernel.  A kernel of its own for the same reason as k_decode_direct: inside k_decode<MODE, true> it shared one register
// allocation with five other launch shapes (two more VGPRs there are the difference between 8 and 7 waves per SIMD).
// LDS: [waves x ring][table][2 next-group words, 64 B][table-build scratch, 1 KiB].
template <int MODE, bool LEAN>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_grouped(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const PlanView pv = plan_view(kp.plan);
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = pv.hdr->bits;
  c.S = pv.hdr->states;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  const uint32_t ring_stride = fast_ring_mode(MODE) ? kFastRingBytes : kWaveRingBytes; // (launch_shape sizes the LDS the same way)
  // [rings][table][next-group words][build scratch], or with the table first: [table][next-group words][build scratch][rings]
  c.rings = (table_first_mode(MODE) ? smem + table_bytes_for(MODE, c.bits) + 64 + 1024 : smem) + wave * ring_stride;
  c.table = table_first_mode(MODE) ? smem : smem + waves * ring_stride;
  c.table_b = c.table;
  c.gtable = nullptr;
  // the table build's scratch has an area of its own: a round's first stream chunks are requested before its table is built
  c.scratch_cnt = (uint16_t *)(c.table + table_bytes_for(MODE, c.bits) + 64);
  c.scratch_cum = c.scratch_cnt + 256;
  run_grouped<MODE, LEAN>(c, pv, kp, waves, wave);
}

// The same for uniform-interval raw plans with the 8-byte table (run_persistent / run_persistent_pair): in a kernel of their own
// the 32-state pair loop can be the hand-scheduled one as well (inside k_decode<3, true> its pinned registers mean scratch).
template <int MODE> // kModePack64, kModeRank (14 / 15 bits)
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_persist(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = kp.pa.bits;
  c.S = kp.pa.S;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  uint8_t *ring0 = table_first_mode(MODE) ? smem + table_bytes_for(MODE, c.bits) : smem;
  c.rings = ring0 + wave * kFastRingBytes;
  c.table = table_first_mode(MODE) ? smem : smem + waves * kFastRingBytes;
  c.table_b = c.table;
  c.gtable = kp.pa.table;
  c.scratch_cnt = (uint16_t *)ring0;
  c.scratch_cum = (uint16_t *)(ring0 + 512);
  const uint32_t chain = blockIdx.x * waves + wave;
  if (c.S == 32)
    run_persistent_pair<MODE, true>(c, kp, waves, chain);
  else
    run_persistent<MODE>(c, kp, waves, chain);
}

// ---------------------------------------------------------------------------------------------------------------
// Two 64-state chains per wave (k_decode_dual): for the table layouts that leave room for only one workgroup per CU (the
// 8-byte-per-slot table at 13 bits: 64 KiB) or whose group step is three dependent LDS round trips (the rank table at 14 / 15
// bits), a wave's single dependent chain leaves the SIMD idle most of the time (4 waves per SIMD, each waiting on LDS).
// Wave w decodes chains 2w and 2w + 1 of a one-chain-per-wave index side by side: two independent dependency chains in one
// instruction stream, which the scheduler interleaves.
//
// Two rings per wave need exact waits: "vmcnt(2)" in ring_advance is right for ONE ring (derivation there) but would make
// ring A wait for a request ring B issued a moment ago — a full memory round trip every few groups.  So this path COUNTS its
// vector-memory instructions (the stream requests and the output stores are all issued from asm here, nothing else touches
// vmcnt inside the loop) and waits with vmcnt(number of operations issued after the one it needs): exact, because vector
// memory operations of a wave complete in issue order.
// ---------------------------------------------------------------------------------------------------------------
struct RingD
{
  Ring r;
  uint32_t seq1, seq2, seq3; // value of the wave's VM-instruction count right after the requests for chunks k+1 / k+2 / k+3
};

__device__ __forceinline__ void wait_vm_outstanding(uint32_t n) // returns when at most n vector-memory operations are outstanding (n wave-uniform)
{
  switch (n)
  {
  case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
  case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
  case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
  case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
  case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
  case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
  case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
  case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
  case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
  case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break; // more than 16 younger operations: waiting for fewer is only stricter
  }
}

__device__ __forceinline__ void ring_request_counted(const StreamWin &sw, const Ring &r, const WaveCtx &c, uint32_t chunk, uint32_t &vm)
{
  ring_request(sw, r, c, chunk);
  vm += (chunk & (kRingSlots - 1)) == 0 ? 2 : 1; // slot 0 also refills the mirror
}

// (`later`: chunks 0 and 1 only, without chunk 0's mirror — the caller asks for the rest once what its first groups read has landed: ring_begin)
__device__ __forceinline__ void ring_begin_counted(const StreamWin &sw, RingD &d, const WaveCtx &c, uint64_t pos, uint32_t &vm, bool later = false)
{
  pos = uni64(pos);
  const uint32_t rel = (uint32_t)(pos - sw.base);
  d.r.voff0 = rel & ~15u;
  d.r.cur = (rel - d.r.voff0) >> 1;
  d.r.k = 0;
  if (later)
  {
    ring_request(sw, d.r, c, 0, false);
    ring_request(sw, d.r, c, 1);
    vm += 2;
    d.seq1 = d.seq2 = d.seq3 = vm;
    return;
  }
  ring_request_counted(sw, d.r, c, 0, vm);
  ring_request_counted(sw, d.r, c, 1, vm);
  d.seq1 = vm;
  ring_request_counted(sw, d.r, c, 2, vm);
  d.seq2 = vm;
  if (HSRANS_RING_AHEAD == 3)
    ring_request_counted(sw, d.r, c, 3, vm);
  d.seq3 = vm;
}

// call at least once per 256 consumed words (see ring_advance)
__device__ __forceinline__ void ring_advance_counted(const StreamWin &sw, RingD &d, const WaveCtx &c, uint32_t &vm)
{
  if ((d.r.cur >> (d.r.clog - 1)) > d.r.k)
  {
    d.r.k++;
    d.seq1 = d.seq2;
    d.seq2 = d.seq3;
    ring_request_counted(sw, d.r, c, d.r.k + HSRANS_RING_AHEAD, vm);
    if (HSRANS_RING_AHEAD == 3)
      d.seq3 = vm;
    else
      d.seq2 = d.seq3 = vm;
    wait_vm_outstanding(vm - d.seq1); // everything up to the request for chunk k+1 has completed
  }
}

__device__ __forceinline__ void store_counted(uint8_t *row_base, uint32_t voff, uint32_t v, uint32_t &vm)
{
  HSRANS_STORE_U32_SADDR(row_base, voff, v);
  vm++;
}

// One group of chain A and one of chain B, written out side by side: inline asm with side effects (the EXEC-masked merge at the
// end of group_step) is a scheduling barrier for the compiler, so two group_step calls in a row are emitted one after the other,
// each LDS read followed by its own full wait (seen in the ISA).  Here both table gathers are issued before either is needed,
// both word reads likewise, and ONE asm block at the end merges both chains.
template <int MODE>
__device__ __forceinline__ void group_step_dual(uint32_t &xa, uint32_t &xb, Ring &ra, Ring &rb, const WaveCtx &c, uint32_t &ea, uint32_t &eb)
{
  const uint32_t slot_a = xa & c.v_mask, slot_b = xb & c.v_mask;
  const uint32_t qa = xa >> c.v_bits, qb = xb >> c.v_bits;
  uint32_t nxa, nxb;
  if (MODE == kModePack64)
  {
    const uint2 ta = ((const uint2 *)c.table)[slot_a];
    const uint2 tb = ((const uint2 *)c.table)[slot_b];
    ea = ta.x;
    eb = tb.x;
    nxa = __umul24(qa, ta.x) + ta.y;
    nxb = __umul24(qb, tb.x) + tb.y;
  }
  else // kModeRank
  {
    const uint2 *ent = (const uint2 *)(c.table + (1u << c.bits));
    const uint2 ta = ent[c.table[slot_a]];
    const uint2 tb = ent[c.table[slot_b]];
    ea = ta.x;
    eb = tb.x;
    nxa = __umul24(qa, ta.x) + ta.y + slot_a;
    nxb = __umul24(qb, tb.x) + tb.y + slot_b;
  }
  const unsigned long long ma = __builtin_amdgcn_ballot_w64(nxa < kConsume);
  const unsigned long long mb = __builtin_amdgcn_ballot_w64(nxb < kConsume);
  const uint32_t rank_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0));
  const uint32_t rank_b = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0));
  uint32_t wa_addr, wb_addr;
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(wa_addr) : "v"(rank_a), "s"(ra.lds + ((ra.cur << 1) & (ring_bytes(ra) - 1))));
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(wb_addr) : "v"(rank_b), "s"(rb.lds + ((rb.cur << 1) & (ring_bytes(rb) - 1))));
  const uint32_t wa = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)wa_addr;
  const uint32_t wb = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)wb_addr;
  xa = nxa;
  xb = nxb;
  // x = low ? (nx << 16 | w) : nx for both chains (EXEC is all ones here: wave-uniform control flow of a full wave)
  asm volatile("s_mov_b64 exec, %4\n\tv_lshl_or_b32 %0, %0, 16, %2\n\ts_mov_b64 exec, %5\n\tv_lshl_or_b32 %1, %1, 16, %3\n\ts_mov_b64 exec, -1"
               : "+v"(xa), "+v"(xb)
               : "v"(wa), "v"(wb), "s"(ma), "s"(mb));
  ra.cur += (uint32_t)__popcll(ma);
  rb.cur += (uint32_t)__popcll(mb);
}

// The same step hand-scheduled (8-byte table entries): at 4 waves per SIMD a wave issues one instruction every 4-5 cycles, so
// the instruction COUNT per group is what a two-chain wave is bound by — the compiler's version of the loop above spends ~59
// instructions per group (31 of them scalar: two wrapped cursors, two rings' advance logic, mask bookkeeping); this one 16.5:
// cursors are plain LDS addresses (re-based every 4 groups: whole-chunk mirrors), chain A's mask lives in s[92:93], chain B's
// in VCC, the word reads and the merges run under EXEC = mask.
#define HSRANS_DUAL_GROUP(A0, A1, B0, B1)                                                                                                            \
  "v_and_b32 %[ta], %[xa], %[vmask]\n\t"                                                                                                             \
  "v_and_b32 %[tb], %[xb], %[vmask]\n\t"                                                                                                             \
  "v_lshl_add_u32 %[ta], %[ta], 3, %[stab]\n\t"                                                                                                      \
  "v_lshl_add_u32 %[tb], %[tb], 3, %[stab]\n\t"                                                                                                      \
  "ds_read_b64 v[" #A0 ":" #A1 "], %[ta]\n\t"                                                                                                        \
  "ds_read_b64 v[" #B0 ":" #B1 "], %[tb]\n\t"                                                                                                        \
  "v_lshrrev_b32 %[xa], %[vbits], %[xa]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[xb], %[vbits], %[xb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xa], v" #A0 ", %[xa], v" #A1 "\n\t"                                                                                               \
  "v_cmp_gt_u32 s[92:93], %[lim], %[xa]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xb], v" #B0 ", %[xb], v" #B1 "\n\t"                                                                                               \
  "v_cmp_gt_u32 vcc, %[lim], %[xb]\n\t"                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[ta], s92, 0\n\t"                                                                                                            \
  "v_mbcnt_hi_u32_b32 %[ta], s93, %[ta]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[ta], %[ta], 1, %[sa]\n\t"                                                                                                       \
  "v_mbcnt_lo_u32_b32 %[tb], vcc_lo, 0\n\t"                                                                                                         \
  "v_mbcnt_hi_u32_b32 %[tb], vcc_hi, %[tb]\n\t"                                                                                                     \
  "v_lshl_add_u32 %[tb], %[tb], 1, %[sb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "ds_read_u16 %[ta], %[ta]\n\t"                                                                                                                    \
  "s_mov_b64 exec, vcc\n\t"                                                                                                                         \
  "ds_read_u16 %[tb], %[tb]\n\t"                                                                                                                    \
  "s_bcnt1_i32_b64 %[st], s[92:93]\n\t"                                                                                                             \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                         \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                  \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_lshl_or_b32 %[xb], %[xb], 16, %[tb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "v_lshl_or_b32 %[xa], %[xa], 16, %[ta]\n\t"                                                                                                       \
  "s_mov_b64 exec, -1\n\t"

// four groups of chain A and four of chain B; acc_a / acc_b = this lane's four symbols of each (before the quad transpose)
__device__ __forceinline__ void dual_groups4(uint32_t &xa, uint32_t &xb, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_table, uint32_t &acc_a, uint32_t &acc_b)
{
  uint32_t ta, tb, st;
  asm volatile(HSRANS_DUAL_GROUP(64, 65, 72, 73) HSRANS_DUAL_GROUP(66, 67, 74, 75) HSRANS_DUAL_GROUP(68, 69, 76, 77) HSRANS_DUAL_GROUP(70, 71, 78, 79)
               "v_perm_b32 %[aa], v66, v64, %[selp]\n\t"
               "v_perm_b32 %[ta], v70, v68, %[selp]\n\t"
               "v_perm_b32 %[aa], %[ta], %[aa], %[selq]\n\t"
               "v_perm_b32 %[ab], v74, v72, %[selp]\n\t"
               "v_perm_b32 %[tb], v78, v76, %[selp]\n\t"
               "v_perm_b32 %[ab], %[tb], %[ab], %[selq]"
               : [xa] "+v"(xa), [xb] "+v"(xb), [sa] "+s"(s_a), [sb] "+s"(s_b), [aa] "=&v"(acc_a), [ab] "=&v"(acc_b), [ta] "=&v"(ta), [tb] "=&v"(tb), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [stab] "s"(s_table), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "s92", "s93", "vcc", "scc", "memory");
}

// The step for the rank table (kModeRank; 14 / 15 bits).  The rank bytes start at LDS address 0 (k_decode_dual puts the table
// first), so the slot is the address of the first gather; the entries follow at %[sent] = 2^bits.  Three dependent LDS reads
// per group and chain — rank byte, entry, stream word — the two chains' reads interleaved; A's mask in s[92:93], B's in VCC.
// Per group and chain: 10 vector instructions (+ packing), 3 LDS.  Measured (15 bits, 100 MB): 12.3 vector instructions and
// 13.3 LDS cycles per group, 7.3 of them bank conflicts — 5 from the byte gather alone: 64 random dwords over the LDS's 32 banks
// (with every lane reading ONE entry the conflicts fall to 5.0, with the byte read made conflict-free as well to 0.03 and the
// LDS cycles to 6.1; one-off builds with the gathers' addresses replaced, not kept).
#define HSRANS_DUAL_GROUP_RANK(A0, A1, B0, B1)                                                                                                       \
  "v_and_b32 %[ga], %[xa], %[vmask]\n\t"                                                                                                             \
  "v_and_b32 %[gb], %[xb], %[vmask]\n\t"                                                                                                             \
  "ds_read_u8 v" #A0 ", %[ga]\n\t"                                                                                                                   \
  "ds_read_u8 v" #B0 ", %[gb]\n\t"                                                                                                                   \
  "v_lshrrev_b32 %[xa], %[vbits], %[xa]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[xb], %[vbits], %[xb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_lshl_add_u32 %[ta], v" #A0 ", 3, %[sent]\n\t"                                                                                                   \
  "ds_read_b64 v[" #A0 ":" #A1 "], %[ta]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_lshl_add_u32 %[tb], v" #B0 ", 3, %[sent]\n\t"                                                                                                   \
  "ds_read_b64 v[" #B0 ":" #B1 "], %[tb]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xa], v" #A0 ", %[xa], v" #A1 "\n\t"                                                                                               \
  "v_add_u32 %[xa], %[xa], %[ga]\n\t"                                                                                                                \
  "v_cmp_gt_u32 s[92:93], %[lim], %[xa]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xb], v" #B0 ", %[xb], v" #B1 "\n\t"                                                                                               \
  "v_add_u32 %[xb], %[xb], %[gb]\n\t"                                                                                                                \
  "v_cmp_gt_u32 vcc, %[lim], %[xb]\n\t"                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[ta], s92, 0\n\t"                                                                                                            \
  "v_mbcnt_hi_u32_b32 %[ta], s93, %[ta]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[ta], %[ta], 1, %[sa]\n\t"                                                                                                       \
  "v_mbcnt_lo_u32_b32 %[tb], vcc_lo, 0\n\t"                                                                                                         \
  "v_mbcnt_hi_u32_b32 %[tb], vcc_hi, %[tb]\n\t"                                                                                                     \
  "v_lshl_add_u32 %[tb], %[tb], 1, %[sb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "ds_read_u16 %[ta], %[ta]\n\t"                                                                                                                    \
  "s_mov_b64 exec, vcc\n\t"                                                                                                                         \
  "ds_read_u16 %[tb], %[tb]\n\t"                                                                                                                    \
  "s_bcnt1_i32_b64 %[st], s[92:93]\n\t"                                                                                                             \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                         \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                  \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_lshl_or_b32 %[xb], %[xb], 16, %[tb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "v_lshl_or_b32 %[xa], %[xa], 16, %[ta]\n\t"                                                                                                       \
  "s_mov_b64 exec, -1\n\t"

__device__ __forceinline__ void dual_groups4_rank(uint32_t &xa, uint32_t &xb, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_entries, uint32_t &acc_a, uint32_t &acc_b)
{
  uint32_t ta, tb, ga, gb, st;
  asm volatile(HSRANS_DUAL_GROUP_RANK(64, 65, 72, 73) HSRANS_DUAL_GROUP_RANK(66, 67, 74, 75) HSRANS_DUAL_GROUP_RANK(68, 69, 76, 77) HSRANS_DUAL_GROUP_RANK(70, 71, 78, 79)
               "v_perm_b32 %[aa], v66, v64, %[selp]\n\t"
               "v_perm_b32 %[ta], v70, v68, %[selp]\n\t"
               "v_perm_b32 %[aa], %[ta], %[aa], %[selq]\n\t"
               "v_perm_b32 %[ab], v74, v72, %[selp]\n\t"
               "v_perm_b32 %[tb], v78, v76, %[selp]\n\t"
               "v_perm_b32 %[ab], %[tb], %[ab], %[selq]"
               : [xa] "+v"(xa), [xb] "+v"(xb), [sa] "+s"(s_a), [sb] "+s"(s_b), [aa] "=&v"(acc_a), [ab] "=&v"(acc_b), [ta] "=&v"(ta), [tb] "=&v"(tb), [ga] "=&v"(ga), [gb] "=&v"(gb),
                 [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [sent] "s"(s_entries), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "s92", "s93", "vcc", "scc", "memory");
}

// `both` (a multiple of 4) groups of each of the two chains
template <int MODE>
__device__ __forceinline__ void run_dual_fast(uint32_t &xa, uint32_t &xb, const StreamWin &sw, RingD &ra, RingD &rb, const WaveCtx &c, uint64_t &oa_ref, uint64_t &ob_ref, uint32_t both,
                                              uint32_t &vm)
{
  uint64_t oa = uni64(oa_ref), ob = uni64(ob_ref);
  const OutLanes ol = out_lanes(c.lane, 64);
  const uint32_t s_table = uni(lds_address(c.table));
  FastCursor fa = fast_cursor_open(ra.r), fb = fast_cursor_open(rb.r);
  // (the loop's bookkeeping as in run_groups_fast: nothing is counted but the iterations, one output pointer per chain)
  uint32_t iters = both >> 2;
  uint8_t *pa = (uint8_t *)uni64((uint64_t)(uintptr_t)(c.out + oa)), *pb = (uint8_t *)uni64((uint64_t)(uintptr_t)(c.out + ob)); // one pointer per chain, not base + offset
  oa += (uint64_t)iters * 256;
  ob += (uint64_t)iters * 256;
  auto crossed = [&](FastCursor &f, RingD &d) {
    fast_cursor_cross(f, d.r);
    ring_request(sw, d.r, c, d.r.k + HSRANS_RING_AHEAD);
    // (the constant wait: at most 6 outstanding = this ring's requests for k + 2 and k + 3 and the two stores of each of the two
    // iterations that any three of its crossings span.  Against the exact count (wait_after_crossing<2>): 13 / 14 / 15 bits replayed
    // 0.455 / 0.413 / 0.411 -> 0.475 / 0.421 / 0.420, 15 bits rotated 54.2 -> 53.0 us)
    if (HSRANS_RING_AHEAD == 3)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // (this ring's request for k + 2 and the two stores of an iteration)
  };
  for (; iters != 0; iters--)
  {
    uint32_t acc_a, acc_b;
    if (MODE == kModeRank)
      dual_groups4_rank(xa, xb, fa.addr, fb.addr, c, 1u << c.bits, acc_a, acc_b); // (the table starts at LDS address 0: k_decode_dual)
    else
      dual_groups4(xa, xb, fa.addr, fb.addr, c, s_table, acc_a, acc_b);
    acc_a = quad_transpose(acc_a, ol.sel_a, ol.sel_b);
    acc_b = quad_transpose(acc_b, ol.sel_a, ol.sel_b);
    HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)pa), ol.store_off, acc_a);
    HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)pb), ol.store_off, acc_b);
    pa += 256;
    pb += 256;
    if (fa.addr >= fa.next_cross)
      crossed(fa, ra);
    if (fb.addr >= fb.next_cross)
      crossed(fb, rb);
  }
  vm = 0;
  ra.seq1 = ra.seq2 = ra.seq3 = rb.seq1 = rb.seq2 = rb.seq3 = 0; // (not kept in the loop; the caller drains the queue behind it anyway)
  fast_cursor_close(fa, ra.r);
  fast_cursor_close(fb, rb.r);
  oa_ref = oa;
  ob_ref = ob;
}

template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(102))) k_decode_dual(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const PersistentArgs &pa = kp.pa;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = pa.bits;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  constexpr uint32_t kDualRing = kFastRingBytes; // whole-chunk mirrors for the hand-scheduled loop (launch_shape sizes the LDS the same way)
  if (MODE == kModeRank)
  {
    // the rank bytes at LDS address 0 (this kernel has no static LDS): the hand-scheduled group uses the slot as the address
    if (uni(lds_address(smem)) != 0) // (would decode garbage silently: report instead; the host discards the output)
    {
      if (threadIdx.x == 0)
        atomicOr(kp.status, kStatusOutOfRange);
      return;
    }
    c.table = smem;
    c.rings = smem + table_bytes_for(MODE, c.bits) + wave * 2 * kDualRing;
  }
  else
  {
    c.rings = smem + wave * 2 * kDualRing;
    c.table = smem + waves * 2 * kDualRing;
  }
  c.table_b = c.table;
  c.gtable = pa.table;
  c.scratch_cnt = (uint16_t *)smem;
  c.scratch_cum = (uint16_t *)(smem + 512);
  const uint32_t W = gridDim.x * waves;
  const uint32_t w = blockIdx.x * waves + wave;
  const uint64_t t_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t t_table = 0, t_ready = 0;

  // the host-built table (always: the launcher only picks this kernel for plans that carry their histogram)
  bool table_pending = true;
  auto fetch_table = [&]() {
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8;
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
      bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo;
      if (same && pa.hist_off >= c.stream_lo)
      {
        const uint64_t mine = *(const uint64_t *)(pa.hist_copy + 4 * c.lane);
        uint64_t theirs = 0;
        for (int b = 3; b >= 0; b--)
          theirs = (theirs << 16) | *(const uint16_t *)(c.stream + pa.hist_off + 8 * c.lane + 2 * b);
        same = mine == theirs;
      }
      if (__builtin_amdgcn_ballot_w64(!same) != 0 && c.lane == 0)
        atomicOr(c.status, kStatusBadHist);
    }
    __syncthreads();
    if (HSRANS_STAMPS(kp))
      t_table = __builtin_amdgcn_s_memrealtime();
  };

#if defined(HSRANS_DUAL_ASM) && !HSRANS_DUAL_ASM
  constexpr uint32_t kSymByte = 3;
  const OutLanes ol = out_lanes(c.lane, 64);
#endif
  for (uint32_t a = 2 * w; a < pa.n_chains; a += 2 * W)
  {
    const bool have_b = a + 1 < pa.n_chains;
    const DirectPiece da = direct_piece(c, pa, a);
    const DirectPiece db = have_b ? direct_piece(c, pa, a + 1) : da;
    uint32_t xa = pa.states[(uint64_t)a * 64 + c.lane];
    uint32_t xb = pa.states[(uint64_t)(have_b ? a + 1 : a) * 64 + c.lane];
    StreamWin sw;
    RingD ra, rb;
    ring_bind(ra.r, c.rings, 9, true);
    ring_bind(rb.r, c.rings + kDualRing, 9, true);
    uint32_t vm = 0; // vector-memory instructions issued from here on (everything older completes before them anyway)
    win_open(sw, c, da.words, have_b ? db.limit : da.limit); // the two chains are neighbours in the stream: one window
    // what the first groups read first (states, chunks 0 and 1 of both rings, the table), the chunks the rings keep ahead and the
    // mirrors behind that: every wave of the device is here at the same time and a CU takes in ~11 bytes per clock (run_direct)
    ring_begin_counted(sw, ra, c, da.words, vm, true);
    if (have_b)
      ring_begin_counted(sw, rb, c, db.words, vm, true);
    if (table_pending)
    {
      fetch_table();
      table_pending = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa), "+v"(xb)::"memory"); // start of a chain pair: states, table and the first two chunks of both rings
    ring_request_mirror0(sw, ra.r, c);
    ring_request(sw, ra.r, c, 2);
    if (have_b)
    {
      ring_request_mirror0(sw, rb.r, c);
      ring_request(sw, rb.r, c, 2);
    }
    if (HSRANS_RING_AHEAD == 3)
    {
      ring_request(sw, ra.r, c, 3);
      if (have_b)
        ring_request(sw, rb.r, c, 3);
    }
    if (!have_b)
      rb = ra;
    // (the loop's constant wait — at most 6 outstanding at a crossing — holds from its first crossing on: behind a ring's request for
    // chunk 2 there are the other requests just made and two stores per iteration since)
    vm = 0;
    ra.seq1 = ra.seq2 = ra.seq3 = rb.seq1 = rb.seq2 = rb.seq3 = 0;
    if (HSRANS_STAMPS(kp) && t_ready == 0)
      t_ready = __builtin_amdgcn_s_memrealtime();
    uint64_t oa = da.out, ob = db.out;
    uint32_t sa = da.steps, sb = have_b ? db.steps : 0;
    uint32_t both = have_b ? (sa < sb ? sa : sb) & ~3u : 0;
    sa -= both;
    sb -= both;
#if !defined(HSRANS_DUAL_ASM) || HSRANS_DUAL_ASM
    run_dual_fast<MODE>(xa, xb, sw, ra, rb, c, oa, ob, both, vm);
#else // the compiler's version of the same loop (A/B builds)
    for (; both != 0; both -= 4)
    {
      uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
      group_step_dual<MODE>(xa, xb, ra.r, rb.r, c, a0, b0);
      group_step_dual<MODE>(xa, xb, ra.r, rb.r, c, a1, b1);
      group_step_dual<MODE>(xa, xb, ra.r, rb.r, c, a2, b2);
      group_step_dual<MODE>(xa, xb, ra.r, rb.r, c, a3, b3);
      store_counted(c.out + oa, ol.store_off, pack4<kSymByte>(a0, a1, a2, a3, ol), vm);
      store_counted(c.out + ob, ol.store_off, pack4<kSymByte>(b0, b1, b2, b3, ol), vm);
      oa += 256;
      ob += 256;
      ring_advance_counted(sw, ra, c, vm);
      ring_advance_counted(sw, rb, c, vm);
    }
#endif
    // what is left (a few groups of the longer chain, the stream's final partial group): one chain at a time, the ordinary way
    // (the single-ring wait in ring_advance is only ever stricter than needed here: the other ring's requests are older or done)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    run_groups<MODE>(xa, sw, ra.r, c, oa, sa);
    if (have_b)
      run_groups<MODE>(xb, sw, rb.r, c, ob, sb);
    run_tail<MODE>(xa, ra.r, c, oa, da.tail);
    if (have_b)
      run_tail<MODE>(xb, rb.r, c, ob, db.tail);
  }
  if (table_pending)
    fetch_table();
  if (HSRANS_STAMPS(kp) && c.lane == 0)
  {
    uint64_t *st = kp.stamps + (uint64_t)w * 8;
    st[0] = t_entry;
    st[1] = t_table;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = st[3];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ONE chain, as fast as one chain goes (k_decode_single): a raw stream without an index is a single dependent chain — one
// wavefront, and its speed is the length of the dependency chain of one group.  In k_decode that chain has TWO LDS round trips
// (table gather, then — after compare, prefix count and address — the word read, whose result the next gather needs):
// ~200 cycles per 64 symbols = 0.65 GB/s.  Here a second wavefront (the producer) runs ahead through the word stream and
// leaves, for every word k, {table entry of (word_k & mask), word_k} in an LDS ring: a lane that renormalises reads its word
// AND the table entry its next step needs in one access while the other lanes gather their next entry from nx, so a group is
// ONE LDS round trip long: mad -> compare -> prefix count -> address -> LDS -> merge.
// Workgroup = 2 waves: wave 0 decodes, wave 1 produces.  LDS: [table 8 << bits][ring (entries + kSingleMirror) x 16 B][flags 64 B].
// ---------------------------------------------------------------------------------------------------------------
// The consumer's group, hand-scheduled (64 states): the current table entry lives in v60:v61, the word in v62 (register
// variables pinned there, see the loop).  All lanes form the table address of nx right after the multiply; the lanes that
// renormalise then read {next entry, word} from the ring under EXEC = mask while the others gather their next entry under
// EXEC = ~mask into the same registers.  10 vector, 3 LDS, 6 scalar instructions; one LDS round trip on the dependent chain.
#define HSRANS_SINGLE_GROUP(SEL)                                                                                                                     \
  "v_lshrrev_b32 %[t], %[vbits], %[x]\n\t"                                                                                                           \
  "v_mad_u32_u24 %[x], v60, %[t], v61\n\t"                                                                                                           \
  "v_perm_b32 %[acc], v60, %[acc], %[" #SEL "]\n\t"                                                                                                  \
  "v_and_b32 %[t2], %[x], %[vmask]\n\t"                                                                                                              \
  "v_lshl_add_u32 %[t2], %[t2], 3, %[stab]\n\t"                                                                                                      \
  "v_cmpx_gt_u32 vcc, %[lim], %[x]\n\t"                                                                                                              \
  "s_nop 1\n\t"                                                                                                                                      \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[t], %[t], 4, %[sew]\n\t"                                                                                                         \
  "ds_read_b64 v[60:61], %[t]\n\t"                                                                                                                   \
  "ds_read_b32 v62, %[t] offset:8\n\t"                                                                                                               \
  "s_not_b64 exec, vcc\n\t"                                                                                                                          \
  "ds_read_b64 v[60:61], %[t2]\n\t"                                                                                                                  \
  "s_mov_b64 exec, vcc\n\t"                                                                                                                          \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                   \
  "s_lshl4_add_u32 %[sew], %[st], %[sew]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
  "v_lshl_or_b32 %[x], %[x], 16, v62\n\t"                                                                                                            \
  "s_mov_b64 exec, -1\n\t"

__global__ void __launch_bounds__(128) k_decode_single(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const SingleArgs &a = kp.single;
  const uint32_t wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t table_bytes = 8u << a.bits;
  const uint32_t R = a.ring_entries; // power of two
  uint8_t *table = smem;
  uint8_t *ew = smem + table_bytes;
  // [0] words produced, [1] words released, [2] consumer done — LDS words, read with ds_read and made wave-uniform
  volatile __attribute__((address_space(3))) uint32_t *flags =
      (volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)lds_address(ew + (R + kSingleMirror) * 16);
  WaveCtx c{};
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.status = kp.status;
  c.bits = a.bits;
  c.S = a.S;
  c.lane = lane;
  c.table = table;
  c.scratch_cnt = (uint16_t *)ew; // the ring area doubles as the table builder's scratch
  c.scratch_cum = (uint16_t *)(ew + 512);
  if (threadIdx.x < 3)
    flags[threadIdx.x] = 0;
  build_table<kModePack64, true>(c, a.hist_off, threadIdx.x, 128); // a bad histogram raises the status bit; the output is then discarded by the host
  __syncthreads();
  const uint32_t mask = (1u << a.bits) - 1;
  const uint32_t ew_lds = uni(lds_address(ew)), table_lds = uni(lds_address(table));

  if (wave == 1)
  {
    // ---- producer: 512 words per round = 16 bytes per lane (one round of loads in flight), 8 table gathers, 8 ring writes ----
    const uint8_t *src = kp.stream + a.words_off;
    auto fetch = [&](uint64_t first_word) {
      const uint64_t byte = (first_word + 8 * lane) * 2;
      uint32_t w[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) // the stream is only 2-byte aligned; past its end words read as zero (like the bounds-checked ring)
        w[j] = a.words_off + byte + 2 * j + 2 <= kp.stream_len ? *(const uint16_t *)(src + byte + 2 * j) : 0;
      const u32x4 v = {w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16)};
      return v;
    };
    uint32_t produced = 0, released_seen = 0; // 32-bit word counters, compared by difference
    uint64_t next_word = 0;
    // four rounds of loads in flight: a round is consumed in ~2.5 us, a load from HBM takes about as long
    u32x4 q0 = fetch(0), q1 = fetch(512), q2 = fetch(1024), q3 = fetch(1536);
    next_word = 2048;
    auto round = [&](u32x4 &blk) -> bool {
      while (produced + 512 - released_seen > R) // never more than the ring ahead of what the consumer has released
      {
        if (uni(flags[2]) != 0)
          return false;
        released_seen = uni(flags[1]);
        __builtin_amdgcn_s_sleep(1);
      }
      const u32x4 cur = blk;
      blk = fetch(next_word); // refill this slot: in flight for the next three rounds
      next_word += 512;
      const uint32_t w[8] = {cur.x & 0xFFFF, cur.x >> 16, cur.y & 0xFFFF, cur.y >> 16, cur.z & 0xFFFF, cur.z >> 16, cur.w & 0xFFFF, cur.w >> 16};
      uint2 e[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++)
        e[j] = ((const uint2 *)table)[w[j] & mask];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++)
      {
        const uint32_t idx = (produced + 8 * lane + j) & (R - 1);
        const u32x4 t = {e[j].x, e[j].y, w[j], 0};
        *(u32x4 *)(ew + idx * 16) = t;
        if (idx < kSingleMirror) // the first entries once more behind the end: the reads of 4 groups never wrap
          *(u32x4 *)(ew + (idx + R) * 16) = t;
      }
      produced += 512;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0)
        flags[0] = produced;
      return true;
    };
    while (round(q0) && round(q1) && round(q2) && round(q3))
    {
    }
    return;
  }

  // ---- consumer: wave 0 ----
  const uint32_t S = a.S;
  const bool act_lane = lane < S;
  const unsigned long long act = __builtin_amdgcn_ballot_w64(act_lane);
  uint32_t x = act_lane ? kp.single_states[lane] : 0;
  const OutLanes ol = out_lanes(lane, S);
  uint32_t cur = 0, produced_seen = 0; // words
  uint32_t refreshes = 0, starved = 0;  // diagnostics (HSRANS_DEBUG_STAMPS): reads of the producer's count / of those, how many found the ring short
  const uint64_t t_begin = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  const uint64_t c_begin = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memtime() : 0; // shader clock: what does a lone wave run at?
  uint64_t o = uni64(a.out_off);
  uint32_t v_mask, v_bits;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v_mask) : "s"(mask));
  asm volatile("v_mov_b32 %0, %1" : "=v"(v_bits) : "s"(a.bits));
  unsigned long long e64 = *(const unsigned long long *)(table + (uint64_t)(x & mask) * 8); // entry of the start state
  // one group: returns the table word whose byte 3 is this lane's symbol; `lanes` = the lanes that take part
  // the ring must hold the words the next group can take (64); checked on a cached count: one LDS read per ~R words
  auto need_words = [&](uint32_t n) {
    while (produced_seen - cur < n)
    {
      produced_seen = uni(flags[0]);
      refreshes++;
      if (produced_seen - cur < n)
      {
        __builtin_amdgcn_s_sleep(1);
        starved++;
      }
    }
  };
  // (checking once per 4 groups instead — the producer can always be 256 words ahead with the big ring — measured SLOWER: 260
  // instead of 226 shader clocks per group; the per-group scalar check stays)
  auto step = [&](unsigned long long lanes) -> uint32_t {
    need_words(64);
    const uint32_t ex = (uint32_t)e64, ey = (uint32_t)(e64 >> 32);
    const uint32_t nx = __umul24(x >> v_bits, ex) + ey;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(nx < kConsume) & lanes;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    uint32_t ew_addr, tab_addr;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(ew_addr) : "v"(rank), "s"(ew_lds + ((cur & (R - 1)) << 4)));
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(tab_addr) : "v"(nx & v_mask), "s"(table_lds));
    uint32_t w;
    x = nx;
    // lanes that renormalise: {entry, word} from the ring; the others: the entry of nx from the table — complementary EXEC
    // masks, the same destination registers, all three reads in flight together; then x = nx << 16 | w on the first set
    asm volatile("s_mov_b64 exec, %5\n\t"
                 "ds_read_b64 %0, %3\n\t"
                 "ds_read_b32 %1, %3 offset:8\n\t"
                 "s_andn2_b64 exec, %6, %5\n\t"
                 "ds_read_b64 %0, %4\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_lshl_or_b32 %2, %2, 16, %1\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(e64), "=&v"(w), "+v"(x)
                 : "v"(ew_addr), "v"(tab_addr), "s"(m), "s"(lanes)
                 : "memory");
    cur += (uint32_t)__popcll(m);
    return ex;
  };
  uint32_t steps = a.steps;
  if (S == 64 && steps >= 4)
  {
    // the hand-scheduled loop: entry and word registers pinned (the asm names them), cursor as a plain LDS address
    register uint32_t r_e0 asm("v60") = (uint32_t)e64;
    register uint32_t r_e1 asm("v61") = (uint32_t)(e64 >> 32);
    const uint8_t *out_base = kp.out;
    for (; steps >= 4; steps -= 4)
    {
      need_words(256);
      uint32_t s_ew = uni(ew_lds + ((cur & (R - 1)) << 4));
      const uint32_t s_ew0 = s_ew;
      uint32_t acc, t, t2, st;
      asm volatile(HSRANS_SINGLE_GROUP(s0) HSRANS_SINGLE_GROUP(s1) HSRANS_SINGLE_GROUP(s2) HSRANS_SINGLE_GROUP(s3)
                   : [x] "+v"(x), "+v"(r_e0), "+v"(r_e1), [acc] "=&v"(acc), [t] "=&v"(t), [t2] "=&v"(t2), [st] "=&s"(st), [sew] "+s"(s_ew)
                   : [vmask] "v"(v_mask), [vbits] "v"(v_bits), [stab] "s"(table_lds), [lim] "s"(kConsume), [s0] "s"(0x0c0c0c07u), [s1] "s"(0x0c0c0700u), [s2] "s"(0x0c070100u),
                     [s3] "s"(0x07020100u)
                   : "v62", "vcc", "scc", "memory");
      cur += (s_ew - s_ew0) >> 4;
      acc = quad_transpose(acc, ol.sel_a, ol.sel_b);
      HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)(out_base + o)), ol.store_off, acc);
      o += 256;
      if (lane == 0)
        flags[1] = cur; // released: the producer may overwrite everything before the cursor
    }
    e64 = (unsigned long long)r_e0 | ((unsigned long long)r_e1 << 32);
  }
  for (; steps >= 4; steps -= 4)
  {
    const uint32_t e0 = step(act), e1 = step(act), e2 = step(act), e3 = step(act);
    const uint32_t acc = pack4<3>(e0, e1, e2, e3, ol);
    if (act_lane)
      HSRANS_STORE_U32((uint32_t *)(kp.out + o + ol.store_off), acc);
    o += 4 * S;
    if (lane == 0)
      flags[1] = cur; // released: the producer may overwrite everything before the cursor
  }
  const uint32_t p = lane_to_byte(lane);
  for (; steps > 0; steps--)
  {
    const uint32_t e = step(act);
    if (act_lane)
      kp.out[o + p] = (uint8_t)(e >> 24);
    o += S;
    if (lane == 0)
      flags[1] = cur;
  }
  if (a.tail) // the final partial group: only lanes whose byte exists take part (rANS32x64_16w.cpp:252-280)
  {
    const bool in_tail = act_lane && p < a.tail;
    const uint32_t e = step(__builtin_amdgcn_ballot_w64(in_tail));
    if (in_tail)
      kp.out[o + p] = (uint8_t)(e >> 24);
  }
  if (lane == 0)
    flags[2] = 1;
  if (HSRANS_STAMPS(kp) && lane == 0)
  {
    kp.stamps[0] = t_begin;
    kp.stamps[1] = refreshes;
    kp.stamps[2] = starved;
    kp.stamps[3] = __builtin_amdgcn_s_memrealtime();
    kp.stamps[4] = a.steps;
    kp.stamps[5] = __builtin_amdgcn_s_memtime() - c_begin;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2: mt_ header-chain walk on the device (one wavefront; lane 0 steers, all lanes copy states / sum counts).
// Mirrors hsrans::plan_build's mt_ branch step by step, which mirrors mt_rANS32x64_16w_decode.cpp:41-96.
// plan == nullptr: count only.  Otherwise plan is a blob sized for `n_chains` single-piece chains: the kernel fills
// chain_first, pieces and states (the host writes the 64-byte header).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load_u64_2b(const uint8_t *p) // stream offsets are only 2-byte aligned
{
  uint64_t v = 0;
  for (int b = 3; b >= 0; b--)
    v = (v << 16) | *(const uint16_t *)(p + 2 * b);
  return v;
}

// Pass 1, the pointer chase (one wavefront, all lanes in lockstep on uniform values): per block ONE 16-byte read
// {size, skip}, nothing else — the next header's position depends on it, so this read is the whole critical path.  Every
// block's {header position, output offset} goes to `blocks`; everything that is not needed to find the next header (start
// states, histogram sum check, the piece records) is left to pass 2, which is parallel over the blocks.
__global__ void __launch_bounds__(64) k_mt_chase(const uint8_t *in, uint64_t in_len, uint64_t out_cap, uint32_t S, uint64_t *blocks, uint32_t max_blocks,
                                                 WalkResult *result)
{
  uint32_t count = 0, error = 0;
  uint64_t out_len = 0;
  do
  {
    // the checks every reference decoder opens with (mt_…decode.cpp:15-32)
    if (in_len < 16 + 4 * (uint64_t)S + 512) { error = 1; break; }
    out_len = uni64(load_u64_2b(in));
    const uint64_t stored = uni64(load_u64_2b(in + 8));
    if (out_len > out_cap || in_len < stored || out_len == 0 || out_len + 1 < S) { error = 1; break; }
    const uint64_t whole = out_len - S + 1;
    uint64_t pos = 16, i = 0;
    bool last_is_rans = false;
    do
    {
      if (pos + 8 > in_len) { error = 2; break; }
      const uint64_t at = pos;
      // {size, skip} in one go; the skip word only exists (and is only used) for coded blocks, so it is read only when in range
      const bool have16 = pos + 16 <= in_len;
      uint64_t size_val = load_u64_2b(in + pos);
      uint64_t skip = have16 ? load_u64_2b(in + pos + 8) : 0;
      size_val = uni64(size_val);
      skip = uni64(skip);
      pos += 8;
      const uint64_t i0 = i;
      if (size_val >> 63)
      {
        const uint64_t len = size_val & (((uint64_t)1 << 54) - 1);
        if (len == 0 || i > out_len || len > out_len - i) { error = 3; break; }
        i += len;
        last_is_rans = false;
      }
      else
      {
        if (pos + 8 + 4 * (uint64_t)S + 512 > in_len) { error = 2; break; }
        pos += 8;
        if (skip > in_len) { error = 2; break; }
        const uint64_t after = pos + 2 * (skip + 1);
        uint64_t end = i + size_val;
        if (end > whole || end < i)
          end = whole;
        else if (end & (S - 1)) { error = 5; break; }
        const uint64_t steps = end > i ? (end - i + S - 1) / S : 0;
        if (steps > 0xFFFFFFFFull || size_val == 0) { error = 5; break; }
        i += steps * S;
        last_is_rans = true;
        pos = i > whole ? ~(uint64_t)0 : after; // both outcomes of mt_…decode.cpp:86-92 leave the loop
      }
      if (i >= whole && i < out_len && (!last_is_rans || out_len - i >= S)) { error = 6; break; } // see hsrans::plan_build
      if (count >= max_blocks) { error = 7; break; } // the block list is full: the host retries with a larger one
      if (threadIdx.x == 0)
      {
        blocks[2 * (uint64_t)count] = at;
        blocks[2 * (uint64_t)count + 1] = i0;
      }
      count++;
      if (pos == ~(uint64_t)0)
        break;
    } while (i < whole);
  } while (false);
  if (threadIdx.x == 0)
  {
    result->n_chains = count;
    result->error = error;
    result->decoded_len = out_len;
  }
}

// Pass 2, one wavefront per block: the block's chain record, start states and histogram sum check (what
// mt_…decode.cpp:62-72 reads from a block header), written into the plan blob sized for n_chains chains.
__global__ void __launch_bounds__(64) k_mt_fill(const uint8_t *in, uint64_t in_len, uint32_t S, uint32_t bits, const uint64_t *blocks, uint8_t *plan,
                                                uint32_t n_chains, uint64_t out_len, WalkResult *result)
{
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  uint32_t *cf = (uint32_t *)(plan + plan_chain_first_off());
  Piece *pieces = (Piece *)(plan + plan_pieces_off(n_chains));
  uint32_t *states = (uint32_t *)(plan + plan_states_off(n_chains, n_chains));
  const uint64_t whole = out_len - S + 1;
  uint64_t pos = blocks[2 * (uint64_t)b];
  const uint64_t i = blocks[2 * (uint64_t)b + 1];
  const uint64_t size_val = uni64(load_u64_2b(in + pos));
  pos += 8;
  Piece p{};
  uint64_t i_end = i;
  if (size_val >> 63)
  {
    p.flags = kPieceFill | kPieceChainStart;
    p.out_off = i;
    p.fill_len = size_val & (((uint64_t)1 << 54) - 1);
    p.hist_off = (size_val >> 54) & 0xFF;
    i_end = i + p.fill_len;
  }
  else
  {
    pos += 8; // skip
    if (lane < S)
      states[(uint64_t)b * S + lane] = (uint32_t)*(const uint16_t *)(in + pos + 4 * lane) | ((uint32_t)*(const uint16_t *)(in + pos + 4 * lane + 2) << 16);
    pos += 4 * (uint64_t)S;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < 4; k++)
      sum += *(const uint16_t *)(in + pos + 2 * (4 * lane + k));
    for (int d = 32; d >= 1; d >>= 1)
      sum += __shfl_xor(sum, d, 64);
    if (uni(sum) != (1u << bits) && lane == 0) // inplace_complete_hist, hist.cpp:308-324
      atomicMax(&result->error, 4u);
    p.flags = kPieceChainStart;
    p.hist_off = pos;
    p.words_off = pos + 512;
    p.out_off = i;
    uint64_t end = i + size_val;
    if (end > whole || end < i)
      end = whole;
    const uint64_t steps = end > i ? (end - i + S - 1) / S : 0;
    p.steps = (uint32_t)steps;
    i_end = i + steps * S;
  }
  p.state_idx = b;
  if (i_end >= whole && i_end < out_len)
    p.tail = (uint16_t)(out_len - i_end); // final partial group: a tail on the last chain (mt_…decode.cpp:99-130)
  if (lane == 0)
  {
    pieces[b] = p;
    cf[b] = b;
    if (b + 1 == n_chains)
      cf[n_chains] = n_chains;
  }
}

hipError_t launch_mt_chase(const uint8_t *d_stream, uint64_t stream_len, uint64_t out_cap, uint32_t S, uint64_t *d_blocks, uint32_t max_blocks, WalkResult *d_result,
                           hipStream_t stream)
{
  (void)hipGetLastError(); // (the runtime's last error is sticky per thread: an earlier failed call must not be reported as this launch's)
  hipLaunchKernelGGL(k_mt_chase, dim3(1), dim3(64), 0, stream, d_stream, stream_len, out_cap, S, d_blocks, max_blocks, d_result);
  return hipGetLastError();
}

hipError_t launch_mt_fill(const uint8_t *d_stream, uint64_t stream_len, uint32_t S, uint32_t bits, const uint64_t *d_blocks, uint8_t *d_plan, uint32_t n_chains,
                          uint64_t out_len, WalkResult *d_result, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_mt_fill, dim3(n_chains), dim3(64), 0, stream, d_stream, stream_len, S, bits, d_blocks, d_plan, n_chains, out_len, d_result);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------------------------
// tuning knobs (environment, read once per process; the defaults are the measured best)
static uint32_t g_pack64_max_bits = 14; // HSRANS_PACK64_MAX_BITS: widest histogram decoded with the 8-byte-per-slot shared table
static uint32_t g_waves_per_wg = 16;    // HSRANS_WAVES_PER_WG: waves per workgroup of the shared-table launches
static uint32_t g_static_percent = 100; // HSRANS_STATIC_PERCENT: share of the chains handed out statically (uniform persistent launches)
// HSRANS_SLOT_WEIGHTS: per-mille run length of the 8 wave classes, see PersistentArgs::run_len.  Measured on
// MI355X at 8 waves per SIMD (bits <= 12): with equal runs the four age classes of a workgroup finish at 33/36/39/42 us,
// with these weights all at 39 us (tools/stamps.py), 3-7 % less kernel time; at 4 waves per SIMD (bits >= 13) equal
// runs are better and are kept.
static uint32_t g_slot_weights[8] = {1328, 1268, 1211, 1145, 1018, 875, 665, 490}; // (re-fitted after the wait fix; runs are whole chains, so this fit is coarse)
// HSRANS_SLOT_WEIGHTS4: the same for launches with one 16-wave workgroup per CU (4 waves per SIMD: 13-bit tables)
static uint32_t g_slot_weights4[8] = {1150, 1050, 950, 850, 1150, 1050, 950, 850};
// HSRANS_DIRECT_WEIGHTS / HSRANS_DIRECT_WEIGHTS4: the chain lengths of the one-chain-per-wave index (hsrans_index_boundaries),
// per mille of the mean, by the same 8 classes.  Here nothing evens out a wrong weight afterwards (the queues above only hold the
// short tail chains), so these are fitted until all classes finish together (tools/tune_weights.py: the spread of the classes'
// mean finish times goes from 19.8 us with the weights above to 0.1 us): the waves of a CU's first workgroup run ahead of
// the second one's on every SIMD, and inside a workgroup the older waves a little ahead of the younger.
// One set per occupancy (waves per SIMD): 8 = two 16-wave workgroups per CU (bits <= 12), 6 = two 12-wave workgroups (15 bits,
// rank table), 4 = one 16-wave workgroup (13 bits), 3 = one 12-wave workgroup (14 bits).
// (Re-fitted after the decode loops stopped draining the memory queue every iteration: the oldest class now runs three times as
// many groups as the youngest in the same time.  The same fit with the buffers rotated through HBM lands within 2 % of these.)
static uint32_t g_direct_weights[8] = {1396, 1332, 1244, 1131, 960, 809, 643, 484}; // (round 3, after the loop's scalar bookkeeping was trimmed: between the fits of two boxes; hsrans_ctx_calibrate fits them to the device at hand)
static uint32_t g_direct_weights6[8] = {1192, 1159, 1120, 1072, 976, 907, 829, 745};
static uint32_t g_direct_weights4[8] = {1097, 1053, 977, 873, 1098, 1053, 977, 873};
static uint32_t g_direct_weights3[8] = {1052, 1025, 986, 936, 1052, 1025, 986, 936};
// 32-state plans (two chains per wave, one per half: run_direct_pair, hand-scheduled pair loop; HSRANS_DIRECT_WEIGHTS_PAIR): with 7
// scalar instructions per group the CU's scalar unit is contended and the oldest waves get nearly all of it
static uint32_t g_direct_weights_pair[8] = {1662, 1550, 1365, 1142, 887, 656, 450, 289}; // (re-fitted twice in round 3 as the pair loop lost scalar instructions: 1847 ... 204 before)
// HSRANS_PRIVATE_PAIR: 0 = never, 1 = when there are more chains than wave slots (default), 2 = always pair the
// chains of 32-state plans in private-table launches.  Measured: 2^30 B in 16,384 blocks 1.40 -> 1.33 ms, but 100 MB in 1,526
// blocks 0.25 -> 0.30 ms (everything is latency-bound there and half as many waves are in flight)
static uint32_t g_private_pair = 1;
static bool g_weights_two_level = false; // HSRANS_WEIGHTS_TWO_LEVEL: apply the weights to the two-level table mode as well
static bool g_table_spill = false;       // HSRANS_TABLE_SPILL: host-built tables stay in global memory (kModeSpill; comparison only)
// one-chain-per-wave plans (hsrans_index_boundaries): share of the stream (per mille) left to short chains that the ticket
// queues hand to waves that are done early, and the length of those chains in groups

static void read_tuning_once();
static uint32_t g_persist_kernel = 1; // HSRANS_PERSIST_KERNEL: 0 = uniform-interval plans on k_decode<3, true> (A/B)
static uint32_t g_single_fast = 1; // HSRANS_SINGLE_FAST: 0 = un-indexed raw streams on the general kernel (one wave, two LDS round trips per group)
static bool g_rank_table = true;      // HSRANS_NO_RANK_TABLE: the wide histograms fall back to the 8-byte-per-slot / two-level tables (comparison)
static uint32_t g_dual_waves = 16; // HSRANS_DUAL_WAVES: waves per workgroup of k_decode_dual (12: two workgroups per CU fit beside a 16 KiB table)
static uint32_t g_dual = 1; // HSRANS_DUAL: 0 = never run two chains per wave (k_decode_dual), 1 = where it pays (default), 2 = for every width (experiment)
// the one-chain-per-wave weights of the dual kernel's launches (one 16-wave workgroup per CU, two chains per wave)
static uint32_t g_dual_weights[8] = {1232, 1112, 934, 722, 1232, 1112, 934, 722};        // 13 bits (8-byte table)
static uint32_t g_dual_weights_wide[8] = {1160, 1077, 955, 810, 1160, 1077, 955, 810}; // 14 / 15 bits (rank table; fitted with 4 pairs rotated: spread of the classes' finish 5.4 -> 0.2 us)

typedef void (*KernelFn)(KParams);
static KernelFn kernel_for(int mode, bool shared)
{
  switch (mode * 2 + (shared ? 1 : 0))
  {
  case 0: return k_decode<kModePack, false>;
  case 1: return k_decode<kModePack, true>;
  case 2: return k_decode<kModePackM1, false>;
  case 3: return k_decode<kModePackM1, true>;
  case 4: return k_decode<kModeTwoLevel, false>;
  case 5: return k_decode<kModeTwoLevel, true>;
  case 8: case 9: return k_decode<kModeRank, true>;
  case 10: case 11: return k_decode<kModeSpill, true>;
  default: return k_decode<kModePack64, true>;
  }
}

uint32_t pack64_max_bits() { return g_pack64_max_bits; }
bool table_spill() { return g_table_spill; }

// Which host-built decode table a plan that carries its histogram gets, and whether its launch runs two chains per wave.
// `direct`: the plan has one chain per wave (PlanHeader::interval == 0), which is what the dual kernel is written for.
TableChoice choose_table(uint32_t bits, uint32_t states, bool direct)
{
  read_tuning_once();
  TableChoice t{0, false};
  if (g_table_spill)
    t.mode = kModeSpill;
  else if (direct && g_dual == 2 && states == 64 && bits <= 12) // experiment: the dual kernel below 13 bits too
  {
    t.mode = kModePack64;
    t.dual = true;
  }
  else if (direct && g_dual && states == 64 && bits >= 13)
  {
    // one workgroup of 16 waves per CU, two chains per wave: 13 bits keeps the 8-byte-per-slot table (64 KiB + 32 rings = 144 KiB);
    // at 14 / 15 bits that table does not fit beside 32 rings, so the rank table (18 / 34 KiB).  Measured (100 MB, last wave
    // done, fitted weights, hand-scheduled loops): 13 bits 42.7 us against 49.7 us one chain per wave; 14 / 15 bits 52.3 / 52.7 us
    // (round 2's coarse + fine tables: 57.8 / 57.7; one chain per wave beside the 128 KiB one-lookup table at 14 bits: 63.3)
    t.mode = bits == 13 ? kModePack64 : kModeRank;
    t.dual = true;
  }
  else if (bits <= g_pack64_max_bits && !(bits == 14 && g_rank_table))
    t.mode = kModePack64;
  else if (bits >= 13 && g_rank_table)
    t.mode = kModeRank; // (14 bits too: beside the 128 KiB one-lookup table only 12 waves fit a CU — 0.31 against 0.37 with a checkpoint every 32 groups)
  return t;
}

size_t rank_table_entries(uint32_t bits) { return table_bytes_for(kModeRank, bits) / 8; }

size_t build_rank_table(const uint16_t counts[256], uint32_t bits, uint2 *out, size_t capacity_entries)
{
  if (bits < 13 || bits > 15 || capacity_entries < rank_table_entries(bits))
    return 0;
  // the 32 most frequent symbols — nearly every lane of a group — get 32 different bank pairs of the entry table (indexed by
  // symbol value, symbols 32 apart would share banks: 8.0 instead of 7.3 bank-conflict cycles per group)
  uint8_t *rank_of_slot = (uint8_t *)out;
  uint2 *ent = (uint2 *)(rank_of_slot + (1u << bits));
  uint32_t order[256];
  for (uint32_t s = 0; s < 256; s++)
    order[s] = s;
  std::stable_sort(order, order + 256, [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
  uint8_t rank[256];
  for (uint32_t r = 0; r < 256; r++)
    rank[order[r]] = (uint8_t)r;
  uint32_t cumul = 0;
  for (uint32_t s = 0; s < 256; s++)
  {
    ent[rank[s]] = make_uint2((uint32_t)counts[s] | (s << 24), 0u - cumul);
    for (uint32_t k = 0; k < counts[s] && cumul + k < (1u << bits); k++)
      rank_of_slot[cumul + k] = rank[s];
    cumul += counts[s];
  }
  return cumul == (1u << bits) ? rank_table_entries(bits) : 0;
}

static void read_tuning_impl();
static void read_tuning_once() // contexts may be created from several threads
{
  static std::once_flag once;
  std::call_once(once, read_tuning_impl);
}
static void read_tuning_impl()
{
  if (const char *e = getenv("HSRANS_STATIC_PERCENT"))
    g_static_percent = (uint32_t)atoi(e) > 100 ? 100 : (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_PACK64_MAX_BITS"))
    if (atoi(e) >= 9 && atoi(e) <= 14)
      g_pack64_max_bits = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_WAVES_PER_WG"))
    if (atoi(e) == 4 || atoi(e) == 8 || atoi(e) == 12 || atoi(e) == 16)
      g_waves_per_wg = (uint32_t)atoi(e);
  auto read_weights = [](const char *name, uint32_t *w) { // 8 comma-separated per-mille values, rescaled to mean 1000
    const char *e = getenv(name);
    if (e == nullptr)
      return;
    uint32_t v[8], n = 0;
    uint64_t sum = 0;
    for (const char *p = e; n < 8 && *p; n++)
    {
      v[n] = (uint32_t)strtoul(p, (char **)&p, 10);
      sum += v[n];
      if (*p == ',')
        p++;
    }
    if (n == 8 && sum > 0)
      for (uint32_t k = 0; k < 8; k++)
        w[k] = (uint32_t)((uint64_t)v[k] * 8000 / sum);
  };
  read_weights("HSRANS_SLOT_WEIGHTS", g_slot_weights);
  read_weights("HSRANS_SLOT_WEIGHTS4", g_slot_weights4);
  read_weights("HSRANS_DIRECT_WEIGHTS", g_direct_weights);
  read_weights("HSRANS_DIRECT_WEIGHTS4", g_direct_weights4);
  read_weights("HSRANS_DIRECT_WEIGHTS6", g_direct_weights6);
  read_weights("HSRANS_DIRECT_WEIGHTS3", g_direct_weights3);
  read_weights("HSRANS_DIRECT_WEIGHTS_PAIR", g_direct_weights_pair);
  g_weights_two_level = getenv("HSRANS_WEIGHTS_TWO_LEVEL") != nullptr;
  g_table_spill = getenv("HSRANS_TABLE_SPILL") != nullptr;
  if (const char *e = getenv("HSRANS_PRIVATE_PAIR"))
    g_private_pair = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_DUAL"))
    g_dual = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_DUAL_WAVES"))
    if (atoi(e) == 8 || atoi(e) == 12 || atoi(e) == 16)
      g_dual_waves = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_PERSIST_KERNEL"))
    g_persist_kernel = (uint32_t)atoi(e);
  if (const char *e = getenv("HSRANS_SINGLE_FAST"))
    g_single_fast = (uint32_t)atoi(e);
  read_weights("HSRANS_DUAL_WEIGHTS", g_dual_weights);               // 13 bits (8-byte table)
  read_weights("HSRANS_DUAL_WEIGHTS_WIDE", g_dual_weights_wide); // 14 / 15 bits (rank table)
  g_rank_table = getenv("HSRANS_NO_RANK_TABLE") == nullptr;
}

// per device (the CURRENT device): dynamic-LDS limit of every kernel variant, CU count
hipError_t prepare_kernels(DeviceGeom *geom)
{
  read_tuning_once();
  geom->max_lds = 160 * 1024;
  geom->num_cus = 256;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
    geom->num_cus = (uint32_t)cus;
  for (int mode = 0; mode < 6; mode++)
    for (int shared = 0; shared < 2; shared++)
    {
      const hipError_t e = hipFuncSetAttribute((const void *)kernel_for(mode, shared != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
      if (e != hipSuccess)
        return e;
    }
  for (KernelFn fn : {(KernelFn)k_decode_single, (KernelFn)k_decode_persist<kModePack64>, (KernelFn)k_decode_persist<kModeRank>, (KernelFn)k_calibrate, (KernelFn)k_decode_dual<kModePack64>, (KernelFn)k_decode_dual<kModeRank>, (KernelFn)k_decode_direct<kModePack>, (KernelFn)k_decode_direct<kModePackM1>,
                      (KernelFn)k_decode_direct<kModeTwoLevel>, (KernelFn)k_decode_direct<kModePack64>, (KernelFn)k_decode_direct<kModeRank>, (KernelFn)k_decode_direct<kModeSpill>,
                      (KernelFn)k_decode_grouped<kModePack, false>, (KernelFn)k_decode_grouped<kModePackM1, false>, (KernelFn)k_decode_grouped<kModeTwoLevel, false>,
                      (KernelFn)k_decode_grouped<kModePack64, false>, (KernelFn)k_decode_grouped<kModeTwoLevel, true>, (KernelFn)k_decode_grouped<kModePack64, true>,
                      (KernelFn)k_decode_grouped<kModeRank, false>, (KernelFn)k_decode_grouped<kModeRank, true>})
  {
    const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geom->max_lds);
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

DeviceGeom default_geom()
{
  read_tuning_once();
  DeviceGeom g{};
  g.max_lds = 160 * 1024;
  g.num_cus = 256; // MI355X
  return g;
}

// Everything about a launch that follows from the plan header and the device alone (no pointers): the table layout, the
// workgroup shape and the grid.  launch_decode uses it; direct_boundaries uses it to size one chain per resident wave.
LaunchShape launch_shape(const PlanHeader &h, const DeviceGeom &dg, bool persistent, uint32_t table_mode, uint32_t n_groups, bool index_pass, bool direct, bool dual)
{
  read_tuning_once();
  LaunchShape L{};
  const bool walk = (h.flags & kPlanWalk) != 0;
  const bool grouped = n_groups != 0 && !index_pass;
  L.walk = walk;
  L.shared = !walk && (grouped || (h.shared_hist != 0 && h.n_chains > 1));
  // 64-bit entries only where one table serves a whole workgroup (LDS: 16 KiB table + 16 x 2.25 KiB rings, two per CU)
  const bool rank_table = L.shared && persistent && table_mode == kModeRank;
  const bool spill = L.shared && persistent && table_mode == kModeSpill;
  // (grouped launches from 13 bits on: the rank table, built per block on the device — beside the 64 / 128 KiB one-lookup tables
  // only one workgroup fits a CU: 100 MB in 256 KiB blocks + G=32: 0.269 / 0.206 / 0.190 at 13 / 14 / 15 bits)
  const bool grouped_rank = grouped && g_rank_table && h.bits >= 13;
  L.mode = spill ? kModeSpill : rank_table || grouped_rank ? kModeRank : L.shared && h.bits <= pack64_max_bits() ? kModePack64 : h.bits >= 13 ? kModeTwoLevel : h.bits == 12 ? kModePackM1 : kModePack;
  const bool two_level = L.mode == kModeTwoLevel;
  const uint32_t table_bytes = table_bytes_for(L.mode, h.bits);
  const uint32_t wave_bytes = kWaveRingBytes + ((table_bytes + 15) & ~15u); // private rings + table
  uint32_t waves, lds, grid;
  const uint32_t dual_ring = kFastRingBytes;
  L.dual = dual && L.shared && persistent && (L.mode == kModePack64 || L.mode == kModeRank) && g_dual_waves * 2 * dual_ring + table_bytes <= dg.max_lds;
  if (L.dual)
  {
    // k_decode_dual: workgroups of 16 (or HSRANS_DUAL_WAVES) waves, two rings per wave, wave w decodes chains 2w and 2w + 1
    waves = g_dual_waves;
    lds = waves * 2 * dual_ring + table_bytes;
    const uint32_t per_cu = dg.max_lds / lds ? dg.max_lds / lds : 1;
    L.resident = dg.num_cus * (per_cu * waves > 32 ? 32 / waves : per_cu);
    grid = (h.n_chains + 2 * waves - 1) / (2 * waves);
    if (grid > L.resident)
      grid = L.resident;
  }
  else if (L.shared)
  {
    // (k_decode_direct's hand-scheduled loop wants a whole-chunk mirror behind every ring)
    const uint32_t ring = fast_ring_mode(L.mode) ? kFastRingBytes : kWaveRingBytes;
    waves = g_waves_per_wg;
    // Grouped launches (one workgroup per block, round after round): FOUR workgroups of 8 waves per CU instead of two of 16
    // wherever four fit the LDS.  A round is table build + records + first chunks (~12 us in which the workgroup decodes
    // nothing) and then the decode; with two workgroups per CU the other one is alone on the CU meanwhile, 4 waves per SIMD,
    // which is too few to hide the loop's LDS latency.  With four, three are decoding while one is between rounds, and a round
    // is twice as long for the same fixed cost.  Measured at 2^30 bytes: 256 KiB blocks 0.461 -> 0.492 of 8 TB/s, 64 KiB blocks
    // 0.358 -> 0.427 (1 MiB blocks: unchanged).  HSRANS_WAVES_PER_WG still overrides.
    // (only with at least four groups per CU: fewer, e.g. a 100 MB stream in 256 KiB blocks, fill more wave slots as 16-wave workgroups)
    if (grouped && getenv("HSRANS_WAVES_PER_WG") == nullptr && 4 * (8 * ring + table_bytes + 64 + 1024) <= dg.max_lds && n_groups >= 4 * dg.num_cus)
      waves = 8;
    if (L.mode == kModeRank && waves * ring + table_bytes > dg.max_lds / 2)
      waves = 12; // 15 bits: 48 KiB of tables + 12 rings = 75 KiB, two workgroups per CU
    if (waves * ring + table_bytes > dg.max_lds && table_bytes + 4 * ring <= dg.max_lds)
      waves = (dg.max_lds - table_bytes) / ring / 4 * 4; // a big table: as many waves as still fit (multiple of 4: one per SIMD)
    while (waves > 1 && (waves / 2 >= h.n_chains || waves * ring + table_bytes > dg.max_lds))
      waves /= 2;
    lds = waves * ring + table_bytes + (grouped ? 64 + 1024 : 0); // (run_grouped's two next-group words and its table-build scratch)
    grid = (h.n_chains + waves - 1) / waves;
    if (grouped)
      grid = n_groups;
    const uint32_t per_cu = dg.max_lds / lds ? dg.max_lds / lds : 1; // workgroups one CU can hold (LDS-limited; 32 waves max)
    L.resident = dg.num_cus * (per_cu * waves > 32 ? 32 / waves : per_cu);
    if ((persistent || grouped) && grid > L.resident)
      grid = L.resident;
  }
  else
  {
    // 32-state plans: two chains per wave, one per half, when two tables fit (run_private_pair); not for the index-build pass
    const bool pair = (g_private_pair == 2 || (g_private_pair == 1 && h.n_chains >= 32 * dg.num_cus)) && !walk && h.states == 32 && h.n_chains > 1 &&
                      table_bytes <= 16384 && !index_pass;
    L.private_pair = pair ? 1 : 0;
    const uint32_t wave_lds = pair ? wave_bytes + ((table_bytes + 15) & ~15u) : wave_bytes;
    const uint32_t work = pair ? (h.n_chains + 1) / 2 : h.n_chains;
    waves = walk ? 1 : 4;
    while (waves > 1 && (waves * wave_lds > dg.max_lds / 2 || waves / 2 >= work))
      waves /= 2;
    lds = waves * wave_lds;
    grid = walk ? 1 : (work + waves - 1) / waves;
    L.resident = dg.num_cus * (dg.max_lds / (lds ? lds : 1));
  }
  L.waves = waves;
  L.lds = lds;
  L.grid = grid ? grid : 1;
  // run-length weights of the 8 wave classes (per mille of the mean): class = (workgroup in the grid's second half) * 4 + wave / (waves / 4)
  const bool weighted = (waves == 16 || waves == 12) && (!two_level || g_weights_two_level);
  for (uint32_t k = 0; k < 8; k++)
    L.weights[k] = !weighted ? 1000
                   : L.dual   ? (L.mode == kModeRank ? g_dual_weights_wide : g_dual_weights)[k]
                   : direct   ? (L.grid > dg.num_cus ? (waves == 16 ? (h.states == 32 ? g_direct_weights_pair : dg.have_direct_weights ? dg.direct_weights : g_direct_weights) : g_direct_weights6) : (waves == 16 ? g_direct_weights4 : g_direct_weights3))[k]
                              : (L.grid > dg.num_cus ? (persistent && !grouped && h.states == 32 ? g_direct_weights_pair : g_slot_weights) : g_slot_weights4)[k];
  return L;
}

// Group boundaries that cut `total_groups` whole groups into one chain per wave of the direct launch this device would use
// for a mergeable plan of (states, bits): chain w belongs to wave w, its length follows the wave's class weight.
// Boundaries are multiples of 4 groups (the decode loop stores 4 groups at a time).  Returns the number of chains;
// out[k] = first group of chain k + 1 (k < chains - 1).
size_t direct_boundaries(const DeviceGeom &dg, uint32_t states, uint32_t bits, uint64_t total_groups, uint64_t *out, size_t cap)
{
  PlanHeader h{};
  h.states = states;
  h.bits = bits;
  h.shared_hist = 1;
  h.n_chains = 1u << 30; // "many": the full machine
  const TableChoice tc = choose_table(bits, states, true);
  const LaunchShape L = launch_shape(h, dg, true, tc.mode, 0, false, true, tc.dual);
  const uint32_t runs_per_wave = (states == 32 || L.dual) ? 2 : 1;
  const uint64_t W = (uint64_t)L.grid * L.waves;
  uint64_t chains = W * runs_per_wave;
  const uint64_t all_units = total_groups / 4; // boundaries in units of 4 groups
  if (chains > all_units / 8) // a chain is worth its 308 bytes of index and its prologue from about 32 groups on
    chains = all_units / 8 ? all_units / 8 : 1;
  if (chains <= 1)
    return 1;
  if (chains - 1 > cap)
    return 0;
  // cumulative weight up to chain k, then boundaries at units * cum / all
  const uint32_t first_half = (L.grid + 1) / 2;
  const uint32_t per_class = L.waves >= 4 ? L.waves / 4 : 1;
  auto weight_of = [&](uint64_t chain) {
    const uint64_t w = chain / runs_per_wave;
    const uint32_t blk = (uint32_t)(w / L.waves), wave_in_wg = (uint32_t)(w % L.waves);
    const uint32_t cls = (blk >= first_half ? 4 : 0) + (wave_in_wg / per_class < 4 ? wave_in_wg / per_class : 3);
    return (uint64_t)L.weights[cls];
  };
  uint64_t all = 0;
  for (uint64_t k = 0; k < chains; k++)
    all += weight_of(k);
  uint64_t cum = 0, prev = 0;
  size_t n = 0;
  for (uint64_t k = 0; k + 1 < chains; k++)
  {
    cum += weight_of(k);
    uint64_t b = (uint64_t)((unsigned __int128)all_units * cum / all);
    if (b <= prev)
      b = prev + 1; // every chain gets at least one unit
    if (b >= all_units)
      break;
    out[n++] = b * 4;
    prev = b;
  }
  return n + 1;
}

hipError_t launch_decode(const KParams &kp_in, const PlanHeader &h, const DeviceGeom &dg, hipStream_t stream, LaunchInfo *info)
{
  KParams kp = kp_in;
  const bool persistent = kp.pa.pieces != nullptr;
  const bool index_pass = kp.ckpt_interval != 0 || kp.ckpt_groups != nullptr;
  const LaunchShape L = launch_shape(h, dg, persistent, persistent && kp.pa.table != nullptr ? kp.pa.table_mode : 0, kp.groups != nullptr ? kp.n_groups : 0, index_pass, persistent && kp.pa.interval == 0, persistent && kp.pa.dual != 0);
  const bool grouped = kp.groups != nullptr && !index_pass;
  const uint32_t waves = L.waves, grid = L.grid;
  kp.private_pair = L.private_pair;

  if (grouped)
  {
    const uint32_t per_class = waves >= 4 ? waves / 4 : 1;
    for (uint32_t hf = 0; hf < 2; hf++)
    {
      uint32_t cum = 0;
      for (uint32_t k = 0; k <= 16; k++)
      {
        kp.group_cum[hf][k] = (uint16_t)cum;
        const uint32_t cls = k / per_class < 4 ? k / per_class : 3;
        cum += k < waves ? L.weights[hf * 4 + cls] / 10 : 0;
      }
    }
  }
  if (persistent && kp.pa.interval != 0)
  {
    // static share: a fixed fraction of the chains, split over the waves by class weight; the rest goes through the queues
    const uint64_t W = (uint64_t)grid * waves;
    // 32-state streams: two runs per wave (run_persistent_pair)
    kp.pa.static_per_wave = (uint32_t)((uint64_t)h.n_chains * g_static_percent / 100 / (h.states == 32 ? 2 * W : W));
    const uint32_t first_half = (grid + 1) / 2, second_half = grid - first_half;
    const uint32_t per_class = waves >= 4 ? waves / 4 : 1, classes = waves / per_class;
    const uint32_t runs_per_wave = h.states == 32 ? 2 : 1; // run_persistent_pair decodes two runs side by side
    uint32_t longest = 0;
    for (uint32_t hf = 0; hf < 2; hf++)
    {
      uint32_t off = 0;
      for (uint32_t k = 0; k < 4; k++)
      {
        kp.pa.run_len[hf * 4 + k] = k < classes ? (uint32_t)((uint64_t)h.n_chains * g_static_percent / 100 * L.weights[hf * 4 + k] / (1000 * W * runs_per_wave)) : 0;
        kp.pa.class_off[hf * 4 + k] = off;
        off += kp.pa.run_len[hf * 4 + k] * per_class * runs_per_wave;
        longest = kp.pa.run_len[hf * 4 + k] > longest ? kp.pa.run_len[hf * 4 + k] : longest;
      }
      kp.pa.wg_chains[hf] = off;
    }
    kp.pa.half_base[0] = 0;
    kp.pa.half_base[1] = first_half * kp.pa.wg_chains[0];
    kp.pa.static_total = first_half * kp.pa.wg_chains[0] + second_half * kp.pa.wg_chains[1];
    if (kp.pa.static_total > h.n_chains) // weights sum to <= 8000 by construction; belt and braces
      return hipErrorInvalidValue;
    // a merged run is read through one 32-bit window of the stream (at most one 16-bit word per symbol)
    if ((uint64_t)(longest ? longest : 1) * kp.pa.interval * h.states * 2 >= 0xFFFF0000ull)
      return hipErrorInvalidValue;
  }
  if (persistent && kp.pa.interval == 0)
  {
    // one-chain-per-wave launches: the plan's chains as W runs of consecutive chains (run_direct); 32-state and two-chain launches keep their own dealing
    const uint64_t W = (uint64_t)grid * waves;
    kp.pa.run_chains = h.states == 64 && !L.dual ? (uint32_t)((h.n_chains + W - 1) / W) : 1;
  }
  if (kp.single.valid && !index_pass && g_single_fast)
  {
    // one chain of one piece (a raw stream without an index): the two-wave latency kernel
    const uint32_t lds = (8u << h.bits) + (kp.single.ring_entries + kSingleMirror) * 16 + 64;
    if (info)
    {
      *info = LaunchInfo{};
      info->grid = 1;
      info->block = 128;
      info->lds_bytes = lds;
      info->waves_per_block = 2;
      info->chains = 1;
      info->table_mode = kModePack64;
      info->chains_per_wave = 1;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_decode_single, dim3(1), dim3(128), lds, stream, kp);
    return hipGetLastError();
  }
  KernelFn fn = kernel_for(L.mode, L.shared);
  if (L.dual)
    fn = L.mode == kModeRank ? (KernelFn)k_decode_dual<kModeRank> : (KernelFn)k_decode_dual<kModePack64>;
  else if (persistent && kp.pa.interval != 0 && L.shared && (L.mode == kModePack64 || L.mode == kModeRank) && kp.pa.table != nullptr && !index_pass && g_persist_kernel)
    fn = L.mode == kModeRank ? (KernelFn)k_decode_persist<kModeRank> : (KernelFn)k_decode_persist<kModePack64>;

  else if (grouped && L.shared)
    switch (L.mode)
    {
    case kModePack: fn = k_decode_grouped<kModePack, false>; break;
    case kModePackM1: fn = k_decode_grouped<kModePackM1, false>; break;
    case kModeTwoLevel: fn = kp.groups_lean ? k_decode_grouped<kModeTwoLevel, true> : k_decode_grouped<kModeTwoLevel, false>; break;
    case kModeRank: fn = kp.groups_lean ? k_decode_grouped<kModeRank, true> : k_decode_grouped<kModeRank, false>; break;
    default: fn = kp.groups_lean ? k_decode_grouped<kModePack64, true> : k_decode_grouped<kModePack64, false>; break;
    }
  else if (persistent && kp.pa.interval == 0 && L.shared)
    switch (L.mode)
    {
    case kModePack: fn = k_decode_direct<kModePack>; break;
    case kModePackM1: fn = k_decode_direct<kModePackM1>; break;
    case kModeTwoLevel: fn = k_decode_direct<kModeTwoLevel>; break;
    case kModeRank: fn = k_decode_direct<kModeRank>; break;
    case kModeSpill: fn = k_decode_direct<kModeSpill>; break;
    default: fn = kp.finish != nullptr && h.states == 64 ? (KernelFn)k_calibrate : (KernelFn)k_decode_direct<kModePack64>; break;
    }
  if (info)
  {
    info->grid = grid;
    info->block = waves * 64;
    info->lds_bytes = L.lds;
    info->waves_per_block = waves;
    info->chains = h.n_chains;
    info->shared_table = L.shared;
    info->walk = L.walk;
    info->two_level = L.mode == kModeTwoLevel;
    info->table_mode = (uint32_t)L.mode;
    info->chains_per_wave = L.dual ? 2 : 1;
    for (uint32_t k = 0; k < 8; k++)
      info->class_weights[k] = L.weights[k];
    info->dynamic_groups = grouped && kp.group_tickets != nullptr && kp.n_groups > grid ? 1 : 0;
  }
  (void)hipGetLastError(); // (sticky per thread: an earlier failed call — e.g. an allocation a hostile stream asked for — is not this launch's error)
  hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), L.lds, stream, kp);
  return hipGetLastError();
}

} // namespace hsrans
