// kernels_common.h — Shared device code of the decode kernels: constants, table layouts, the stream ring (LDS-DMA), the in-kernel table build, the group step and its hand-scheduled forms, the output path, the generic chain runner.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_COMMON_H
#define HSRANS_KERNELS_COMMON_H

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

// the stores off the fast path (a run's last < 4 groups, the final partial group, single-symbol fills).  WT: written through like the
// fast path's — the launches that publish completion words (a sharded decode's sub-runs in one launch, PartArgs) must not leave ANY
// decoded byte dirty in an XCD's L2, because nothing flushes it before the word is published (a device-scope release fence per
// group — buffer_wbl2, a walk of the whole L2 — was measured: 2,048 of them cost a 2^30-byte decode 190 of 200 us)
template <bool WT>
__device__ __forceinline__ void store_u8(uint8_t *ptr, uint32_t v)
{
  if (WT)
    asm volatile("global_store_byte %0, %1, off sc0 sc1" : : "v"(ptr), "v"(v) : "memory");
  else
    *ptr = (uint8_t)v;
}
template <bool WT>
__device__ __forceinline__ void store_u128(u32x4 *ptr, u32x4 v)
{
  if (WT)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(ptr), "v"(v) : "memory");
  else
    *ptr = v;
}

// Divisions the prologues can do without.  A 64-bit division is ~20 vector instructions behind a range check and ~100 when the operands
// are wide, a v_rcp each, and every wave of every round did five of them (its share of the chains = weight x count / weights' sum, twice;
// the run's groups = bytes / S): a third of what a grouped round issues outside its decode loop.
//   groups_of(S, bytes): bytes / S for S = 32 or 64 (the two state counts there are)
//   Recip / share_of: floor(a * n / d) with d's reciprocal taken once (host or kernel entry); exact while a * n < 2^32, else the division
__device__ __forceinline__ uint64_t groups_of(uint32_t S, uint64_t bytes) { return S == 64 ? bytes >> 6 : S == 32 ? bytes >> 5 : bytes / S; }
struct Recip
{
  uint32_t d, inv; // inv = floor((2^32 - 1) / d): __umulhi(n, inv) is n / d or one less for every n < 2^32
};
__host__ __device__ __forceinline__ Recip recip_of(uint32_t d)
{
  Recip r;
  r.d = d ? d : 1;
  r.inv = 0xFFFFFFFFu / r.d;
  return r;
}
__device__ __forceinline__ uint32_t share_of(uint32_t a, uint32_t n, const Recip &r)
{
  const uint64_t wide = (uint64_t)a * n;
  if (wide >> 32) // (more than a million chains in one unit: not a shape the launches that use this see, but exact when they do)
    return (uint32_t)(wide / r.d);
  const uint32_t v = (uint32_t)wide;
  uint32_t q = __umulhi(v, r.inv);
  if (v - q * r.d >= r.d)
    q++;
  return q;
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
// cache-policy bits of the stream requests: none.  Measured (round 2) on the
// 100 MB headline decode: sc1 / sc0 sc1 change nothing; nt makes the requests bypass the Infinity Cache, i.e. even a replayed
// stream comes from HBM every time (58.6 us against 44.0 us) — the default (no bits) is right.
#define HSRANS_STREAM_LOAD_FLAGS "" // (sc1 / sc0 sc1: no difference; nt: the requests bypass the Infinity Cache — measured in round 2, see above)
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
__device__ __forceinline__ void wait_vm_at_most(uint32_t n)
{
  if (n >= 8)
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
// The 8-byte table without a search per slot (round 6).  A search is 8 DEPENDENT LDS reads per slot; with 4 slots a thread
// (2,048 slots, 512 threads) that is 32 LDS round trips and ~140 vector instructions a wave per build — a 100 MB mt_ stream in
// 64 KiB blocks builds ~3,000 tables a launch, a fifth of the launch's vector instructions (SQ_INSTS_VALU 21.3 M against the raw
// stream's 17.5 M).  Instead: every symbol with a count marks its first slot with its own number (a byte per slot, kept in the
// table's own first 2^bits bytes), and a slot's symbol is the largest mark at or before it — symbols ascend with the slots.  A
// thread owns 4 or 8 CONSECUTIVE slots: the running maximum inside its marks, a wave-wide max-scan in DPP for the lanes before
// it, and for the waves before it the symbol of the wave's first slot counted directly (the number of prefix sums <= that slot:
// four ballots).  Four LDS round trips in all.  Same table as the search for every histogram that sums to 2^bits (the largest s
// with cum[s] <= slot: zero-count symbols mark nothing); for the others the status bit is already raised and every index stays
// in range (marks are symbols, slots beyond 2^bits are not marked).
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr bool pack64_marks(uint32_t total, uint32_t nthreads)
{
  // a thread's run = 4 or 8 slots (its marks are held in two registers across the barrier that frees them for the table)
  return (nthreads & (nthreads - 1)) == 0 && nthreads >= 64 && total >= 1024 && total / (nthreads < total / 4 ? nthreads : total / 4) <= 8;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t max_dpp(uint32_t v)
{
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true); // lanes without a source: 0
  return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) // inclusive, lanes 0 .. 63
{
  v = max_dpp<0x111, 0xF>(v); // row_shr:1
  v = max_dpp<0x112, 0xF>(v); // row_shr:2
  v = max_dpp<0x114, 0xF>(v); // row_shr:4
  v = max_dpp<0x118, 0xF>(v); // row_shr:8
  v = max_dpp<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
  v = max_dpp<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ void pack64_zero_marks(uint2 *tab, uint32_t total, uint32_t tid, uint32_t nthreads)
{
  for (uint32_t q = tid; q < total / 4; q += nthreads)
    ((uint32_t *)tab)[q] = 0;
}
// by the thread that holds four consecutive symbols' counts and the prefix sum before them (the marks are zero by now)
__device__ __forceinline__ void pack64_mark4(uint2 *tab, uint32_t total, uint32_t s0, uint32_t excl, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
  uint8_t *mk = (uint8_t *)tab;
  const uint32_t u0 = excl, u1 = u0 + c0, u2 = u1 + c1, u3 = u2 + c2;
  if (c0 != 0 && u0 < total)
    mk[u0] = (uint8_t)s0;
  if (c1 != 0 && u1 < total)
    mk[u1] = (uint8_t)(s0 + 1);
  if (c2 != 0 && u2 < total)
    mk[u2] = (uint8_t)(s0 + 2);
  if (c3 != 0 && u3 < total)
    mk[u3] = (uint8_t)(s0 + 3);
}
// marks, cnt[] and cum[] complete and visible -> the table; SYNC() between the last read of a mark and the first table store
template <typename SYNC>
__device__ __forceinline__ void pack64_from_marks(uint2 *tab, const uint16_t *cnt, const uint16_t *cum, uint32_t total, uint32_t tid, uint32_t nthreads, SYNC sync)
{
  const uint32_t active = nthreads < total / 4 ? nthreads : total / 4; // a multiple of 64: whole waves take part or not at all
  const uint32_t per = total / active;                                  // 4 or 8 (pack64_marks)
  const bool mine = tid < active;
  uint32_t w0 = 0, w1 = 0, prev = 0; // all that is held across the barrier: the thread's marks and the largest mark before them
  if (mine)
  {
    const uint32_t lane = tid & 63;
    const uint32_t first = (tid - lane) * per; // the wave's first slot: its symbol = (number of s with cum[s] <= first) - 1
    uint32_t before = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      before += (uint32_t)__builtin_popcountll(__ballot((uint32_t)cum[64 * k + lane] <= first));
    const uint32_t *mk = (const uint32_t *)tab + tid * (per / 4);
    w0 = mk[0];
    w1 = per == 8 ? mk[1] : 0;
    uint32_t top = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++)
    {
      const uint32_t b = ((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xFF;
      top = b > top ? b : top;
    }
    const uint32_t incl = wave_scan_max(top);
    prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xF, 0xF, true); // wave_shr:1: the lanes before this one (lane 0: 0)
    const uint32_t base = before - 1;                                                   // cum[0] == 0 <= first: before >= 1
    prev = prev > base ? prev : base;
  }
  sync(); // every mark has been read: the table may overwrite them
  if (mine)
  {
    const uint32_t slot0 = tid * per;
    uint32_t s = prev;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++)
      if (j < per)
      {
        const uint32_t b = ((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xFF;
        s = b > s ? b : s;
        tab[slot0 + j] = make_uint2((uint32_t)cnt[s] | (s << 24), slot0 + j - (uint32_t)cum[s]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// decode table build (hist.cpp:291-306 make_dec_pack_hist, :356-384 inplace_make_hist_dec2, :308-324 the sum check)
// `tid`/`nthreads` = the threads that share this table (one wave, or the whole workgroup); SYNC() orders their LDS traffic.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool BLOCK_SYNC, bool MARKS = true> // MARKS false: the 8-byte table by a search per slot even where the marks apply (register budget of the caller)
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
  const bool marks = MARKS && MODE == kModePack64 && BLOCK_SYNC && pack64_marks(total, nthreads); // (uniform)
  for (uint32_t s = tid; s < 256; s += nthreads)
    cnt[s] = in_range ? *(const uint16_t *)(c.stream + hist_off + 2 * s) : (uint16_t)0;
  if (marks)
    pack64_zero_marks((uint2 *)c.table, total, tid, nthreads);
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
    if (marks)
      pack64_mark4((uint2 *)c.table, total, 4 * tid, excl, c0, c1, c2, c3);
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
  if (MODE == kModePack64 && marks)
    pack64_from_marks((uint2 *)c.table, cnt, cum, total, tid, nthreads, sync);
  else if (MODE == kModePack64)
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
template <int MODE, bool FULL, bool ALLWT = false> // ALLWT: every store written through (see store_u8)
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
      if (ALLWT)
        store_u32_saddr<true>((uint8_t *)uni64((uint64_t)(uintptr_t)row_base), ol.store_off, acc);
      else
        HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)row_base), ol.store_off, acc); // (uni64: the asm needs the base in an SGPR pair whatever the compiler thinks of its uniformity)
      r.vm++;
    }
    else if (act)
    {
      if (ALLWT)
        store_u32<true>(row_base + ol.store_off, acc);
      else
        HSRANS_STORE_U32((uint32_t *)(row_base + ol.store_off), acc);
    }
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
      store_u8<ALLWT>(c.out + o + p, (e >> (8 * kSymByte)) & 0xFFu);
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
  "ds_read_u8 v" #P0 ", %[g] offset:%[off]\n\t"                                                                                                      \
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

// (TABLE_OFF: the LDS address of the rank bytes, a compile-time constant that rides in the byte gather's offset field — 0 for the launches
// with one table at the start of the workgroup's LDS; k_decode_dealt's second table sits behind the first)
template <uint32_t TABLE_OFF = 0>
__device__ __forceinline__ uint32_t fast_groups4_rank(uint32_t &x, uint32_t &s_addr, const WaveCtx &c, uint32_t s_entries)
{
  uint32_t acc, t, g, st;
  asm volatile(HSRANS_FAST_GROUP_RANK(52, 53) HSRANS_FAST_GROUP_RANK(54, 55) HSRANS_FAST_GROUP_RANK(56, 57) HSRANS_FAST_GROUP_RANK(58, 59)
               "v_perm_b32 %[acc], v54, v52, %[selp]\n\t"
               "v_perm_b32 %[t], v58, v56, %[selp]\n\t"
               "v_perm_b32 %[acc], %[t], %[acc], %[selq]"
               : [x] "+v"(x), [sa] "+s"(s_addr), [acc] "=&v"(acc), [t] "=&v"(t), [g] "=&v"(g), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [sent] "s"(s_entries), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u),
                 [off] "i"(TABLE_OFF)
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

template <bool STRICT, int MODE = kModePack64, bool WT = false, uint32_t TABLE_OFF = 0>
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
      const uint32_t acc = quad_transpose(MODE == kModeRank ? fast_groups4_rank<TABLE_OFF>(x, s_addr, c, TABLE_OFF + (1u << c.bits)) : fast_groups4(x, s_addr, c, s_table), ol.sel_a, ol.sel_b);
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
template <int MODE, bool FAST = false, bool STRICT = false, bool WT = false, bool ALLWT = false, uint32_t TABLE_OFF = 0> // STRICT: the constant wait of run_groups_fast (the grouped launches); WT: write-through stores in the hand-scheduled loop; ALLWT: in what is left over too; TABLE_OFF: where the rank table sits (fast_groups4_rank)
__device__ __forceinline__ void run_groups(uint32_t &x, const StreamWin &sw, Ring &r, const WaveCtx &c, uint64_t &o, uint32_t steps)
{
  if (FAST && MODE == kModePack64 && c.S == 64 && r.mirror_lanes == 0xFFFFFFFFu)
    run_groups_fast<STRICT, kModePack64, WT>(x, sw, r, c, o, steps); // the hand-scheduled loop; leaves < 4 groups
  if (FAST && MODE == kModeRank && c.S == 64 && r.mirror_lanes == 0xFFFFFFFFu && uni(lds_address(c.table)) == TABLE_OFF)
    run_groups_fast<STRICT || TABLE_OFF == 0, kModeRank, WT, TABLE_OFF>(x, sw, r, c, o, steps); // (its rank byte's address is the slot itself + TABLE_OFF: the table at that LDS address)
  if (c.S == 64)
    run_groups_impl<MODE, true, ALLWT>(x, sw, r, c, o, steps);
  else
    run_groups_impl<MODE, false, ALLWT>(x, sw, r, c, o, steps);
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
// (both halves' word addresses are formed on all lanes and picked with a v_cndmask — 5 vector, 0 scalar instructions; forming chain B's
// under EXEC = upper half was 4 vector + 2 scalar and lost: the CU's scalar unit is the contended one in this loop)
#define HSRANS_PAIR_ADDR                                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                                                                                                           \
  "v_mbcnt_hi_u32_b32 %[w], vcc_hi, 0\n\t"                                                                                                           \
  "v_lshl_add_u32 %[t], %[t], 1, %[sa]\n\t"                                                                                                          \
  "v_lshl_add_u32 %[w], %[w], 1, %[sb]\n\t"                                                                                                          \
  "v_cndmask_b32 %[w], %[t], %[w], %[up]\n\t"
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
template <int MODE, bool ALLWT = false>
__device__ __forceinline__ void run_tail(uint32_t &x, Ring &r, const WaveCtx &c, uint64_t o, uint32_t tail)
{
  if (tail == 0)
    return;
  const uint32_t p = lane_to_byte(c.lane);
  const bool act = c.lane < c.S && p < tail;
  const uint32_t e = group_step<MODE, false>(x, r, c, __builtin_amdgcn_ballot_w64(act));
  if (act)
    store_u8<ALLWT>(c.out + o + p, (e >> ((MODE == kModePack64 || MODE == kModeRank || MODE == kModeSpill) ? 24 : 0)) & 0xFFu);
}

// single-symbol block (block_rANS32x64_16w_decode.cpp:52-60): wave-wide fill
template <bool ALLWT = false>
__device__ void wave_fill(const WaveCtx &c, uint64_t o, uint64_t len, uint32_t symbol)
{
  uint8_t *p = c.out + o;
  uint64_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head > len)
    head = len;
  if (c.lane < head)
    store_u8<ALLWT>(p + c.lane, symbol & 0xFFu);
  p += head;
  len -= head;
  const uint32_t s4 = symbol * 0x01010101u;
  const u32x4 v = {s4, s4, s4, s4};
  const uint64_t vecs = len / 16;
  for (uint64_t i = c.lane; i < vecs; i += 64)
    store_u128<ALLWT>((u32x4 *)p + i, v);
  const uint64_t done = vecs * 16;
  if (c.lane < len - done)
    store_u8<ALLWT>(p + done + c.lane, symbol & 0xFFu);
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

} // namespace hsrans

#endif // HSRANS_KERNELS_COMMON_H
