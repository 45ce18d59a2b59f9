// kernels_dealt.h — block_/mt_ plans with checkpoints, ONE round: every resident workgroup takes a host-dealt share of the plan's chains that
// touches at most two blocks — run_dealt, k_decode_dealt (round 6).
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_DEALT_H
#define HSRANS_KERNELS_DEALT_H

namespace hsrans
{

// Why another launch shape.  The grouped launch gives a workgroup a whole block per round, whatever its age: where there are about as
// many blocks as workgroup slots — a rank's 128 MiB of a sharded 2^30-byte stream in 256 KiB blocks: 512 blocks, 512 slots — the older
// workgroup of every CU is done at 39 us and the younger one at 52 (per-class stamps, profiles/r06_grouped_stamps.txt), and nothing can
// move work between them.  k_decode_spread (round 4) deals the chain list out by age-class weight instead, but only while its longest
// share is shorter than every block (so that no share needs a third table) and shorter than 128 chains (its LDS record area), and it
// pays three dependent trips to memory before its first group: piece records -> histograms -> states and stream chunks.
// Here the HOST deals (deal_shares, hsrans_kernels.hip; once per plan and weight set): workgroup b's share is chains
// [begin[b], begin[b + 1]), sized by the workgroups' age-class weights and cut back where it would reach into a third block, and
// `split[b]` says where inside the share the second block starts.  The table rides in the kernel arguments, so a wave knows its chains —
// and with them the addresses of its piece records, its start states and its workgroup's one or two histograms — from arithmetic alone:
//   trip 1 (scalar loads): the share's two histogram offsets, the wave's first / last / next piece record;
//   trip 2: the histogram counts (first in issue order: the build waits for them alone), the start states, stream chunks 0 and 1;
//   then the table build (LDS only) while trip 2's tail lands, the rest of the ring, and the decode loop of k_decode_direct.
// That is the one-chain-per-wave launch of a raw stream plus a table build.  Shares without single-symbol blocks only (the host checks;
// such plans keep k_decode_spread / k_decode_grouped), 8-byte tables, <= 11 bits (two tables beside 16 rings, two workgroups per CU).
// (DealtTable, DealtParams: hsrans_kernels.h)
typedef const __attribute__((address_space(4))) Piece *kpiece_ptr;

// build_table<kModePack64, true>'s body for counts that were requested earlier: `mine` = count of symbol `tid` (tid < 256), already in a register
__device__ __forceinline__ void dealt_build(uint2 *tab, uint16_t *cnt, uint16_t *cum, uint32_t bits, uint32_t mine, uint32_t tid, uint32_t nthreads, uint32_t *status)
{
  const uint32_t total = 1u << bits;
  const bool marks = pack64_marks(total, nthreads); // (kernels_common.h: the table without a search per slot)
  if (tid < 256)
    cnt[tid] = (uint16_t)mine;
  if (marks)
    pack64_zero_marks(tab, total, tid, nthreads);
  __syncthreads();
  if (tid < 64)
  {
    const uint32_t c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    const uint32_t sum4 = c0 + c1 + c2 + c3;
    uint32_t incl = sum4;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t up = __shfl_up(incl, d, 64);
      if (tid >= (uint32_t)d)
        incl += up;
    }
    const uint32_t excl = incl - sum4;
    cum[4 * tid] = (uint16_t)excl;
    cum[4 * tid + 1] = (uint16_t)(excl + c0);
    cum[4 * tid + 2] = (uint16_t)(excl + c0 + c1);
    cum[4 * tid + 3] = (uint16_t)(excl + c0 + c1 + c2);
    if (marks)
      pack64_mark4(tab, total, 4 * tid, excl, c0, c1, c2, c3);
    if ((uint32_t)__shfl(incl, 63, 64) != total && tid == 0) // (hist.cpp:308-324: the decoder returns 0; here: status, the host discards the output)
      atomicOr(status, kStatusBadHist);
  }
  __syncthreads();
  if (marks)
    pack64_from_marks(tab, cnt, cum, total, tid, nthreads, []() { __syncthreads(); });
  else
    for (uint32_t slot = tid; slot < total; slot += nthreads)
    {
      uint32_t s = 0;
#pragma unroll
      for (uint32_t step = 128; step >= 1; step >>= 1)
        s += ((uint32_t)cum[s + step] <= slot) ? step : 0;
      tab[slot] = make_uint2((uint32_t)cnt[s] | (s << 24), slot - (uint32_t)cum[s]);
    }
  __syncthreads();
}

template <bool WT, bool PARTS>
__device__ __forceinline__ void run_dealt(const DealtParams &dp, const DealtTable &dt, uint8_t *smem)
{
  constexpr int MODE = kModePack64;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t b = blockIdx.x;
  const uint32_t c0 = dt.begin[b], c1 = dt.begin[b + 1];
  const uint32_t count = c1 - c0;
  const uint32_t split = dt.split[b] < count ? dt.split[b] : count; // chains [c0, c0 + split) = first block, [c0 + split, c1) = second
  const bool two = split < count;
  const uint32_t N = dp.n_chains;
#if HSRANS_HAVE_STAMPS
  const uint64_t t_entry = dp.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t t_table = 0, t_ready = 0;
#endif
  WaveCtx c;
  c.stream = dp.stream;
  c.stream_len = dp.stream_len;
  c.stream_lo = dp.stream_lo;
  c.out = dp.out;
  c.out_cap = dp.out_cap;
  c.status = dp.status;
  c.bits = dp.bits;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  const uint32_t table_bytes = table_bytes_for(MODE, c.bits);
  // LDS: [waves x ring][table A][table B][build scratch: 2 x (counts + prefix sums)]
  c.rings = smem + wave * kFastRingBytes;
  uint8_t *const table0 = smem + waves * kFastRingBytes;
  uint16_t *const scratch0 = (uint16_t *)(table0 + 2 * table_bytes);
  c.gtable = nullptr;
  c.scratch_cnt = c.scratch_cum = nullptr;
  // This wave's chains: its age class's part of the workgroup's share.  A wave whose part straddles the block boundary decodes two runs,
  // and the second one costs it a second prologue (start states, first chunks: two dependent trips, ~2.5 us) — every workgroup whose
  // share has two blocks has such a wave, and the launch ends with its last wave.  So the dealing counts `gap` VIRTUAL chains at the
  // boundary (the host's estimate of that prologue in chains of this plan): the wave they fall to gets that much less real work.
  // (Dealing the two blocks to disjoint sets of waves instead — no wave straddles — was measured: some wave then gets up to 1.5x its
  // share whenever a block's part of the share is not a whole number of waves' worth: 100 MB in 256 KiB blocks 42.7 -> 51 us.)
  const uint32_t half = b >= (gridDim.x + 1) / 2 ? 1 : 0;
  const uint32_t cum_all = dp.cum[half][waves];
  const uint32_t gap = two ? dp.gap_chains : 0;
  const uint32_t vcount = count + gap;
  auto real = [&](uint32_t v) { return v < split ? v : v < split + gap ? split : v - gap; }; // virtual position -> chain of the share
  // (these two 64-bit divisions stay: the reciprocal form that pays in the grouped rounds measured 0.7 us SLOWER here, 41.0 against 40.4 us)
  const uint32_t first = c0 + real((uint32_t)((uint64_t)dp.cum[half][wave] * vcount / cum_all));
  const uint32_t last = c0 + real((uint32_t)((uint64_t)dp.cum[half][wave + 1] * vcount / cum_all));
  const uint32_t mid = c0 + split; // first chain of the second block
  // run 1 = [first, e1) in one block; run 2 = [e1, last) in the second block when the wave's chains straddle `mid`
  const uint32_t e1 = first < mid && mid < last ? mid : last;
  const bool have = first < last;
  // ---- trip 1: everything whose address is arithmetic -------------------------------------------------------------------------
  const kpiece_ptr pa = (kpiece_ptr)(uintptr_t)(dp.pieces + c0);
  const kpiece_ptr pb = (kpiece_ptr)(uintptr_t)(dp.pieces + (two ? mid : c0));
  const uint64_t hist_a = count ? pa->hist_off : 0, hist_b = count ? pb->hist_off : 0;
  uint64_t words = 0, out0 = 0, limit = 0, out_end = 0;
  uint32_t tail = 0;
  if (have)
  {
    const kpiece_ptr p0 = (kpiece_ptr)(uintptr_t)(dp.pieces + first);
    const kpiece_ptr p1 = (kpiece_ptr)(uintptr_t)(dp.pieces + (e1 - 1));
    const kpiece_ptr pn = (kpiece_ptr)(uintptr_t)(dp.pieces + (e1 < N ? e1 : e1 - 1));
    words = p0->words_off;
    out0 = p0->out_off;
    out_end = p1->out_off + (uint64_t)p1->steps * 64;
    tail = p1->tail;
    // the first stream byte the run cannot need: the next chain's cursor when that continues the block, else the next block's histogram
    // (a block's words end before the next block's header), else the end of the stream
    limit = e1 >= N ? c.stream_len : pn->hist_off != p0->hist_off ? pn->hist_off : pn->words_off; // (e1 == mid: the second block's histogram)
  }
  // ---- trip 2: the counts first (the build waits for them alone), then the start states and the first chunks --------------------
  const uint32_t half_threads = blockDim.x / 2;
  const uint32_t side = two && threadIdx.x >= half_threads ? 1 : 0;
  const uint32_t btid = two ? threadIdx.x - side * half_threads : threadIdx.x; // thread of this side's builder
  const uint32_t bthreads = two ? half_threads : blockDim.x;
  const uint64_t my_hist = side ? hist_b : hist_a;
  uint32_t my_count = 0;
  const bool loads_count = btid < 256 && count != 0 && HSRANS_HIST_IN_RANGE(c, my_hist);
  if (btid < 256 && count != 0 && !HSRANS_HIST_IN_RANGE(c, my_hist) && btid == 0)
    atomicOr(c.status, kStatusOutOfRange);
  // (ordinary loads, so that the compiler knows these registers are in flight — an asm load's result register is free game for a move
  // long before the data lands; the ring's requests are asm and unknown to it, which only makes its waits stricter than it thinks)
  if (loads_count)
    my_count = *(const uint16_t *)(c.stream + my_hist + 2 * btid);
  uint32_t x = 0;
  StreamWin sw;
  Ring r;
  ring_bind(r, c.rings, 9, true);
  if (have)
  {
    x = dp.states[(uint64_t)first * 64 + c.lane];
    win_open(sw, c, words, limit);
    ring_begin(sw, r, c, words, true, true); // chunks 0 and 1
  }
  if (count != 0) // (the same in every wave of the workgroup)
    dealt_build((uint2 *)(table0 + side * table_bytes), scratch0 + side * 512, scratch0 + side * 512 + 256, c.bits, my_count, btid, bthreads, c.status);
#if HSRANS_HAVE_STAMPS
  if (dp.stamps)
    t_table = __builtin_amdgcn_s_memrealtime();
#endif
  if (have)
  {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory"); // states and chunks 0, 1 (the count came first: long done)
    ring_begin_rest(sw, r, c);
#if HSRANS_HAVE_STAMPS
    if (dp.stamps)
      t_ready = __builtin_amdgcn_s_memrealtime();
#endif
    c.table = table0 + (two && first >= mid ? table_bytes : 0);
    c.table_b = c.table;
    uint64_t o = out0;
    static_assert(WT || !PARTS, "a launch that publishes completion words writes everything through");
    run_groups<MODE, true, false, WT, PARTS>(x, sw, r, c, o, (uint32_t)((out_end - o) / 64));
    run_tail<MODE, PARTS>(x, r, c, o, tail);
    if (e1 < last) // the wave's chains go on in the share's second block: a chain of its own (its start states, its cursor, table B)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no request of run 1 may still land in the ring run 2 begins
      const kpiece_ptr q0 = (kpiece_ptr)(uintptr_t)(dp.pieces + e1);
      const kpiece_ptr q1 = (kpiece_ptr)(uintptr_t)(dp.pieces + (last - 1));
      const kpiece_ptr qn = (kpiece_ptr)(uintptr_t)(dp.pieces + (last < N ? last : last - 1));
      const uint64_t w2 = q0->words_off;
      const uint64_t lim2 = last >= N ? c.stream_len : qn->hist_off != q0->hist_off ? qn->hist_off : qn->words_off;
      uint32_t x2 = dp.states[(uint64_t)e1 * 64 + c.lane];
      win_open(sw, c, w2, lim2);
      ring_begin(sw, r, c, w2);
      ring_ready(x2);
      c.table = table0 + table_bytes;
      c.table_b = c.table;
      uint64_t o2 = q0->out_off;
      const uint64_t end2 = q1->out_off + (uint64_t)q1->steps * 64;
      run_groups<MODE, true, true, WT, PARTS>(x2, sw, r, c, o2, (uint32_t)((end2 - o2) / 64));
      run_tail<MODE, PARTS>(x2, r, c, o2, q1->tail);
    }
  }
#if HSRANS_HAVE_STAMPS
  if (dp.stamps && c.lane == 0)
  {
    uint64_t *st = dp.stamps + (uint64_t)(b * waves + wave) * 8;
    st[0] = t_entry;
    st[1] = t_table;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = st[3];
    st[5] = 0;
    st[6] = 0;
    st[7] = 0;
  }
#endif
  if (PARTS && dp.parts.n != 0) // a rank's sub-runs in this one launch: the workgroup counts itself into every sub-run its share overlaps (kernels_grouped.h)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && count != 0)
    {
      uint32_t lo = 0, hi = 0, begin = 0;
      bool any = false;
      for (uint32_t p = 0; p < dp.parts.n; p++)
      {
        const uint32_t end = dp.parts.chain_end[p];
        if (end > begin && c0 < end && c1 > begin)
        {
          lo = any ? lo : p;
          hi = p;
          any = true;
        }
        begin = end > begin ? end : begin;
      }
      if (any)
        part_signal(dp.parts, lo, hi);
    }
  }
}

// build_table<kModeRank, true>'s body for counts already in cnt[0 .. 255] (as dealt_build for the 8-byte table): the slot -> symbol bytes, every
// thread a contiguous run of dwords (one search for its first slot, then the symbol only moves forward), and the 256 eight-byte entries by
// symbol value behind them
__device__ __forceinline__ void rank_build_from_counts(uint8_t *table, uint16_t *cnt, uint16_t *cum, uint32_t bits, uint32_t tid, uint32_t nthreads, uint32_t *status)
{
  const uint32_t total = 1u << bits;
  __syncthreads();
  if (tid < 64)
  {
    const uint32_t c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    const uint32_t sum4 = c0 + c1 + c2 + c3;
    uint32_t incl = sum4;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t up = __shfl_up(incl, d, 64);
      if (tid >= (uint32_t)d)
        incl += up;
    }
    const uint32_t excl = incl - sum4;
    cum[4 * tid] = (uint16_t)excl;
    cum[4 * tid + 1] = (uint16_t)(excl + c0);
    cum[4 * tid + 2] = (uint16_t)(excl + c0 + c1);
    cum[4 * tid + 3] = (uint16_t)(excl + c0 + c1 + c2);
    if ((uint32_t)__shfl(incl, 63, 64) != total && tid == 0)
      atomicOr(status, kStatusBadHist);
  }
  __syncthreads();
  uint32_t *sym4 = (uint32_t *)table;
  const uint32_t dwords = total / 4;
  const uint32_t per = (dwords + nthreads - 1) / nthreads;
  const uint32_t q0 = tid * per, q1 = q0 + per < dwords ? q0 + per : dwords;
  if (q0 < q1)
  {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t step = 128; step >= 1; step >>= 1)
      s += ((uint32_t)cum[s + step] <= 4 * q0) ? step : 0;
    uint32_t next = s < 255 ? (uint32_t)cum[s + 1] : 0x10000u;
    for (uint32_t q = q0; q < q1; q++)
    {
      uint32_t packed = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
      {
        const uint32_t slot = 4 * q + k;
        while (next <= slot)
        {
          s++;
          next = s < 255 ? (uint32_t)cum[s + 1] : 0x10000u;
        }
        packed |= s << (8 * k);
      }
      sym4[q] = packed;
    }
  }
  uint2 *ent = (uint2 *)(table + total);
  for (uint32_t s = tid; s < 256; s += nthreads)
    ent[s] = make_uint2((uint32_t)cnt[s] | (s << 24), 0u - (uint32_t)cum[s]);
  __syncthreads();
}

// The same launch for 13- and 14-bit histograms: two RANK tables per workgroup (kModeRank: a byte per slot + 256 eight-byte entries; 10 / 18 KiB
// each, where the 8-byte-per-slot table is 64 / 128 KiB).  The hand-scheduled rank loop takes the slot as the address of its rank byte, so the
// first table sits at LDS address 0 and the second one's address rides in the byte gather's offset field — a compile-time constant, hence a
// kernel per width.  LDS: [table A][table B][build scratch 2 KiB][16 rings] = 79,872 bytes at 14 bits: two workgroups per CU.
// The prologue is the plain one (piece records + start states, then the builder's own count load, then the ring): the table build is the
// longer part at these widths anyway.
template <uint32_t BITS, bool PARTS>
__device__ __forceinline__ void run_dealt_rank(const DealtParams &dp, const DealtTable &dt, uint8_t *smem)
{
  constexpr int MODE = kModeRank;
  constexpr uint32_t TB = table_bytes_for(kModeRank, BITS); // = the second table's LDS address
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t b = blockIdx.x;
  const uint32_t c0 = dt.begin[b], c1 = dt.begin[b + 1];
  const uint32_t count = c1 - c0;
  const uint32_t split = dt.split[b] < count ? dt.split[b] : count;
  const bool two = split < count;
  const uint32_t N = dp.n_chains;
  WaveCtx c;
  c.stream = dp.stream;
  c.stream_len = dp.stream_len;
  c.stream_lo = dp.stream_lo;
  c.out = dp.out;
  c.out_cap = dp.out_cap;
  c.status = dp.status;
  c.bits = BITS;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << BITS) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(BITS));
  if (uni(lds_address(smem)) != 0 || dp.bits != BITS) // (the rank loop's addressing; the launcher picks the kernel by the plan's width)
  {
    if (threadIdx.x == 0)
      atomicOr(c.status, kStatusOutOfRange);
    return;
  }
  uint8_t *const table0 = smem;
  uint16_t *const scratch0 = (uint16_t *)(smem + 2 * TB);
  c.rings = smem + 2 * TB + 2048 + wave * kFastRingBytes;
  c.gtable = nullptr;
  const uint32_t half = b >= (gridDim.x + 1) / 2 ? 1 : 0;
  const uint32_t cum_all = dp.cum[half][waves];
  const uint32_t gap = two ? dp.gap_chains : 0;
  const uint32_t vcount = count + gap;
  auto real = [&](uint32_t v) { return v < split ? v : v < split + gap ? split : v - gap; };
  // (these two 64-bit divisions stay: the reciprocal form that pays in the grouped rounds measured 0.7 us SLOWER here, 41.0 against 40.4 us)
  const uint32_t first = c0 + real((uint32_t)((uint64_t)dp.cum[half][wave] * vcount / cum_all));
  const uint32_t last = c0 + real((uint32_t)((uint64_t)dp.cum[half][wave + 1] * vcount / cum_all));
  const uint32_t mid = c0 + split;
  const kpiece_ptr pa = (kpiece_ptr)(uintptr_t)(dp.pieces + c0);
  const kpiece_ptr pb = (kpiece_ptr)(uintptr_t)(dp.pieces + (two ? mid : c0));
  const uint64_t hist_a = count ? pa->hist_off : 0, hist_b = count ? pb->hist_off : 0;
  // trip 2 as in run_dealt: the counts first, then the start states and chunks 0 and 1 of the wave's first run; the two tables are built side
  // by side (half the workgroup each) while the chunks are in flight
  const uint32_t half_threads = blockDim.x / 2;
  const uint32_t side = two && threadIdx.x >= half_threads ? 1 : 0;
  const uint32_t btid = two ? threadIdx.x - side * half_threads : threadIdx.x;
  const uint32_t bthreads = two ? half_threads : blockDim.x;
  const uint64_t my_hist = side ? hist_b : hist_a;
  uint32_t my_count = 0;
  if (btid < 256 && count != 0)
  {
    if (HSRANS_HIST_IN_RANGE(c, my_hist))
      my_count = *(const uint16_t *)(c.stream + my_hist + 2 * btid);
    else if (btid == 0)
      atomicOr(c.status, kStatusOutOfRange);
  }
  const uint32_t f0 = first, e0 = first < mid && mid < last ? mid : last;
  StreamWin sw;
  Ring r;
  ring_bind(r, c.rings, 9, true);
  uint32_t x = 0;
  uint64_t words0 = 0, limit0 = 0;
  if (f0 < e0)
  {
    const kpiece_ptr p0 = (kpiece_ptr)(uintptr_t)(dp.pieces + f0);
    const kpiece_ptr pn = (kpiece_ptr)(uintptr_t)(dp.pieces + (e0 < N ? e0 : e0 - 1));
    words0 = p0->words_off;
    limit0 = e0 >= N ? c.stream_len : pn->hist_off != p0->hist_off ? pn->hist_off : pn->words_off;
    x = dp.states[(uint64_t)f0 * 64 + c.lane];
  }
  if (count != 0 && btid < 256)
    (scratch0 + side * 512)[btid] = (uint16_t)my_count;
  if (f0 < e0)
  {
    win_open(sw, c, words0, limit0);
    ring_begin(sw, r, c, words0, true, true);
  }
  if (count != 0)
    rank_build_from_counts(table0 + side * TB, scratch0 + side * 512, scratch0 + side * 512 + 256, BITS, btid, bthreads, c.status);
  // the wave's chains: one run, or two where they straddle the share's block boundary
  for (uint32_t run = 0; run < 2; run++)
  {
    const uint32_t f = run == 0 ? f0 : e0;
    const uint32_t e = run == 0 ? e0 : last;
    if (f >= e)
      continue;
    const kpiece_ptr p0 = (kpiece_ptr)(uintptr_t)(dp.pieces + f);
    const kpiece_ptr p1 = (kpiece_ptr)(uintptr_t)(dp.pieces + (e - 1));
    if (run == 0)
    {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory"); // start states, chunks 0 and 1
      ring_begin_rest(sw, r, c);
    }
    else
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no request of run 1 may still land in the ring run 2 begins
      const kpiece_ptr pn = (kpiece_ptr)(uintptr_t)(dp.pieces + (e < N ? e : e - 1));
      const uint64_t words = p0->words_off;
      const uint64_t limit = e >= N ? c.stream_len : pn->hist_off != p0->hist_off ? pn->hist_off : pn->words_off;
      x = dp.states[(uint64_t)f * 64 + c.lane];
      win_open(sw, c, words, limit);
      ring_begin(sw, r, c, words);
      ring_ready(x);
    }
    uint64_t o = p0->out_off;
    const uint64_t out_end = p1->out_off + (uint64_t)p1->steps * 64;
    if (two && f >= mid)
    {
      c.table = table0 + TB;
      c.table_b = c.table;
      run_groups<MODE, true, true, true, PARTS, TB>(x, sw, r, c, o, (uint32_t)((out_end - o) / 64));
    }
    else
    {
      c.table = table0;
      c.table_b = c.table;
      run_groups<MODE, true, true, true, PARTS, 0>(x, sw, r, c, o, (uint32_t)((out_end - o) / 64));
    }
    run_tail<MODE, PARTS>(x, r, c, o, p1->tail);
  }
  if (PARTS && dp.parts.n != 0)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && count != 0)
    {
      uint32_t lo = 0, hi = 0, begin = 0;
      bool any = false;
      for (uint32_t p = 0; p < dp.parts.n; p++)
      {
        const uint32_t end = dp.parts.chain_end[p];
        if (end > begin && c0 < end && c1 > begin)
        {
          lo = any ? lo : p;
          hi = p;
          any = true;
        }
        begin = end > begin ? end : begin;
      }
      if (any)
        part_signal(dp.parts, lo, hi);
    }
  }
}

template <uint32_t BITS, bool PARTS>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_dealt_rank(DealtParams dp, DealtTable dt)
{
  extern __shared__ u32x4 smem_v[];
  run_dealt_rank<BITS, PARTS>(dp, dt, (uint8_t *)smem_v);
}

template <bool WT, bool PARTS>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_dealt(DealtParams dp, DealtTable dt)
{
  extern __shared__ u32x4 smem_v[];
  run_dealt<WT, PARTS>(dp, dt, (uint8_t *)smem_v);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_DEALT_H
