// kernels_spread.h — block_/mt_ plans with checkpoints and FEW, LARGE blocks: the chains dealt out evenly over every resident wave,
// a workgroup building the (at most two) tables its share touches — run_spread, k_decode_spread.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_SPREAD_H
#define HSRANS_KERNELS_SPREAD_H

namespace hsrans
{

// The grouped launch gives a workgroup one block (or one part of one) at a time.  With fewer blocks than a few per CU that cannot
// come out even: 100 MB in 256 KiB blocks are 382 blocks of 128 chains for 512 workgroup slots — a block per workgroup leaves a
// quarter of the wave slots empty and half the CUs with twice the work of the others, halves by ticket send half the workgroups
// through two rounds (57 and 51 us; the one-chain-per-wave launch of a raw stream of the same size: 40).
// Here the plan's N chains are ONE list, dealt out over every resident wave in proportion to its age class's weight exactly like
// the one-chain-per-wave launch of a raw stream (the first half of the grid is resident first and decodes faster: its workgroups
// get 1.7x the chains of the second half's) — one share per resident workgroup, one launch round.  A share is
// shorter than a block (the host checks: every coded block but the last has more chains than the longest share), so it touches at most two
// blocks: the workgroup builds the table of its first and of its last coded chain, side by side, and a wave decodes its chains as
// maximal runs inside one block (fill chains — single-symbol blocks — one by one), pointing c.table at the run's table.
// Chain c is piece c with start states c (single-piece chains: n_pieces == n_chains), so nothing but the piece records is read.
// MODE: kModePack64 only (the rank table of 13-15 bits sits at LDS address 0 by construction: there is no second one).
// PARTS (round 6): the launch decodes a rank's sub-runs of a sharded decode; a workgroup that is done counts itself into every sub-run
// its share of the chains overlaps (kernels_grouped.h part_signal; the host counted the same overlaps for PartArgs::target).
template <int MODE, bool PARTS = false>
__device__ void run_spread(WaveCtx &c, const PlanView &pv, const KParams &kp, uint32_t waves, uint32_t wave)
{
  const uint32_t N = kp.pa.n_chains; // (from the launcher: reading the plan's header here is a round trip to memory before anything can start)
  // a workgroup's share is the sum of its waves' age-class weights (the first half of the grid is resident first: spread_share_begin)
  const uint32_t c0 = spread_share_begin(N, blockIdx.x, gridDim.x, kp.group_cum[0][waves], kp.group_cum[1][waves]);
  const uint32_t c1 = spread_share_begin(N, blockIdx.x + 1, gridDim.x, kp.group_cum[0][waves], kp.group_cum[1][waves]);
  const uint32_t count = c1 - c0;           // <= kSpreadMaxShare (the launcher checks)
  // (with PARTS every workgroup must reach the count at the end whatever happens: a completion word nobody publishes leaves the
  // exchange's stream waiting for ever)
  auto count_into_parts = [&]() {
    if (!PARTS || kp.parts.n == 0)
      return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && c1 > c0)
    {
      uint32_t lo = 0, hi = 0, begin = 0;
      bool any = false;
      for (uint32_t p = 0; p < kp.parts.n; p++)
      {
        const uint32_t end = kp.parts.chain_end[p];
        if (end > begin && c0 < end && c1 > begin) // part p = chains [begin, end)
        {
          lo = any ? lo : p;
          hi = p;
          any = true;
        }
        begin = end > begin ? end : begin;
      }
      if (any)
        part_signal(kp.parts, lo, hi);
    }
  };
  if (count > kSpreadMaxShare) // (the launcher checks the same weights it hands the kernel; a share that would overrun the LDS record area must never run)
  {
    if (threadIdx.x == 0)
      atomicOr(c.status, kStatusOutOfRange);
    count_into_parts();
    return;
  }
  const uint32_t n_rec = count + (c1 < N);  // + the chain behind the share: where the last run's words end
  uint8_t *const table0 = c.table;
  const uint32_t table_bytes = table_bytes_for(MODE, c.bits);
  // ONE trip to the plan: the share's piece records into LDS, 16 bytes per thread; everything below reads them there (walking
  // them in global memory — a dependent load per chain to find where a run ends — cost 13 us on cold caches)
  Piece *lp = (Piece *)(table0 + 2 * table_bytes);
  {
    const uint4 *src = (const uint4 *)(pv.pieces + c0);
    uint4 *dst = (uint4 *)lp;
    for (uint32_t u = threadIdx.x; u < n_rec * (uint32_t)(sizeof(Piece) / 16); u += blockDim.x)
      dst[u] = src[u];
  }
  __syncthreads();
  // first and last coded (non-fill) chain of the share: their blocks' tables are the (at most two) tables the share needs
  const bool cod0 = c.lane < count && !(lp[c.lane].flags & kPieceFill);
  const bool cod1 = c.lane + 64 < count && !(lp[c.lane + 64 < n_rec ? c.lane + 64 : 0].flags & kPieceFill);
  const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cod0), m1 = __builtin_amdgcn_ballot_w64(cod1);
  const bool coded = (m0 | m1) != 0;
  const uint32_t lo = !coded ? 0 : m0 ? (uint32_t)__builtin_ctzll(m0) : 64 + (uint32_t)__builtin_ctzll(m1);
  const uint32_t hi = !coded ? 0 : m1 ? 127 - (uint32_t)__builtin_clzll(m1) : 63 - (uint32_t)__builtin_clzll(m0);
  const uint64_t hist_a = coded ? uni64(lp[lo].hist_off) : 0, hist_b = coded ? uni64(lp[hi].hist_off) : 0;
  // the wave's chains: its age class's share of the workgroup's (kp.group_cum: the weights of the one-chain-per-wave launch)
  const uint32_t half = blockIdx.x >= (gridDim.x + 1) / 2 ? 1 : 0;
  const uint32_t cum_all = kp.group_cum[half][waves];
  const Recip by = recip_of(uni(cum_all));
  const uint32_t first = share_of(kp.group_cum[half][wave], count, by);
  const uint32_t last = share_of(kp.group_cum[half][wave + 1], count, by);
  // a run: chains [i, e) of one block, back to back in stream and output; `limit` = the first stream byte it cannot need: the next
  // chain's cursor when that continues the block, else the next block's histogram (a block's words end before the next block's
  // header), else the end of the stream
  auto run_from = [&](uint32_t i, uint32_t &e, uint64_t &limit) {
    const uint64_t hist = uni64(lp[i].hist_off);
    e = i + 1;
    while (e < last && !(uni(lp[e].flags) & kPieceFill) && uni64(lp[e].hist_off) == hist)
      e++;
    limit = c.stream_len;
    if (e < n_rec && !(uni(lp[e].flags) & kPieceFill))
      limit = uni64(lp[e].hist_off) == hist ? uni64(lp[e].words_off) : uni64(lp[e].hist_off);
  };
  if (coded)
  {
    // two tables: one half of the workgroup builds each, side by side (the builder's barriers are the same in number for both);
    // its scratch sits in a ring (none has a request in flight yet): wave 0's for the first half, wave waves/2's for the second
    const bool two = hist_a != hist_b;
    const uint32_t half_threads = blockDim.x / 2;
    const uint32_t side = two && threadIdx.x >= half_threads ? 1 : 0;
    c.table = table0 + side * table_bytes;
    c.table_b = c.table;
    c.scratch_cnt = (uint16_t *)(c.rings - wave * kFastRingBytes + side * (waves / 2) * kFastRingBytes);
    c.scratch_cum = c.scratch_cnt + 256;
    build_table<MODE, true>(c, side ? hist_b : hist_a, two ? threadIdx.x - side * half_threads : threadIdx.x, two ? half_threads : blockDim.x);
  }
  uint32_t i = first; // (indices into the share)
  while (i < last)
  {
    const Piece *p0 = lp + i;
    if (uni(p0->flags) & kPieceFill)
    {
      wave_fill<PARTS>(c, uni64(p0->out_off), uni64(p0->fill_len), (uint32_t)uni64(p0->hist_off) & 0xFF);
      i++;
      continue;
    }
    const uint64_t hist = uni64(p0->hist_off);
    uint32_t e;
    uint64_t limit;
    run_from(i, e, limit);
    if (hist != hist_a && hist != hist_b) // a third block inside one share: the launcher's check should have kept this plan away
    {
      if (c.lane == 0)
        atomicOr(c.status, kStatusOutOfRange);
      i = e;
      continue;
    }
    c.table = table0 + (hist == hist_a ? 0 : table_bytes);
    c.table_b = c.table;
    const Piece *p1 = lp + (e - 1);
    // (requesting the first run's states and chunks BEFORE the table build, to land meanwhile, was measured: 49.3 against 47.8 us —
    // the builder's own loads queue behind them)
    StreamWin sw;
    Ring r;
    uint32_t x = c.lane < c.S ? pv.states[(uint64_t)(c0 + i) * c.S + c.lane] : 0;
    ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
    win_open(sw, c, uni64(p0->words_off), limit);
    ring_begin(sw, r, c, uni64(p0->words_off));
    uint64_t o = uni64(p0->out_off);
    const uint64_t run_steps = groups_of(c.S, uni64(p1->out_off) - o) + uni(p1->steps);
    const uint32_t run_tail_syms = uni(p1->tail);
    ring_ready(x);
    run_groups<MODE, true, true, PARTS, PARTS>(x, sw, r, c, o, (uint32_t)run_steps); // (write-through stores, counted waits: no difference here)
    run_tail<MODE, PARTS>(x, r, c, o, run_tail_syms);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no stream request of this run may still land in the ring the next one begins
    i = e;
  }
  count_into_parts();
}

// LDS: [waves x ring][table A][table B][the share's piece records, (kSpreadMaxShare + 1) x 48 B]
template <int MODE, bool PARTS = false>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_spread(KParams kp)
{
  static_assert(MODE == kModePack64, "two tables side by side: not for the modes whose table sits at LDS address 0");
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  // the plan's geometry comes with the launch (kp.pa.n_chains / S / bits): nothing waits for the plan's header
  PlanView pv;
  pv.hdr = (const PlanHeader *)kp.plan;
  pv.chain_first = (const uint32_t *)(kp.plan + plan_chain_first_off());
  pv.pieces = (const Piece *)(kp.plan + plan_pieces_off(kp.pa.n_chains));
  pv.states = (const uint32_t *)(kp.plan + plan_states_off(kp.pa.n_chains, kp.pa.n_chains));
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
  c.gtable = nullptr;
  c.scratch_cnt = nullptr; // (run_spread points the builder at a ring)
  c.scratch_cum = nullptr;
  run_spread<MODE, PARTS>(c, pv, kp, waves, wave);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_SPREAD_H
