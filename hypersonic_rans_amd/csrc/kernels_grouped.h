// kernels_grouped.h — block_/mt_ plans with checkpoints: one workgroup per block, round after round (BASELINE config 4) — run_grouped, k_decode_grouped.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_GROUPED_H
#define HSRANS_KERNELS_GROUPED_H

namespace hsrans
{

// Grouped launch: workgroup b walks groups b, b + gridDim.x, ...; per group one table build, then every wave decodes an
// equal contiguous share of the group's chains.
// LEAN: the host promises a 64-state plan whose groups are all mergeable runs or fills (every block_/mt_ stream with checkpoints
// this library's encoders or index builders make): the 32-state pair path and the general chain runner are left out of the
// kernel, which is what keeps it at 8 waves per SIMD
// BATCH (round 5, k_decode_grouped_batch): the group list spans several member streams; bits 16.. of Group::flags name the group's
// member, whose plan arrays, stream, output and status word are picked up at the start of every round (a handful of scalar loads
// beside the group record's own): many small block_/mt_ streams then share the rounds of ONE launch instead of each launching a
// mostly empty device (the reference's pool of per-block tasks over several files: mt_rANS32x64_16w_decode.cpp:182-224, main.cpp:841-898).
// PARTS (round 6, k_decode_grouped<MODE, true, true>): the launch decodes a rank's sub-runs of a sharded decode and publishes a completion
// word per sub-run (hsrans_kernels.h PartArgs; hsrans_comm.cpp).  A group carries the sub-runs it overlaps in Group::flags; when the
// workgroup has passed the barrier that ends a round — every wave has waited for its stores first — its first lane makes them visible
// device-wide and counts the group into those sub-runs.  Nothing is added to the decode loop: the barrier and the wait are the round's own.
// Visibility: every decoded byte of such a launch is WRITTEN THROUGH (sc0 sc1: the hand-scheduled loop's stores and, ALLWT, everything off
// it), and a written-through store counts down vmcnt only when memory has it; so "every wave waited for vmcnt(0), then the barrier" means
// the group's bytes are in memory, and the counting below needs no cache maintenance.  (First version: plain `nt` stores and a device-scope
// release fence here.  The fence is buffer_wbl2 — a walk of the XCD's whole L2, serialised among the 128 workgroups that share it: 2,048
// groups of a 2^30-byte stream took 396 us instead of 204.)  The atomics are relaxed on purpose: an acquire / release at device scope
// would bring the cache walk back.
__device__ __forceinline__ void part_signal(const PartArgs &pa, uint32_t lo, uint32_t hi)
{
  for (uint32_t p = lo; p <= hi && p < pa.n; p++)
  {
    const uint32_t seen = __hip_atomic_fetch_add(pa.count + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (seen == pa.target[p]) // the part's last unit of this launch: every other unit's bytes were in memory before its increment
      __hip_atomic_store(pa.done[p], pa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int MODE, bool LEAN = false, bool FAST = false, bool BATCH = false, bool PARTS = false> // FAST: the hand-scheduled 32-state pair loop too (measured in a kernel of its own: 71 VGPRs, 7 waves per SIMD — not used)
__device__ void run_grouped(const WaveCtx &c_in, const PlanView &pv_in, const KParams &kp, uint32_t waves, uint32_t wave, const BatchGroupParams *bg = nullptr)
{
  WaveCtx c = c_in;   // (BATCH: stream / output / status change with the group's member; otherwise these are the caller's, unchanged)
  PlanView pv = pv_in;
  // the launch's group list and shares: kernel arguments either way (read through the scalar cache; a local copy of the share table
  // would be indexed dynamically and land in scratch)
  const Group *const groups = BATCH ? bg->groups : kp.groups;
  const uint32_t n_groups = BATCH ? bg->n_groups : kp.n_groups;
  unsigned long long *const group_tickets = BATCH ? bg->tickets : kp.group_tickets;
  const uint32_t group_prio = BATCH ? bg->group_prio : kp.group_prio;
  const uint16_t(*const group_cum)[17] = BATCH ? bg->group_cum : kp.group_cum;
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
  // (Round 5, the same idea in full: every wave, once its own share of a round is decoded, reads the NEXT group's record — published by
  // wave 0 through a third LDS word — fetches its piece record, start states and its thread's histogram count and starts its ring
  // before the round's barrier; the table build then reads nothing from memory.  Correct (the mt_/block_/batch suites pass), and
  // slower: 100 MB in 64 KiB blocks, G = 16 / 32 / 64 / 128: 56.0 / 50.8 / 49.9 / 49.8 us against 53.8 / 49.1 / 47.9 / 47.9 without.
  // The barrier wait grows by what the early fetch takes (0.75 -> 1.8 us a round), the decode gets that much company less.  Removed.)
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
  const bool dynamic = group_tickets != nullptr && n_groups > gridDim.x;
  // (the shares' divisors, once per launch instead of two 64-bit divisions per round: kernels_common.h share_of)
  const uint32_t half = blockIdx.x >= (gridDim.x + 1) / 2 ? 1 : 0;
  const Recip by_weights = recip_of(uni(group_cum[half][waves])), by_waves = recip_of(waves);
  volatile uint32_t *lds_next = (volatile uint32_t *)(c.table + table_bytes_for(MODE, c.bits)); // 2 words: launch_shape reserves 64 bytes behind the table
  uint32_t gi = blockIdx.x;
  uint32_t prev_parts = 0xFFFFFFFFu; // PARTS: the sub-runs (first | last << 8) of the group this workgroup decoded in the previous round, not yet counted
  for (uint32_t round = 0;; round++)
  {
    if (!(dynamic && round >= 1) && gi >= n_groups) // (dynamic rounds: decided below, from the published group)
    {
      if (PARTS && prev_parts != 0xFFFFFFFFu) // the last group of a statically ordered workgroup
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
          part_signal(kp.parts, prev_parts & 0xFFu, prev_parts >> 8);
      }
      break;
    }
    // `advance` runs at the end of the round (every path of the loop body ends in it)
    auto advance = [&]() {
      if (!dynamic)
        gi += gridDim.x;
      else if (wave == 0 && c.lane == 0)
      {
        const uint32_t j = (uint32_t)(atomicAdd(group_tickets, 1ull) % n_groups);
        lds_next[(round + 1) & 1] = j < n_groups - gridDim.x ? gridDim.x + j : 0xFFFFFFFFu;
      }
    };
    HSRANS_GS(const uint64_t t0 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no stream request of the previous group may still land in the scratch slot
    __syncthreads();                                    // every wave is done with the previous group's table and rings
    if (PARTS && prev_parts != 0xFFFFFFFFu)
    {
      if (threadIdx.x == 0)
        part_signal(kp.parts, prev_parts & 0xFFu, prev_parts >> 8);
      prev_parts = 0xFFFFFFFFu;
    }
    if (dynamic && round >= 1)
    {
      gi = uni(lds_next[round & 1]);
      if (gi >= n_groups) // (the same in every wave)
        break;
    }
    const Group *G = groups + gi;
    const uint32_t begin = uni(G->begin), count = uni(G->count), flags = uni(G->flags);
    if (PARTS)
      prev_parts = flags >> kGroupPartShift;
    if (BATCH)
    {
      const uint32_t member = flags >> kGroupMemberShift;
      const __attribute__((address_space(4))) GroupMember *mp = (const __attribute__((address_space(4))) GroupMember *)(uintptr_t)(bg->members + member);
      pv.chain_first = mp->chain_first;
      pv.pieces = mp->pieces;
      pv.states = mp->states;
      c.status = mp->status;
      c.stream = bg->io[member].stream;
      c.stream_len = bg->io[member].stream_len;
      c.out = bg->io[member].out;
      c.out_cap = bg->io[member].out_cap;
    }
    // mergeable groups: chain `begin + i` is piece `piece0 + i` and its start states are states[begin + i] (the host checks this
    // when it marks a group mergeable), so a wave's records come straight from the group record: one level of loads, not three
    const uint32_t piece0 = uni(G->piece0);
    HSRANS_GS(const uint64_t t1 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
    // (Requesting the wave's piece records, start states and first stream chunks BEFORE the table build, so that their ~4.5 us of
    // dependent round trips overlap its ~3 us, was built in round 3 and re-measured in round 4 where workgroup slots are empty (100 MB
    // in 256 KiB - 4 MiB blocks, 11 / 13 / 15 bits): +1 us everywhere but one case inside the noise.  Removed.)
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
    // age-class weights only where a wave gets enough chains for them to mean something (else an even split)
    const bool weighted = count >= 8 * waves;
    const Recip &by = weighted ? by_weights : by_waves;
    const uint32_t first = begin + share_of(weighted ? group_cum[half][wave] : wave, count, by);
    const uint32_t last = begin + share_of(weighted ? group_cum[half][wave + 1] : wave + 1, count, by);
    // the wave's run of a mergeable 64-state group: chains [first, last) as one chain
    StreamWin sw;
    Ring r;
    uint32_t x = 0, run_tail_syms = 0;
    uint64_t o = 0, run_steps = 0;
    auto open_run = [&]() {
      const Piece *p0 = pv.pieces + (piece0 + (first - begin));
      const Piece *p1 = pv.pieces + (piece0 + (last - 1 - begin));
      const uint64_t limit = last < begin + count ? uni64(pv.pieces[piece0 + (last - begin)].words_off) : uni64(G->words_end);
      x = c.lane < c.S ? pv.states[(uint64_t)first * c.S + c.lane] : 0;
      ring_bind(r, c.rings, 9, fast_ring_mode(MODE));
      win_open(sw, c, uni64(p0->words_off), limit);
      ring_begin(sw, r, c, uni64(p0->words_off));
      o = uni64(p0->out_off);
      run_steps = groups_of(c.S, uni64(p1->out_off) - o) + uni(p1->steps);
      run_tail_syms = uni(p1->tail);
    };
    if (!(flags & kGroupFill)) // (one call site: every inlined copy of the builder costs the kernel registers)
      build_table<MODE, true, LEAN>(c, uni64(G->hist_off), threadIdx.x, blockDim.x); // (the general instantiation has no registers to spare for the marks)
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
      open_run();
      ring_ready(x);
      HSRANS_GS(const uint64_t t3 = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;)
      // kp.group_prio (per mille; 350 by default): the younger half of the workgroup's waves decodes that share of its run at
      // raised instruction priority (s_setprio) — the SIMD otherwise serves its oldest wave first, the older half of the waves is
      // done 8 us before the younger one and waits at the round's barrier.  Unlike the one-chain-per-wave launch, whose index
      // gives the classes chains of different lengths, a block's checkpoints are where the encoder put them.
      // (not where the wave's share was already sized by its age class: 100 MB in 256 KiB blocks + G=32: 0.359 -> 0.340 with both)
      uint32_t prio_permille = group_prio != 0 && !weighted && wave >= waves / 2 ? group_prio : 0;
      if (!BATCH && kp.group_prio_class[9] != 0xFFFF) // (0xFFFF there: no per-class table, group_prio's rule)
      {
        const uint32_t per = waves >= 4 ? waves / 4 : 1;
        prio_permille = weighted ? kp.group_prio_class[8 + half] : kp.group_prio_class[half * 4 + (wave / per < 4 ? wave / per : 3)];
      }
      const uint32_t prio_steps = prio_permille != 0 ? (uint32_t)(run_steps * prio_permille / 1000) & ~3u : 0;
      if (prio_steps != 0)
      {
        __builtin_amdgcn_s_setprio(1);
        run_groups<MODE, true, true, PARTS, PARTS>(x, sw, r, c, o, prio_steps);
        __builtin_amdgcn_s_setprio(0);
      }
      run_groups<MODE, true, true, PARTS, PARTS>(x, sw, r, c, o, (uint32_t)run_steps - prio_steps);
      run_tail<MODE, PARTS>(x, r, c, o, run_tail_syms);
      HSRANS_GS(if (HSRANS_STAMPS(kp)) {
        acc_meta += t3 - t2;
        acc_dec += __builtin_amdgcn_s_memrealtime() - t3;
      })
    }
    else if (LEAN) // fill chains (single-symbol blocks): one fill piece each
      for (uint32_t ch = first; ch < last; ch++)
      {
        const Piece *pc = pv.pieces + uni(pv.chain_first[ch]);
        wave_fill<PARTS>(c, uni64(pc->out_off), uni64(pc->fill_len), (uint32_t)uni64(pc->hist_off) & 0xFF);
      }
    else
      for (uint32_t ch = first; ch < last; ch++)
        run_planned_chain<MODE, true>(c, pv, ch, kp);
    HSRANS_GS(stamp_out();)
    advance();
  }
}

// The kernel of the grouped launches (block_/mt_ plans with checkpoints: one workgroup per block, run_grouped) — BASELINE config 4's
// kernel.  A kernel of its own for the same reason as k_decode_direct: inside k_decode<MODE, true> it shared one register
// allocation with five other launch shapes (two more VGPRs there are the difference between 8 and 7 waves per SIMD).
// LDS: [waves x ring][table][2 next-group words, 64 B][table-build scratch, 1 KiB].
template <int MODE, bool LEAN, bool PARTS = false>
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
  run_grouped<MODE, LEAN, false, false, PARTS>(c, pv, kp, waves, wave);
}

// K member streams' groups in one launch (hsrans_decode_device_batch: members that are block_/mt_ plans with checkpoints, 64 states,
// bits <= 12): k_decode_grouped<kModePack64, true>'s body with the member picked up per round.  LDS as k_decode_grouped.
template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_grouped_batch(BatchGroupParams bp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = nullptr; // (set per round from the group's member)
  c.stream_len = 0;
  c.stream_lo = 0;
  c.out = nullptr;
  c.out_cap = 0;
  c.status = nullptr;
  c.bits = bp.bits;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  const uint32_t ring_stride = fast_ring_mode(MODE) ? kFastRingBytes : kWaveRingBytes;
  c.rings = smem + wave * ring_stride;
  c.table = smem + waves * ring_stride;
  c.table_b = c.table;
  c.gtable = nullptr;
  c.scratch_cnt = (uint16_t *)(c.table + table_bytes_for(MODE, c.bits) + 64);
  c.scratch_cum = c.scratch_cnt + 256;
  KParams kp{}; // (nothing of it is read in the BATCH instantiation but the stamps pointer: null)
  PlanView pv{};
  run_grouped<MODE, true, false, true>(c, pv, kp, waves, wave, &bp);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_GROUPED_H
