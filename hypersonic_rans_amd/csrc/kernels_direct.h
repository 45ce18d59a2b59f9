// kernels_direct.h — One chain (run of chains) per wave — the headline launch: run_direct, run_direct_pair, k_decode_direct, k_calibrate.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_DIRECT_H
#define HSRANS_KERNELS_DIRECT_H

namespace hsrans
{

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

// `w`: the wave's index in the launch (stamps, finish times); chains [ch, end) of kp.pa's plan are its run (ch >= n_chains: none);
// `check_hist`: this workgroup compares the host-built table's histogram with the one in the stream
template <int MODE>
__device__ __forceinline__ void run_direct_span(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w, uint32_t ch, uint32_t end, bool check_hist)
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
    if (check_hist && threadIdx.x < 64)
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
    run_groups<MODE, true, false, true>(x, sw, r, c, o, (uint32_t)groups_of(c.S, run_end_out - o));
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

template <int MODE>
__device__ void run_direct(const WaveCtx &c, const KParams &kp, uint32_t waves, uint32_t w)
{
  const uint32_t n = kp.pa.n_chains;
  const uint32_t R = kp.pa.run_chains ? kp.pa.run_chains : 1; // launch_decode: W * R >= n
  const uint32_t ch = w * R;                                  // this wave's run: chains [ch, end)
  run_direct_span<MODE>(c, kp, waves, w, ch, ch + R < n ? ch + R : n, blockIdx.x == 0);
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
      // (as run_direct: what the first groups read — states, chunks 0 and 1 of both rings, the table — first; mirrors and the chunks kept ahead after it)
      ring_begin(sw, ra, c, da.words, true, true);
      if (have_b)
        ring_begin(sw, rb, c, db.words, true, true);
      uint32_t x = pa.states[(uint64_t)((c.lane < 32 || !have_b) ? a : a + 1) * 32 + (c.lane & 31)];
      if (table_pending)
      {
        copy_table();
        table_pending = false;
      }
      ring_begin_rest(sw, ra, c);
      if (have_b)
        ring_begin_rest(sw, rb, c);
      if (have_b)
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(x)::"memory"); // (the six requests just made are the only younger ones)
      else
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(x)::"memory");
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

} // namespace hsrans

#endif // HSRANS_KERNELS_DIRECT_H
