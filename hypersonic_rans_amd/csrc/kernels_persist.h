// kernels_persist.h — Uniform-interval raw plans (a checkpoint every G groups): persistent launch with static runs + ticket queues — run_persistent, run_persistent_pair, k_decode_persist.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_PERSIST_H
#define HSRANS_KERNELS_PERSIST_H

namespace hsrans
{

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
    run_groups<MODE, true, true>(x, sw, r, c, g.o, g.steps); // (strict wait: 0.479 -> 0.499 replayed with a checkpoint every 32 groups)
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
    run_groups<MODE, true, true>(x, sw, r, c, g.o, g.steps); // (strict wait: 0.479 -> 0.499 replayed with a checkpoint every 32 groups)
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

} // namespace hsrans

#endif // HSRANS_KERNELS_PERSIST_H
