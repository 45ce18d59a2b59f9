// kernels_batch.h — K independent streams in ONE launch: k_decode_batch (one-chain-per-wave form; the grouped form is run_grouped with Group::member).
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_BATCH_H
#define HSRANS_KERNELS_BATCH_H

namespace hsrans
{

// The reference decodes independent work items from one pool (mt_rANS32x64_16w_decode.cpp:182-224: a task per block; main.cpp:841-898:
// a file after another); here the pool is the device's 8,192 wave slots and a work item is a stream with its own index.  Launched one
// after the other, K streams pay K prologues (every wave of the device fetching states, first chunks and its workgroup's table at the
// same time: ~4.8 us), K tails (the last wave ends ~3.4 us behind the median) and K kernel boundaries (~2.8 us); launched together on
// two HIP streams they are slower still, because every plan is shaped to fill all wave slots and a second launch cannot co-reside.
// One launch over all K streams pays each of the three once: the host deals the wave slots to the streams (whole workgroups: a
// workgroup holds ONE decode table) and every stream's chains to its slots as runs of consecutive chains, sized by the slots' age
// class like the chains of the one-stream launch (hsrans_batch.cpp: batch_deal).  On the device a slot is a 16-byte record
// {member, first chain, end chain, flags}; the wave decodes its run exactly as k_decode_direct does (run_direct_span), from the
// member's own pieces / states / table and into the member's own status word.
typedef const __attribute__((address_space(4))) BatchSlot *kslot_ptr;
typedef const __attribute__((address_space(4))) BatchMember *kmember_ptr;

// The run [first, last) of a member's chains as ONE chain (back to back in stream and output: a mergeable plan): where its words and
// its output start, where its words end, its whole groups and the final partial group's symbols
__device__ __forceinline__ DirectPiece direct_run(const WaveCtx &c, const PersistentArgs &pa, uint32_t first, uint32_t last)
{
  const kptr64 p0 = (kptr64)(uintptr_t)(pa.pieces + first);
  const kptr64 p1 = (kptr64)(uintptr_t)(pa.pieces + (last - 1));
  DirectPiece d;
  d.words = p0[0];
  d.out = p0[1];
  d.steps = (uint32_t)groups_of(c.S, p1[1] - d.out) + ((kptr32)p1)[8];
  d.tail = ((kptr32)p1)[9] & 0xFFFFu;
  d.limit = last < pa.n_chains ? p1[6] : c.stream_len; // the piece behind the run's last one (Piece is 48 bytes), or the stream's end
  return d;
}

// A slot of a 32-state member: its run [ch, end) is cut in two halves that are decoded side by side — A on lanes 0..31, B on lanes
// 32..63, run_pair_groups — as k_decode_direct does with the chains 2w and 2w + 1 of a stream that has the device to itself
// (run_direct_pair); what the pair loop leaves (unequal halves, < 4 groups, the stream's final partial group) is finished one at a time.
// (diagnostics, HSRANS_BATCH_STAMPS: wave w's finish time in finish[w], the launch's first wave's entry in finish[n_waves] — as run_direct_span)
__device__ __forceinline__ void batch_stamp_entry(uint64_t *finish, uint32_t w, uint32_t n_waves, uint32_t lane)
{
  if (finish != nullptr && w == 0 && lane == 0)
    finish[n_waves] = __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void batch_stamp_done(uint64_t *finish, uint32_t w, uint32_t lane)
{
  if (finish != nullptr && lane == 0)
    finish[w] = __builtin_amdgcn_s_memrealtime();
}

template <int MODE>
__device__ __forceinline__ void run_batch_pair(const WaveCtx &c, const KParams &kp, uint32_t ch, uint32_t end, bool check_hist)
{
  const PersistentArgs &pa = kp.pa;
  StreamWin sw;
  Ring ra, rb;
  pair_bind<MODE>(ra, rb, c);
  if (check_hist && threadIdx.x < 64)
  {
    bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off);
    if (same)
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
  if (ch >= end || ch >= pa.n_chains)
  {
    copy_table(); // a wave without chains still takes part in the workgroup's table copy
    return;
  }
  const uint32_t mid = ch + (end - ch + 1) / 2;
  const bool have_b = mid < end;
  const DirectPiece da = direct_run(c, pa, ch, mid);
  const DirectPiece db = have_b ? direct_run(c, pa, mid, end) : da;
  win_open(sw, c, da.words, have_b ? db.limit : da.limit);
  ring_begin(sw, ra, c, da.words, true, true);
  if (have_b)
    ring_begin(sw, rb, c, db.words, true, true);
  uint32_t x = pa.states[(uint64_t)((c.lane < 32 || !have_b) ? ch : mid) * 32 + (c.lane & 31)];
  copy_table();
  ring_begin_rest(sw, ra, c);
  if (have_b)
    ring_begin_rest(sw, rb, c);
  if (have_b)
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(x)::"memory"); // (the six requests just made are the only younger ones)
  else
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(x)::"memory");
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

// A slot of a 64-state member with a WIDE histogram (13-15 bits: k_decode_dual's launch shape — one 16-wave workgroup per CU, two rings
// per wave, the 8-byte table at 13 bits, the rank table at LDS address 0 at 14 / 15): the slot's run [ch, end) cut in two halves that are
// decoded as two chains in one instruction stream (run_dual_fast), as k_decode_dual does with chains 2w and 2w + 1.
template <int MODE>
__device__ __forceinline__ void run_batch_dual(const WaveCtx &c, const KParams &kp, uint32_t ch, uint32_t end, bool check_hist)
{
  const PersistentArgs &pa = kp.pa;
  constexpr uint32_t kDualRing = kFastRingBytes;
  // the host-built table, the first workgroup of the member also checks the histogram it was built from against the stream's
  {
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8;
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    if (check_hist && threadIdx.x < 64)
    {
      bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off);
      if (same)
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
  }
  if (ch >= end || ch >= pa.n_chains)
  {
    __syncthreads(); // (the table copy's barrier: every wave of the workgroup, with or without chains)
    return;
  }
  const uint32_t mid = ch + (end - ch + 1) / 2;
  const bool have_b = mid < end;
  const DirectPiece da = direct_run(c, pa, ch, mid);
  const DirectPiece db = have_b ? direct_run(c, pa, mid, end) : da;
  uint32_t xa = pa.states[(uint64_t)ch * 64 + c.lane];
  uint32_t xb = pa.states[(uint64_t)(have_b ? mid : ch) * 64 + c.lane];
  StreamWin sw;
  RingD ra, rb;
  ring_bind(ra.r, c.rings, 9, true);
  ring_bind(rb.r, c.rings + kDualRing, 9, true);
  uint32_t vm = 0;
  win_open(sw, c, da.words, have_b ? db.limit : da.limit); // the two halves are neighbours in the stream: one window
  ring_begin_counted(sw, ra, c, da.words, vm, true);
  if (have_b)
    ring_begin_counted(sw, rb, c, db.words, vm, true);
  __syncthreads(); // the table is in LDS
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa), "+v"(xb)::"memory"); // states, table and the first two chunks of both rings
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
  vm = 0;
  ra.seq1 = ra.seq2 = ra.seq3 = rb.seq1 = rb.seq2 = rb.seq3 = 0;
  uint64_t oa = da.out, ob = db.out;
  uint32_t sa = da.steps, sb = have_b ? db.steps : 0;
  uint32_t both = have_b ? (sa < sb ? sa : sb) & ~3u : 0;
  sa -= both;
  sb -= both;
  run_dual_fast<MODE>(xa, xb, sw, ra, rb, c, oa, ob, both, vm);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  run_groups<MODE>(xa, sw, ra.r, c, oa, sa);
  if (have_b)
    run_groups<MODE>(xb, sw, rb.r, c, ob, sb);
  run_tail<MODE>(xa, ra.r, c, oa, da.tail);
  if (have_b)
    run_tail<MODE>(xb, rb.r, c, ob, db.tail);
}

// the launch of wide-histogram members: LDS as k_decode_dual ([table][16 x 2 rings] with the rank table first, [rings][table] with the 8-byte one)
template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(102))) k_decode_batch_dual(BatchParams bp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * waves + wave;
  const kslot_ptr sp = (kslot_ptr)(uintptr_t)(bp.slots + w);
  const uint32_t member = uni(sp->member), ch = uni(sp->begin), end = uni(sp->end), flags = uni(sp->flags);
  const kmember_ptr mp = (kmember_ptr)(uintptr_t)(bp.members + member);
  const BatchIO &io = bp.io[member];
  WaveCtx c;
  c.stream = io.stream;
  c.stream_len = io.stream_len;
  c.stream_lo = 0;
  c.out = io.out;
  c.out_cap = io.out_cap;
  c.status = mp->status;
  c.bits = mp->bits;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  constexpr uint32_t kDualRing = kFastRingBytes;
  if (MODE == kModeRank)
  {
    if (uni(lds_address(smem)) != 0) // (the hand-scheduled group takes the slot as the rank byte's address)
    {
      if (threadIdx.x == 0)
        atomicOr(c.status, kStatusOutOfRange);
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
  c.gtable = mp->table;
  c.scratch_cnt = (uint16_t *)smem;
  c.scratch_cum = (uint16_t *)(smem + 512);
  KParams kp{};
  kp.pa.pieces = mp->pieces;
  kp.pa.states = mp->states;
  kp.pa.n_chains = mp->n_chains;
  kp.pa.hist_off = mp->hist_off;
  kp.pa.table = mp->table;
  kp.pa.hist_copy = mp->hist_copy;
  batch_stamp_entry(bp.finish, w, gridDim.x * waves, c.lane);
  run_batch_dual<MODE>(c, kp, ch, end, (flags & kBatchSlotCheckHist) != 0);
  batch_stamp_done(bp.finish, w, c.lane);
}

template <int MODE, bool PAIR> // PAIR: the launch's members are 32-state plans (a kernel of its own: the 64-state launch keeps its register allocation)
__device__ __forceinline__ void batch_body(const BatchParams &bp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * waves + wave;
  // this wave's slot; the workgroup's member is the one of its first slot (the host never mixes members inside a workgroup)
  const kslot_ptr sp = (kslot_ptr)(uintptr_t)(bp.slots + w);
  const uint32_t member = uni(sp->member), ch = uni(sp->begin), end = uni(sp->end), flags = uni(sp->flags);
  const kmember_ptr mp = (kmember_ptr)(uintptr_t)(bp.members + member);
  const BatchIO &io = bp.io[member];
  WaveCtx c;
  c.stream = io.stream;
  c.stream_len = io.stream_len;
  c.stream_lo = 0;
  c.out = io.out;
  c.out_cap = io.out_cap;
  c.status = mp->status;
  c.bits = mp->bits;
  c.S = PAIR ? 32 : 64; // (a launch's members share one state count: the host puts 64- and 32-state members into launches of their own)
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  c.rings = smem + wave * kFastRingBytes;
  c.table = smem + waves * kFastRingBytes;
  c.table_b = c.table;
  c.gtable = mp->table;
  c.scratch_cnt = (uint16_t *)smem;
  c.scratch_cum = (uint16_t *)(smem + 512);
  // the member's plan in the shape run_direct_span reads it (everything below lands in SGPRs; nothing of `kp` survives as memory)
  KParams kp{};
  kp.pa.pieces = mp->pieces;
  kp.pa.states = mp->states;
  kp.pa.n_chains = mp->n_chains;
  kp.pa.hist_off = mp->hist_off;
  kp.pa.table = mp->table;
  kp.pa.hist_copy = mp->hist_copy;
  kp.finish = bp.finish;
  kp.stamps = bp.stamps;
  if (PAIR)
  {
    batch_stamp_entry(bp.finish, w, gridDim.x * waves, c.lane);
    run_batch_pair<MODE>(c, kp, ch, end, (flags & kBatchSlotCheckHist) != 0);
    batch_stamp_done(bp.finish, w, c.lane);
  }
  else
    run_direct_span<MODE>(c, kp, waves, w, ch, end, (flags & kBatchSlotCheckHist) != 0);
}

template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_batch(BatchParams bp)
{
  batch_body<MODE, false>(bp);
}

// the same launch for 32-state members (two runs per wave, one per half: run_batch_pair)
template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_batch_pair(BatchParams bp)
{
  batch_body<MODE, true>(bp);
}

// hsrans_ctx_calibrate_runs' launches (and HSRANS_BATCH_STAMPS' ones): the same kernel under a name of its own, so that a profile
// of a run that calibrates first lists the calibration apart from the decodes it is there to measure (as k_calibrate does)
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_calibrate_batch(BatchParams bp)
{
  batch_body<kModePack64, false>(bp);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_BATCH_H
