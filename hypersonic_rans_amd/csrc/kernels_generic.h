// kernels_generic.h — Everything else: mt_ without index (private tables), the block_ header walk, index-build passes — run_private_pair, run_block_walk, k_decode.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_GENERIC_H
#define HSRANS_KERNELS_GENERIC_H

namespace hsrans
{

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

} // namespace hsrans

#endif // HSRANS_KERNELS_GENERIC_H
