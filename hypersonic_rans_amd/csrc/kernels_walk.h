// kernels_walk.h — K2: the mt_ header chain followed on the device — k_mt_chase, k_mt_fill.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_WALK_H
#define HSRANS_KERNELS_WALK_H

namespace hsrans
{

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

// ---------------------------------------------------------------------------------------------------------------
// The indexed plan of a stream whose first decode has just recorded its checkpoints (hsrans_decode_device_indexing), assembled
// ON THE DEVICE: the base plan (one single-piece chain per mt_ block), the states and read cursors the recording pass left every
// `interval` groups (slot = absolute group / interval) -> the plan with one chain per block start and per checkpoint inside the
// blocks, byte for byte what the host writes for the same input (hsrans_capi.cpp add_interval_chains + PlanBuilder::serialize),
// plus the Group records of the grouped launch.  (Round 3 brought 7.5-15 MB of checkpoints down to the host, assembled the blob
// on one core and sent it up again: 0.8 ms of the first decode's 1.3-2.5 ms.)
//   k_index_count   one workgroup: chains per block, their exclusive scan -> chain_off[], total -> result[0]
//   k_index_fill    one wavefront per block: header, chain table, pieces, start states, groups
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t index_chains_of(const Piece &bp, uint32_t interval)
{
  if (bp.flags & kPieceFill)
    return 1;
  return bp.steps == 0 ? 1 : (bp.steps + interval - 1) / interval; // (add_interval_chains: g = 0, interval, ... < steps; at least g = 0)
}

__global__ void __launch_bounds__(1024) k_index_count(IndexArgs a)
{
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const uint32_t *cf0 = (const uint32_t *)(a.base + plan_chain_first_off());
  const Piece *pc0 = (const Piece *)(a.base + plan_pieces_off(a.n_base));
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < a.n_base; base += 1024)
  {
    const uint32_t ch = base + threadIdx.x;
    uint32_t v = 0;
    if (ch < a.n_base)
    {
      const Piece bp = pc0[cf0[ch]];
      v = index_chains_of(bp, a.interval);
      if (!(bp.flags & kPieceFill)) // blocks with a histogram: when there is exactly one, the plan's chains share its table
      {                             // (PlanBuilder::serialize: shared_hist = every non-fill piece has the same hist_off; mt_ blocks never share one)
        atomicAdd((unsigned long long *)&a.result[1], 1ull);
        atomicMax((unsigned long long *)&a.result[2], (unsigned long long)bp.hist_off);
        // [3]: the fewest chains of a coded block that is not the plan's last chain (k_decode_spread's condition); stored as ~min so
        // that the zeroed word means "none seen" (= any number will do)
        if (ch + 1 < a.n_base)
          atomicMax((unsigned long long *)&a.result[3], ~(unsigned long long)v);
      }
    }
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d)
        incl += o;
    }
    if (lane == 63)
      wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = carry_s;
    for (uint32_t w = 0; w < wave; w++)
      before += wave_tot[w];
    if (ch < a.n_base)
      a.chain_off[ch] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    a.result[0] = carry_s;
}

__global__ void __launch_bounds__(64) k_index_fill(IndexArgs a)
{
  const uint32_t ch = blockIdx.x, lane = threadIdx.x, S = a.S;
  const uint32_t nc = (uint32_t)a.result[0];
  if (nc > a.max_chains) // (the host sized the blob for max_chains: it reads result[0] and fails the call)
    return;
  const uint32_t *cf0 = (const uint32_t *)(a.base + plan_chain_first_off());
  const Piece *pc0 = (const Piece *)(a.base + plan_pieces_off(a.n_base));
  const uint32_t *st0 = (const uint32_t *)(a.base + plan_states_off(a.n_base, a.n_base));
  uint32_t *chain_first = (uint32_t *)(a.plan + plan_chain_first_off());
  Piece *pieces = (Piece *)(a.plan + plan_pieces_off(nc));
  uint32_t *states = (uint32_t *)(a.plan + plan_states_off(nc, nc));
  const Piece bp = pc0[cf0[ch]];
  const bool fill = (bp.flags & kPieceFill) != 0;
  const uint32_t c0 = a.chain_off[ch], count = index_chains_of(bp, a.interval);
  if (ch == 0 && lane == 0)
  {
    PlanHeader h = *(const PlanHeader *)a.base; // container, states, bits, lengths: the base plan's
    h.n_chains = h.n_pieces = nc;
    h.interval = a.interval;
    h.shared_hist = a.result[1] == 1 ? 1 : 0; // (plans K2 wrote leave these two clear even for a single block)
    h.aux_off = h.shared_hist ? a.result[2] : 0;
    *(PlanHeader *)a.plan = h;
    chain_first[nc] = nc;
  }
  const uint64_t T = bp.steps, g_abs0 = bp.out_off / S;
  for (uint32_t k = lane; k < count; k += 64)
  {
    Piece p{};
    if (fill)
      p = bp;
    else
    {
      const uint64_t g = (uint64_t)k * a.interval;
      const uint64_t slot = (g_abs0 + g) / a.interval;
      p.hist_off = bp.hist_off;
      p.out_off = bp.out_off + g * S;
      p.words_off = k == 0 ? bp.words_off : a.ck_words[slot];
      const uint64_t steps = T - g < a.interval ? T - g : a.interval;
      p.steps = (uint32_t)steps;
      p.tail = (uint16_t)(g + steps == T ? bp.tail : 0);
    }
    p.flags |= kPieceChainStart;
    p.state_idx = c0 + k;
    pieces[c0 + k] = p;
    chain_first[c0 + k] = c0 + k;
  }
  for (uint32_t k = 0; k < count; k++)
    if (lane < S)
    {
      uint32_t v = 0;
      if (!fill)
        v = k == 0 ? st0[(uint64_t)bp.state_idx * S + lane] : a.ck_states[((g_abs0 + (uint64_t)k * a.interval) / a.interval) * S + lane];
      states[(uint64_t)(c0 + k) * S + lane] = v;
    }
  if (a.groups != nullptr && lane < a.group_split)
  {
    // the block's words end at the next rANS block's histogram (as hsrans_capi.cpp dplan_fill has it), or at the stream's end
    uint64_t words_end = a.stream_len;
    for (uint32_t nx = ch + 1; nx < a.n_base; nx++)
    {
      const Piece &q = pc0[cf0[nx]];
      if (!(q.flags & kPieceFill))
      {
        words_end = q.hist_off;
        break;
      }
    }
    const uint32_t by_size = group_parts_of(count, a.group_split);
    const uint32_t parts = fill || by_size < 1 ? 1 : by_size;
    Group g{};
    g.flags = fill ? kGroupFill : kGroupMergeable;
    g.hist_off = fill ? 0 : bp.hist_off;
    if (lane < parts)
    {
      const uint32_t lo = (uint32_t)((uint64_t)count * lane / parts), hi = (uint32_t)((uint64_t)count * (lane + 1) / parts);
      g.begin = c0 + lo;
      g.piece0 = c0 + lo;
      g.count = hi - lo;
      g.words_end = hi < count ? a.ck_words[(g_abs0 + (uint64_t)hi * a.interval) / a.interval] : words_end;
    }
    else
    {
      g.begin = c0;
      g.piece0 = c0;
      g.count = 0;
      g.flags = kGroupFill;
      g.words_end = words_end;
    }
    a.groups[(uint64_t)ch * a.group_split + lane] = g;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 64-bit fingerprint of a stream in device memory (hsrans_decode_host's index cache: the index a first decode left behind may only
// serve a later call when EVERY byte of the stream is the same).  Fixed launch shape (the value depends on which thread reads which
// word): thread t folds the 16-byte words t, t + T, t + 2T, ... in order; the per-thread values are mixed with t and summed.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kChecksumGrid = 1024, kChecksumBlock = 256;
__global__ void __launch_bounds__(kChecksumBlock) k_stream_checksum(const uint8_t *in, uint64_t n, unsigned long long *sum)
{
  const uint64_t T = (uint64_t)kChecksumGrid * kChecksumBlock, t = (uint64_t)blockIdx.x * kChecksumBlock + threadIdx.x;
  uint64_t h = 0x9E3779B97F4A7C15ull * (t + 1);
  const uint64_t words = n / 16;
  for (uint64_t i = t; i < words; i += T)
  {
    const uint4 w = ((const uint4 *)in)[i];
    h = (h ^ ((uint64_t)w.x | ((uint64_t)w.y << 32))) * 0xFF51AFD7ED558CCDull;
    h = (h ^ (h >> 29) ^ ((uint64_t)w.z | ((uint64_t)w.w << 32))) * 0xC4CEB9FE1A85EC53ull;
  }
  if (t == 0)
    for (uint64_t i = words * 16; i < n; i++)
      h = (h ^ in[i]) * 0xFF51AFD7ED558CCDull;
  h ^= h >> 32;
  // wave sum first: 4,096 atomics instead of 262,144
  for (int d = 32; d >= 1; d >>= 1)
    h += __shfl_xor(h, d, 64);
  if ((threadIdx.x & 63) == 0)
    atomicAdd(sum, (unsigned long long)h);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_WALK_H
