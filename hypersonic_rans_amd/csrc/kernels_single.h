// kernels_single.h — A raw stream without index: one dependent chain, producer + consumer wavefront — k_decode_single.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_SINGLE_H
#define HSRANS_KERNELS_SINGLE_H

namespace hsrans
{

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

} // namespace hsrans

#endif // HSRANS_KERNELS_SINGLE_H
