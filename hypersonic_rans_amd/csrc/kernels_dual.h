// kernels_dual.h — Two 64-state chains per wave (13-15-bit histograms): k_decode_dual.
// Part of the one device translation unit hsrans_kernels.hip (which includes the parts in dependency order and holds the host-side launcher).
#ifndef HSRANS_KERNELS_DUAL_H
#define HSRANS_KERNELS_DUAL_H

namespace hsrans
{

// ---------------------------------------------------------------------------------------------------------------
// Two 64-state chains per wave (k_decode_dual): for the table layouts that leave room for only one workgroup per CU (the
// 8-byte-per-slot table at 13 bits: 64 KiB) or whose group step is three dependent LDS round trips (the rank table at 14 / 15
// bits), a wave's single dependent chain leaves the SIMD idle most of the time (4 waves per SIMD, each waiting on LDS).
// Wave w decodes chains 2w and 2w + 1 of a one-chain-per-wave index side by side: two independent dependency chains in one
// instruction stream, which the scheduler interleaves.
//
// Two rings per wave need exact waits: "vmcnt(2)" in ring_advance is right for ONE ring (derivation there) but would make
// ring A wait for a request ring B issued a moment ago — a full memory round trip every few groups.  So this path COUNTS its
// vector-memory instructions (the stream requests and the output stores are all issued from asm here, nothing else touches
// vmcnt inside the loop) and waits with vmcnt(number of operations issued after the one it needs): exact, because vector
// memory operations of a wave complete in issue order.
// ---------------------------------------------------------------------------------------------------------------
struct RingD
{
  Ring r;
  uint32_t seq1, seq2, seq3; // value of the wave's VM-instruction count right after the requests for chunks k+1 / k+2 / k+3
};

__device__ __forceinline__ void ring_request_counted(const StreamWin &sw, const Ring &r, const WaveCtx &c, uint32_t chunk, uint32_t &vm)
{
  ring_request(sw, r, c, chunk);
  vm += (chunk & (kRingSlots - 1)) == 0 ? 2 : 1; // slot 0 also refills the mirror
}

// (`later`: chunks 0 and 1 only, without chunk 0's mirror — the caller asks for the rest once what its first groups read has landed: ring_begin)
__device__ __forceinline__ void ring_begin_counted(const StreamWin &sw, RingD &d, const WaveCtx &c, uint64_t pos, uint32_t &vm, bool later = false)
{
  pos = uni64(pos);
  const uint32_t rel = (uint32_t)(pos - sw.base);
  d.r.voff0 = rel & ~15u;
  d.r.cur = (rel - d.r.voff0) >> 1;
  d.r.k = 0;
  if (later)
  {
    ring_request(sw, d.r, c, 0, false);
    ring_request(sw, d.r, c, 1);
    vm += 2;
    d.seq1 = d.seq2 = d.seq3 = vm;
    return;
  }
  ring_request_counted(sw, d.r, c, 0, vm);
  ring_request_counted(sw, d.r, c, 1, vm);
  d.seq1 = vm;
  ring_request_counted(sw, d.r, c, 2, vm);
  d.seq2 = vm;
  if (HSRANS_RING_AHEAD == 3)
    ring_request_counted(sw, d.r, c, 3, vm);
  d.seq3 = vm;
}

// The same step hand-scheduled (8-byte table entries): at 4 waves per SIMD a wave issues one instruction every 4-5 cycles, so
// the instruction COUNT per group is what a two-chain wave is bound by — the compiler's version of the loop above spends ~59
// instructions per group (31 of them scalar: two wrapped cursors, two rings' advance logic, mask bookkeeping); this one 16.5:
// cursors are plain LDS addresses (re-based every 4 groups: whole-chunk mirrors), chain A's mask lives in s[92:93], chain B's
// in VCC, the word reads and the merges run under EXEC = mask.
#define HSRANS_DUAL_GROUP(A0, A1, B0, B1)                                                                                                            \
  "v_and_b32 %[ta], %[xa], %[vmask]\n\t"                                                                                                             \
  "v_and_b32 %[tb], %[xb], %[vmask]\n\t"                                                                                                             \
  "v_lshl_add_u32 %[ta], %[ta], 3, %[stab]\n\t"                                                                                                      \
  "v_lshl_add_u32 %[tb], %[tb], 3, %[stab]\n\t"                                                                                                      \
  "ds_read_b64 v[" #A0 ":" #A1 "], %[ta]\n\t"                                                                                                        \
  "ds_read_b64 v[" #B0 ":" #B1 "], %[tb]\n\t"                                                                                                        \
  "v_lshrrev_b32 %[xa], %[vbits], %[xa]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[xb], %[vbits], %[xb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xa], v" #A0 ", %[xa], v" #A1 "\n\t"                                                                                               \
  "v_cmp_gt_u32 s[92:93], %[lim], %[xa]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xb], v" #B0 ", %[xb], v" #B1 "\n\t"                                                                                               \
  "v_cmp_gt_u32 vcc, %[lim], %[xb]\n\t"                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[ta], s92, 0\n\t"                                                                                                            \
  "v_mbcnt_hi_u32_b32 %[ta], s93, %[ta]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[ta], %[ta], 1, %[sa]\n\t"                                                                                                       \
  "v_mbcnt_lo_u32_b32 %[tb], vcc_lo, 0\n\t"                                                                                                         \
  "v_mbcnt_hi_u32_b32 %[tb], vcc_hi, %[tb]\n\t"                                                                                                     \
  "v_lshl_add_u32 %[tb], %[tb], 1, %[sb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "ds_read_u16 %[ta], %[ta]\n\t"                                                                                                                    \
  "s_mov_b64 exec, vcc\n\t"                                                                                                                         \
  "ds_read_u16 %[tb], %[tb]\n\t"                                                                                                                    \
  "s_bcnt1_i32_b64 %[st], s[92:93]\n\t"                                                                                                             \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                         \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                  \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_lshl_or_b32 %[xb], %[xb], 16, %[tb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "v_lshl_or_b32 %[xa], %[xa], 16, %[ta]\n\t"                                                                                                       \
  "s_mov_b64 exec, -1\n\t"

// four groups of chain A and four of chain B; acc_a / acc_b = this lane's four symbols of each (before the quad transpose)
__device__ __forceinline__ void dual_groups4(uint32_t &xa, uint32_t &xb, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_table, uint32_t &acc_a, uint32_t &acc_b)
{
  uint32_t ta, tb, st;
  asm volatile(HSRANS_DUAL_GROUP(64, 65, 72, 73) HSRANS_DUAL_GROUP(66, 67, 74, 75) HSRANS_DUAL_GROUP(68, 69, 76, 77) HSRANS_DUAL_GROUP(70, 71, 78, 79)
               "v_perm_b32 %[aa], v66, v64, %[selp]\n\t"
               "v_perm_b32 %[ta], v70, v68, %[selp]\n\t"
               "v_perm_b32 %[aa], %[ta], %[aa], %[selq]\n\t"
               "v_perm_b32 %[ab], v74, v72, %[selp]\n\t"
               "v_perm_b32 %[tb], v78, v76, %[selp]\n\t"
               "v_perm_b32 %[ab], %[tb], %[ab], %[selq]"
               : [xa] "+v"(xa), [xb] "+v"(xb), [sa] "+s"(s_a), [sb] "+s"(s_b), [aa] "=&v"(acc_a), [ab] "=&v"(acc_b), [ta] "=&v"(ta), [tb] "=&v"(tb), [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [stab] "s"(s_table), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "s92", "s93", "vcc", "scc", "memory");
}

// The step for the rank table (kModeRank; 14 / 15 bits).  The rank bytes start at LDS address 0 (k_decode_dual puts the table
// first), so the slot is the address of the first gather; the entries follow at %[sent] = 2^bits.  Three dependent LDS reads
// per group and chain — rank byte, entry, stream word — the two chains' reads interleaved; A's mask in s[92:93], B's in VCC.
// Per group and chain: 10 vector instructions (+ packing), 3 LDS.  Measured (15 bits, 100 MB): 12.3 vector instructions and
// 13.3 LDS cycles per group, 7.3 of them bank conflicts — 5 from the byte gather alone: 64 random dwords over the LDS's 32 banks
// (with every lane reading ONE entry the conflicts fall to 5.0, with the byte read made conflict-free as well to 0.03 and the
// LDS cycles to 6.1; one-off builds with the gathers' addresses replaced, not kept).
#define HSRANS_DUAL_GROUP_RANK(A0, A1, B0, B1)                                                                                                       \
  "v_and_b32 %[ga], %[xa], %[vmask]\n\t"                                                                                                             \
  "v_and_b32 %[gb], %[xb], %[vmask]\n\t"                                                                                                             \
  "ds_read_u8 v" #A0 ", %[ga]\n\t"                                                                                                                   \
  "ds_read_u8 v" #B0 ", %[gb]\n\t"                                                                                                                   \
  "v_lshrrev_b32 %[xa], %[vbits], %[xa]\n\t"                                                                                                         \
  "v_lshrrev_b32 %[xb], %[vbits], %[xb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_lshl_add_u32 %[ta], v" #A0 ", 3, %[sent]\n\t"                                                                                                   \
  "ds_read_b64 v[" #A0 ":" #A1 "], %[ta]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_lshl_add_u32 %[tb], v" #B0 ", 3, %[sent]\n\t"                                                                                                   \
  "ds_read_b64 v[" #B0 ":" #B1 "], %[tb]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xa], v" #A0 ", %[xa], v" #A1 "\n\t"                                                                                               \
  "v_add_u32 %[xa], %[xa], %[ga]\n\t"                                                                                                                \
  "v_cmp_gt_u32 s[92:93], %[lim], %[xa]\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_mad_u32_u24 %[xb], v" #B0 ", %[xb], v" #B1 "\n\t"                                                                                               \
  "v_add_u32 %[xb], %[xb], %[gb]\n\t"                                                                                                                \
  "v_cmp_gt_u32 vcc, %[lim], %[xb]\n\t"                                                                                                             \
  "v_mbcnt_lo_u32_b32 %[ta], s92, 0\n\t"                                                                                                            \
  "v_mbcnt_hi_u32_b32 %[ta], s93, %[ta]\n\t"                                                                                                        \
  "v_lshl_add_u32 %[ta], %[ta], 1, %[sa]\n\t"                                                                                                       \
  "v_mbcnt_lo_u32_b32 %[tb], vcc_lo, 0\n\t"                                                                                                         \
  "v_mbcnt_hi_u32_b32 %[tb], vcc_hi, %[tb]\n\t"                                                                                                     \
  "v_lshl_add_u32 %[tb], %[tb], 1, %[sb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "ds_read_u16 %[ta], %[ta]\n\t"                                                                                                                    \
  "s_mov_b64 exec, vcc\n\t"                                                                                                                         \
  "ds_read_u16 %[tb], %[tb]\n\t"                                                                                                                    \
  "s_bcnt1_i32_b64 %[st], s[92:93]\n\t"                                                                                                             \
  "s_lshl1_add_u32 %[sa], %[st], %[sa]\n\t"                                                                                                         \
  "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                                                                  \
  "s_lshl1_add_u32 %[sb], %[st], %[sb]\n\t"                                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                        \
  "v_lshl_or_b32 %[xb], %[xb], 16, %[tb]\n\t"                                                                                                       \
  "s_mov_b64 exec, s[92:93]\n\t"                                                                                                                    \
  "v_lshl_or_b32 %[xa], %[xa], 16, %[ta]\n\t"                                                                                                       \
  "s_mov_b64 exec, -1\n\t"

__device__ __forceinline__ void dual_groups4_rank(uint32_t &xa, uint32_t &xb, uint32_t &s_a, uint32_t &s_b, const WaveCtx &c, uint32_t s_entries, uint32_t &acc_a, uint32_t &acc_b)
{
  uint32_t ta, tb, ga, gb, st;
  asm volatile(HSRANS_DUAL_GROUP_RANK(64, 65, 72, 73) HSRANS_DUAL_GROUP_RANK(66, 67, 74, 75) HSRANS_DUAL_GROUP_RANK(68, 69, 76, 77) HSRANS_DUAL_GROUP_RANK(70, 71, 78, 79)
               "v_perm_b32 %[aa], v66, v64, %[selp]\n\t"
               "v_perm_b32 %[ta], v70, v68, %[selp]\n\t"
               "v_perm_b32 %[aa], %[ta], %[aa], %[selq]\n\t"
               "v_perm_b32 %[ab], v74, v72, %[selp]\n\t"
               "v_perm_b32 %[tb], v78, v76, %[selp]\n\t"
               "v_perm_b32 %[ab], %[tb], %[ab], %[selq]"
               : [xa] "+v"(xa), [xb] "+v"(xb), [sa] "+s"(s_a), [sb] "+s"(s_b), [aa] "=&v"(acc_a), [ab] "=&v"(acc_b), [ta] "=&v"(ta), [tb] "=&v"(tb), [ga] "=&v"(ga), [gb] "=&v"(gb),
                 [st] "=&s"(st)
               : [vmask] "v"(c.v_mask), [vbits] "v"(c.v_bits), [sent] "s"(s_entries), [lim] "s"(kConsume), [selp] "s"(0x0c0c0703u), [selq] "s"(0x05040100u)
               : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "s92", "s93", "vcc", "scc", "memory");
}

// `both` (a multiple of 4) groups of each of the two chains
template <int MODE>
__device__ __forceinline__ void run_dual_fast(uint32_t &xa, uint32_t &xb, const StreamWin &sw, RingD &ra, RingD &rb, const WaveCtx &c, uint64_t &oa_ref, uint64_t &ob_ref, uint32_t both,
                                              uint32_t &vm)
{
  uint64_t oa = uni64(oa_ref), ob = uni64(ob_ref);
  const OutLanes ol = out_lanes(c.lane, 64);
  const uint32_t s_table = uni(lds_address(c.table));
  FastCursor fa = fast_cursor_open(ra.r), fb = fast_cursor_open(rb.r);
  // (the loop's bookkeeping as in run_groups_fast: nothing is counted but the iterations, one output pointer per chain)
  uint32_t iters = both >> 2;
  uint8_t *pa = (uint8_t *)uni64((uint64_t)(uintptr_t)(c.out + oa)), *pb = (uint8_t *)uni64((uint64_t)(uintptr_t)(c.out + ob)); // one pointer per chain, not base + offset
  oa += (uint64_t)iters * 256;
  ob += (uint64_t)iters * 256;
  auto crossed = [&](FastCursor &f, RingD &d) {
    fast_cursor_cross(f, d.r);
    ring_request(sw, d.r, c, d.r.k + HSRANS_RING_AHEAD);
    // (the constant wait: at most 6 outstanding = this ring's requests for k + 2 and k + 3 and the two stores of each of the two
    // iterations that any three of its crossings span.  Against the exact count (wait_after_crossing<2>): 13 / 14 / 15 bits replayed
    // 0.455 / 0.413 / 0.411 -> 0.475 / 0.421 / 0.420, 15 bits rotated 54.2 -> 53.0 us)
    if (HSRANS_RING_AHEAD == 3)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // (this ring's request for k + 2 and the two stores of an iteration)
  };
  for (; iters != 0; iters--)
  {
    uint32_t acc_a, acc_b;
    if (MODE == kModeRank)
      dual_groups4_rank(xa, xb, fa.addr, fb.addr, c, 1u << c.bits, acc_a, acc_b); // (the table starts at LDS address 0: k_decode_dual)
    else
      dual_groups4(xa, xb, fa.addr, fb.addr, c, s_table, acc_a, acc_b);
    acc_a = quad_transpose(acc_a, ol.sel_a, ol.sel_b);
    acc_b = quad_transpose(acc_b, ol.sel_a, ol.sel_b);
    HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)pa), ol.store_off, acc_a);
    HSRANS_STORE_U32_SADDR((uint8_t *)uni64((uint64_t)(uintptr_t)pb), ol.store_off, acc_b);
    pa += 256;
    pb += 256;
    if (fa.addr >= fa.next_cross)
      crossed(fa, ra);
    if (fb.addr >= fb.next_cross)
      crossed(fb, rb);
  }
  vm = 0;
  ra.seq1 = ra.seq2 = ra.seq3 = rb.seq1 = rb.seq2 = rb.seq3 = 0; // (not kept in the loop; the caller drains the queue behind it anyway)
  fast_cursor_close(fa, ra.r);
  fast_cursor_close(fb, rb.r);
  oa_ref = oa;
  ob_ref = ob;
}

template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(102))) k_decode_dual(KParams kp)
{
  extern __shared__ u32x4 smem_v[];
  uint8_t *smem = (uint8_t *)smem_v;
  const PersistentArgs &pa = kp.pa;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t wave = uni(threadIdx.x >> 6);
  WaveCtx c;
  c.stream = kp.stream;
  c.stream_len = kp.stream_len;
  c.stream_lo = kp.stream_lo;
  c.out = kp.out;
  c.out_cap = kp.out_cap;
  c.status = kp.status;
  c.bits = pa.bits;
  c.S = 64;
  c.lane = threadIdx.x & 63;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_mask) : "s"((1u << c.bits) - 1));
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.v_bits) : "s"(c.bits));
  constexpr uint32_t kDualRing = kFastRingBytes; // whole-chunk mirrors for the hand-scheduled loop (launch_shape sizes the LDS the same way)
  if (MODE == kModeRank)
  {
    // the rank bytes at LDS address 0 (this kernel has no static LDS): the hand-scheduled group uses the slot as the address
    if (uni(lds_address(smem)) != 0) // (would decode garbage silently: report instead; the host discards the output)
    {
      if (threadIdx.x == 0)
        atomicOr(kp.status, kStatusOutOfRange);
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
  c.gtable = pa.table;
  c.scratch_cnt = (uint16_t *)smem;
  c.scratch_cum = (uint16_t *)(smem + 512);
  const uint32_t W = gridDim.x * waves;
  const uint32_t w = blockIdx.x * waves + wave;
  const uint64_t t_entry = HSRANS_STAMPS(kp) ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t t_table = 0, t_ready = 0;

  // the host-built table (always: the launcher only picks this kernel for plans that carry their histogram)
  bool table_pending = true;
  auto fetch_table = [&]() {
    const uint32_t entries = table_bytes_for(MODE, c.bits) / 8;
    for (uint32_t i = threadIdx.x * 2; i < entries; i += blockDim.x * 2)
      *(u32x4 *)(c.table + (uint64_t)i * 8) = *(const u32x4 *)(pa.table + i);
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
      bool same = HSRANS_HIST_IN_RANGE(c, pa.hist_off) || pa.hist_off + 512 <= c.stream_lo;
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
    __syncthreads();
    if (HSRANS_STAMPS(kp))
      t_table = __builtin_amdgcn_s_memrealtime();
  };

  for (uint32_t a = 2 * w; a < pa.n_chains; a += 2 * W)
  {
    const bool have_b = a + 1 < pa.n_chains;
    const DirectPiece da = direct_piece(c, pa, a);
    const DirectPiece db = have_b ? direct_piece(c, pa, a + 1) : da;
    uint32_t xa = pa.states[(uint64_t)a * 64 + c.lane];
    uint32_t xb = pa.states[(uint64_t)(have_b ? a + 1 : a) * 64 + c.lane];
    StreamWin sw;
    RingD ra, rb;
    ring_bind(ra.r, c.rings, 9, true);
    ring_bind(rb.r, c.rings + kDualRing, 9, true);
    uint32_t vm = 0; // vector-memory instructions issued from here on (everything older completes before them anyway)
    win_open(sw, c, da.words, have_b ? db.limit : da.limit); // the two chains are neighbours in the stream: one window
    // what the first groups read first (states, chunks 0 and 1 of both rings, the table), the chunks the rings keep ahead and the
    // mirrors behind that: every wave of the device is here at the same time and a CU takes in ~11 bytes per clock (run_direct)
    ring_begin_counted(sw, ra, c, da.words, vm, true);
    if (have_b)
      ring_begin_counted(sw, rb, c, db.words, vm, true);
    if (table_pending)
    {
      fetch_table();
      table_pending = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa), "+v"(xb)::"memory"); // start of a chain pair: states, table and the first two chunks of both rings
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
    // (the loop's constant wait — at most 6 outstanding at a crossing — holds from its first crossing on: behind a ring's request for
    // chunk 2 there are the other requests just made and two stores per iteration since)
    vm = 0;
    ra.seq1 = ra.seq2 = ra.seq3 = rb.seq1 = rb.seq2 = rb.seq3 = 0;
    if (HSRANS_STAMPS(kp) && t_ready == 0)
      t_ready = __builtin_amdgcn_s_memrealtime();
    uint64_t oa = da.out, ob = db.out;
    uint32_t sa = da.steps, sb = have_b ? db.steps : 0;
    uint32_t both = have_b ? (sa < sb ? sa : sb) & ~3u : 0;
    sa -= both;
    sb -= both;
    run_dual_fast<MODE>(xa, xb, sw, ra, rb, c, oa, ob, both, vm);
    // what is left (a few groups of the longer chain, the stream's final partial group): one chain at a time, the ordinary way
    // (the single-ring wait in ring_advance is only ever stricter than needed here: the other ring's requests are older or done)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    run_groups<MODE>(xa, sw, ra.r, c, oa, sa);
    if (have_b)
      run_groups<MODE>(xb, sw, rb.r, c, ob, sb);
    run_tail<MODE>(xa, ra.r, c, oa, da.tail);
    if (have_b)
      run_tail<MODE>(xb, rb.r, c, ob, db.tail);
  }
  if (table_pending)
    fetch_table();
  if (HSRANS_STAMPS(kp) && c.lane == 0)
  {
    uint64_t *st = kp.stamps + (uint64_t)w * 8;
    st[0] = t_entry;
    st[1] = t_table;
    st[2] = t_ready;
    st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = st[3];
  }
}

} // namespace hsrans

#endif // HSRANS_KERNELS_DUAL_H
