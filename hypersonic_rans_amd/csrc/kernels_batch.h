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

template <int MODE>
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
  c.S = 64;
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
  run_direct_span<MODE>(c, kp, waves, w, ch, end, (flags & kBatchSlotCheckHist) != 0);
}

template <int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_decode_batch(BatchParams bp)
{
  batch_body<MODE>(bp);
}

// hsrans_ctx_calibrate_runs' launches (and HSRANS_BATCH_STAMPS' ones): the same kernel under a name of its own, so that a profile
// of a run that calibrates first lists the calibration apart from the decodes it is there to measure (as k_calibrate does)
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) k_calibrate_batch(BatchParams bp)
{
  batch_body<kModePack64>(bp);
}

} // namespace hsrans

#endif // HSRANS_KERNELS_BATCH_H
