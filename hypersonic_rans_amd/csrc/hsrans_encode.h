// Launch interface of the gfx950 encoder (hsrans_encode.hip) for the C ABI (hsrans_capi.cpp).
#ifndef HSRANS_ENCODE_H
#define HSRANS_ENCODE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsrans
{

struct EncParams
{
  const uint8_t *in; // device, 16-byte aligned
  uint64_t n;
  uint8_t *out; // device, 16-byte aligned
  uint64_t out_cap;
  uint8_t *scratch;      // n_blocks slots of slot_bytes; a block's image ends at the end of its slot
  uint64_t slot_bytes;   // encode_slot_bytes()
  uint64_t *image_bytes; // [n_blocks] bytes of block b's image (header + words, or the 8-byte single-symbol marker)
  uint64_t *image_off;   // [n_blocks] position of the image in the stream
  uint64_t *fits;        // device memory: K_scan's copy of result[1] for K_gather (result may be page-locked host memory); unused up to kSelfScanBlocks blocks and by raw encodes
  uint64_t *result;      // [0] stream length, [1] 1 when it fits out_cap (else nothing is written to out), [2] chains,
                         // [3] blocks that are not single-symbol blocks, [4] position of the counts of the last such block
  uint64_t block;        // symbols per block (multiple of 64)
  uint32_t n_blocks;
  uint32_t S, bits;
  // sidecar plan (index_interval != 0 or a device plan was asked for)
  uint32_t interval;     // checkpoint every `interval` groups inside a block (multiple of 4; 0 = none)
  uint32_t max_ck;       // checkpoint slots per block
  uint32_t *ck_states;   // [n_blocks * max_ck * S] coder states at the checkpoints
  uint32_t *ck_pos;      // [n_blocks * max_ck] bytes between the decoder's read cursor at the checkpoint and the end of the block's words
  uint32_t *chain_count; // [n_blocks] chains of block b: 1 + its checkpoints (single-symbol block: 1)
  uint32_t *chain_off;   // [n_blocks] first chain of block b (K_scan)
  uint8_t *plan;         // plan blob to fill (K_plan) or null
  void *groups;          // Group[n_blocks * group_split] for the grouped decode launch (K_plan) or null
  uint32_t group_split;  // parts a block's chains are cut into (few large blocks: more workgroup tasks than blocks)
  uint32_t n_chains;     // total, known after K_scan (K_plan)
  // raw streams (launch_encode_raw): n_blocks = 1, block = n, one slot
  const uint32_t *raw_counts;   // raw: [256] byte counts of the input (k_raw_histogram's output), normalised by the coding wavefront;
                                // mt_: [n_blocks][256] byte counts per block, filled by k_block_histograms in launch_encode (null: every wavefront counts its own block)
  const uint16_t *given_counts; // [256] or null: the caller's normalised histogram (hist_t::symbolCount), used instead
  const uint32_t *ck_groups;    // or null: ascending group indices (multiples of 4) to checkpoint at, instead of `interval`
  uint32_t n_ck_groups;
  uint64_t *stamps; // diagnostics (HSRANS_DEBUG_STAMPS=1): per block {start, histogram done, table done, words done} s_memrealtime; else null
};

uint32_t encode_block_count(uint64_t n, uint64_t block, uint32_t S); // 0: too many blocks
uint64_t encode_slot_bytes(uint64_t block, uint32_t S);
constexpr uint32_t kEncResultWords = 8;
// asynchronous on `stream`: K_enc, K_scan, K_gather.  `prepared` is the calling context's flag: the dynamic-LDS attribute
// is per device, so it is raised once per context, not once per process
hipError_t launch_encode(const EncParams &ep, hipStream_t stream, bool *prepared);
// asynchronous on `stream`: [memset + K_hist ->] K_raw -> K_copy.  result[0] stream length, [1] fits out_cap, [2] listed checkpoints not met (0)
hipError_t launch_encode_raw(const EncParams &ep, uint32_t *d_counts, hipStream_t stream, bool *prepared);
// asynchronous on `stream`: K_plan (needs ep.plan, ep.n_chains; after launch_encode's results are known)
hipError_t launch_encode_plan(const EncParams &ep, hipStream_t stream);

} // namespace hsrans

#endif // HSRANS_ENCODE_H
