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
  uint64_t *result;      // [0] stream length, [1] 1 when it fits out_cap (else nothing is written to out)
  uint64_t block;        // symbols per block (multiple of 64)
  uint32_t n_blocks;
  uint32_t S, bits;
  uint64_t *stamps; // diagnostics (HSRANS_DEBUG_STAMPS=1): per block {start, histogram done, table done, words done} s_memrealtime; else null
};

uint32_t encode_block_count(uint64_t n, uint64_t block, uint32_t S); // 0: too many blocks
uint64_t encode_slot_bytes(uint64_t block, uint32_t S);
// asynchronous on `stream`: K_enc, K_scan, K_gather
hipError_t launch_encode(const EncParams &ep, hipStream_t stream);

} // namespace hsrans

#endif // HSRANS_ENCODE_H
