// hsrans_dropin.hpp — the MI355X decode path under the reference's own C++ function names and signatures.
//
// The reference's boundary is the pair of function-pointer types main.cpp stores in codec_info_t (src/main.cpp:146-155):
//   size_t (*encodeFunc)(const uint8_t *pInData, const size_t length,   uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist);
//   size_t (*decodeFunc)(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);
// Every function below has exactly one of those two signatures (block_/mt_ encoders take no histogram, as
// block_rANS32x64_16w.h:8 / mt_rANS32x64_16w.h:9), returns bytes produced and 0 on failure.  They live in namespace
// hsrans_hip so that they can be linked next to the reference's own objects: `*_decode_hip_N` entries can be added to
// main.cpp's _Codecs[] table beside the CPU decoders.
//
//   reference symbol (src/…)                                      replacement here
//   rANS32x64_16w_capacity            rANS32x64_16w.cpp:10        hsrans_hip::rANS32x64_16w_capacity
//   rANS32x64_16w_encode_scalar_N     rANS32x64_16w.cpp:4189-     hsrans_hip::rANS32x64_16w_encode_scalar_N
//   rANS32x64_16w_decode_scalar_N, rANS32x64_*_16w_decode_avx2_var{A,B,C}_N, …avx512…  ->  hsrans_hip::rANS32x64_16w_decode_hip_N
//   block_rANS32x64_16w_{capacity,encode_N,decode_N}  block_rANS32x64_16w.h:6-20  ->  hsrans_hip::block_rANS32x64_16w_{capacity,encode_N,decode_hip_N}
//   mt_rANS32x64_16w_{capacity,encode_N,decode_N,decode_mt_N}  mt_rANS32x64_16w.h:7-28  ->  hsrans_hip::mt_rANS32x64_16w_{capacity,encode_N,decode_hip_N}
//   (and the rANS32x32 twins: rANS32x32_16w.h, block_rANS32x32_16w.h, mt_rANS32x32_16w.h);  N = 10 … 15
//   make_hist                         hist.cpp:217                hsrans_hip::make_hist
#ifndef HSRANS_DROPIN_HPP
#define HSRANS_DROPIN_HPP

#include <stddef.h>
#include <stdint.h>

#include "hsrans_hip.h"

namespace hsrans_hip
{

typedef hsrans_hist hist_t; // layout of the reference's hist_t (hist.h:16-20)

void make_hist(hist_t *pHist, const uint8_t *pData, const size_t size, const size_t totalSymbolCountBits);

// the context used by the functions below: device $HSRANS_DEVICE (default 0), created on first use; nullptr if no gfx950 device
hsrans_ctx *default_context();

#define HSRANS_DECL_BITS(N)                                                                                                                        \
  size_t rANS32x32_16w_encode_scalar_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist); \
  size_t rANS32x64_16w_encode_scalar_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist); \
  size_t rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x32_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x64_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);            \
  size_t block_rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);            \
  size_t mt_rANS32x32_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                     \
  size_t mt_rANS32x64_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                     \
  size_t mt_rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);               \
  size_t mt_rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);

size_t rANS32x32_16w_capacity(const size_t inputSize);
size_t rANS32x64_16w_capacity(const size_t inputSize);
size_t block_rANS32x32_16w_capacity(const size_t inputSize);
size_t block_rANS32x64_16w_capacity(const size_t inputSize);
size_t mt_rANS32x32_16w_capacity(const size_t inputSize);
size_t mt_rANS32x64_16w_capacity(const size_t inputSize);

HSRANS_DECL_BITS(10)
HSRANS_DECL_BITS(11)
HSRANS_DECL_BITS(12)
HSRANS_DECL_BITS(13)
HSRANS_DECL_BITS(14)
HSRANS_DECL_BITS(15)
#undef HSRANS_DECL_BITS

} // namespace hsrans_hip

#endif // HSRANS_DROPIN_HPP
