/* hsrans_names.h — the reference's own function names with C linkage ("build adds extern "C" aliases").
 *
 * The reference selects container, state count and histogram width by FUNCTION NAME (rANS32x64_16w_decode_scalar_11,
 * block_rANS32x32_16w_decode_14, mt_rANS32x64_16w_decode_mt_12 ...: src/rANS32x64_16w.h:6-51, src/rANS32x32_16w.h,
 * src/block_rANS32x64_16w.h:6-20, src/block_rANS32x32_16w.h, src/mt_rANS32x64_16w.h:7-28, src/mt_rANS32x32_16w.h) and has C++
 * linkage only.  An FFI caller (ctypes, cgo, JNI ...) binds by symbol name, so every one of those names is exported here
 * UNMANGLED as  hsrans_<reference name>  with the reference's argument lists (the prefix keeps them linkable next to the
 * reference's own objects and next to include/hsrans_dropin.hpp, which has the same names in namespace hsrans_hip).
 *
 *   hsrans_<codec>_capacity(inputSize)                                    <- *_capacity               (rANS32x64_16w.h:6 ...)
 *   hsrans_rANS32x{32,64}_16w_encode_scalar_N(in, len, out, cap, hist)    <- encode_scalar_N          (rANS32x64_16w.h:8-13)
 *   hsrans_rANS32x{32,64}_16w_decode_scalar_N(in, len, out, cap)          <- decode_scalar_N          (rANS32x64_16w.h:46-51)
 *   hsrans_block_rANS32x{32,64}_16w_{encode,decode}_N                     <- block_..._{encode,decode}_N (block_rANS32x64_16w.h:8-20)
 *   hsrans_mt_rANS32x{32,64}_16w_{encode,decode}_N, ..._decode_mt_N(..., void *pThreadPool)  <- mt_rANS32x64_16w.h:9-28
 *   hsrans_<codec>_decode_hip_N                                           the gfx950 kernels whatever the container (0 without a GPU)
 * N = 10 ... 15.  decode_scalar_N / decode_N route like `*_decode_auto_N` (hsrans_dropin.hpp): a stream that is one dependent
 * chain -> this library's host SIMD decoder, mt_ -> GPU.  Return: bytes produced, 0 on any failure (the reference's convention).
 */
#ifndef HSRANS_NAMES_H
#define HSRANS_NAMES_H

#include "hsrans_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

size_t hsrans_rANS32x32_16w_capacity(size_t inputSize);
size_t hsrans_rANS32x64_16w_capacity(size_t inputSize);
size_t hsrans_block_rANS32x32_16w_capacity(size_t inputSize);
size_t hsrans_block_rANS32x64_16w_capacity(size_t inputSize);
size_t hsrans_mt_rANS32x32_16w_capacity(size_t inputSize);
size_t hsrans_mt_rANS32x64_16w_capacity(size_t inputSize);

#define HSRANS_C_DECL(S, N)                                                                                                             \
  size_t hsrans_rANS32x##S##_16w_encode_scalar_##N(const uint8_t *pInData, size_t length, uint8_t *pOutData, size_t outCapacity, const hsrans_hist *pHist); \
  size_t hsrans_rANS32x##S##_16w_decode_scalar_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);     \
  size_t hsrans_rANS32x##S##_16w_decode_hip_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);        \
  size_t hsrans_block_rANS32x##S##_16w_encode_##N(const uint8_t *pInData, size_t length, uint8_t *pOutData, size_t outCapacity);        \
  size_t hsrans_block_rANS32x##S##_16w_decode_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);      \
  size_t hsrans_block_rANS32x##S##_16w_decode_hip_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);  \
  size_t hsrans_mt_rANS32x##S##_16w_encode_##N(const uint8_t *pInData, size_t length, uint8_t *pOutData, size_t outCapacity);           \
  size_t hsrans_mt_rANS32x##S##_16w_decode_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);         \
  size_t hsrans_mt_rANS32x##S##_16w_decode_hip_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity);     \
  size_t hsrans_mt_rANS32x##S##_16w_decode_mt_##N(const uint8_t *pInData, size_t inLength, uint8_t *pOutData, size_t outCapacity, void *pThreadPool);
#define HSRANS_C_DECL_BITS(N) HSRANS_C_DECL(32, N) HSRANS_C_DECL(64, N)
HSRANS_C_DECL_BITS(10)
HSRANS_C_DECL_BITS(11)
HSRANS_C_DECL_BITS(12)
HSRANS_C_DECL_BITS(13)
HSRANS_C_DECL_BITS(14)
HSRANS_C_DECL_BITS(15)
#undef HSRANS_C_DECL_BITS
#undef HSRANS_C_DECL

#ifdef __cplusplus
}
#endif

#endif /* HSRANS_NAMES_H */
