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
//   rANS32x64_16w_decode_scalar_N     rANS32x64_16w.h:48          hsrans_hip::rANS32x64_16w_decode_scalar_N  (the SAME name: runtime dispatch, = _auto_)
//   rANS32x64_*_16w_decode_avx2_var{A,B,C}_N, …avx512…  rANS32x64_16w.h:54-177  ->  hsrans_hip::rANS32x64_16w_decode_{auto,hip}_N
//   block_rANS32x64_16w_{capacity,encode_N,decode_N}  block_rANS32x64_16w.h:6-20  ->  hsrans_hip::block_rANS32x64_16w_{capacity,encode_N,decode_N} (+ decode_hip_N)
//   mt_rANS32x64_16w_{capacity,encode_N,decode_N,decode_mt_N}  mt_rANS32x64_16w.h:7-28  ->  hsrans_hip::mt_rANS32x64_16w_{capacity,encode_N,decode_N,decode_mt_N} (+ decode_hip_N)
//   => a caller written against rANS32x64_16w.h:48, block_rANS32x64_16w.h:19, mt_rANS32x64_16w.h:20 compiles and links with nothing but
//      `using namespace hsrans_hip;` (tests/test_dropin_link.py builds such a caller around the reference's codec_info_t).  The names the
//      reference itself uses route like `*_decode_auto_N`: what is ONE dependent chain goes to the host SIMD decoder, mt_ to the GPU.
//      include/hsrans_names.h exports every one of them with C linkage as hsrans_<reference name> for FFI callers.
//   (and the rANS32x32 twins: rANS32x32_16w.h, block_rANS32x32_16w.h, mt_rANS32x32_16w.h);  N = 10 … 15
//   make_hist                         hist.cpp:217                hsrans_hip::make_hist
//   mt_rANS32x64_16w_decode_mt_N(…, thread_pool *)  mt_rANS32x64_16w.h:23-28  ->  hsrans_hip::mt_rANS32x{32,64}_16w_decode_mt_N (same five
//       arguments, so main.cpp's decode_with_thread_pool_wrapper :163-170 wraps it unchanged; the pool pointer is ignored: the GPU grid takes
//       the pool's place, and without a gfx950 device the blocks go to this library's host decoder on all cores, like `_auto_`)
//
//   block_rANS32x64_decode_wrapper's runtime dispatch (block_rANS32x64_16w_decode.cpp:130-152)  ->  `*_decode_auto_N`: the same
//       decodeFunc signature, routed at run time: a stream that is ONE dependent chain (raw / block_ without an index) goes to this
//       library's own host SIMD decoder (AVX-512 / AVX2 / scalar by CPUID; hsrans_decode_cpu), mt_ streams go to the GPU (and
//       to the host decoder on all cores when no gfx950 device is present).  `*_decode_hip_N`: every decoded byte comes from the GPU
//       kernels and nothing falls back to the host — with ONE host-side step to know about: the FIRST call on a raw stream of >= 1 MiB
//       that comes without an index gets its checkpoints from one pass of this library's host SIMD decoder over the stream (~35 ms per
//       100 MB, while the upload runs; the context then keeps the index: hsrans_decode_host).  HSRANS_HIP_STRICT=1 in the environment
//       makes the one wavefront that decodes such a stream record them instead (~125 ms, once): no host core touches the stream.
//
// Beyond the reference's signatures: `*_decode_hip_with_index_N(in, inLen, out, outCap, plan, planLen)` take
// the sidecar index written by `*_encode_with_index_N` (or hsrans_index_build / hsrans_index_boundaries) — with it ONE raw or block_
// stream fills the GPU (without it those formats are a single dependent chain = one wavefront); `*_pipelined_*` additionally
// overlaps the PCIe legs with the kernels (page-locked buffers: hsrans_host_register).
#ifndef HSRANS_DROPIN_HPP
#define HSRANS_DROPIN_HPP

#include <stddef.h>
#include <stdint.h>

#include "hsrans_hip.h"

struct thread_pool; // the reference's pool (src/thread_pool.h:7), only ever passed through: the GPU grid takes its place

namespace hsrans_hip
{

typedef hsrans_hist hist_t; // layout of the reference's hist_t (hist.h:16-20)

void make_hist(hist_t *pHist, const uint8_t *pData, const size_t size, const size_t totalSymbolCountBits);

// the context used by the functions below: device $HSRANS_DEVICE (default 0), created on first use; nullptr if no gfx950 device
hsrans_ctx *default_context();

// encode + sidecar index in one call (raw: one chain per resident wavefront, hsrans_index_boundaries; block_/mt_: a checkpoint
// every 64 groups inside the blocks); *pIndexLength receives the index size, index_capacity tells how much to provide;
// decode with that index, single launch or with the PCIe legs overlapped
#define HSRANS_DECL_INDEXED(codec, N)                                                                                                               \
  size_t codec##_index_capacity_##N(const size_t inputSize);                                                                                        \
  size_t codec##_encode_with_index_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, uint8_t *pIndex,   \
                                       const size_t indexCapacity, size_t *pIndexLength);                                                            \
  size_t codec##_decode_hip_with_index_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity,              \
                                           const uint8_t *pIndex, const size_t indexLength);                                                         \
  size_t codec##_decode_hip_pipelined_with_index_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity,    \
                                                     const uint8_t *pIndex, const size_t indexLength);

#define HSRANS_DECL_BITS(N)                                                                                                                        \
  size_t rANS32x32_16w_encode_scalar_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist); \
  size_t rANS32x64_16w_encode_scalar_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist); \
  size_t rANS32x32_16w_decode_scalar_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);               \
  size_t rANS32x64_16w_decode_scalar_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);               \
  size_t block_rANS32x32_16w_decode_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                \
  size_t block_rANS32x64_16w_decode_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                \
  size_t mt_rANS32x32_16w_decode_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                   \
  size_t mt_rANS32x64_16w_decode_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                   \
  size_t rANS32x32_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                 \
  size_t rANS32x64_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                 \
  size_t block_rANS32x32_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);           \
  size_t block_rANS32x64_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);           \
  size_t mt_rANS32x32_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);              \
  size_t mt_rANS32x64_16w_decode_auto_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);              \
  size_t rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x32_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x64_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                  \
  size_t block_rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);            \
  size_t block_rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);            \
  size_t mt_rANS32x32_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                     \
  size_t mt_rANS32x64_16w_encode_##N(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity);                     \
  size_t mt_rANS32x32_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);               \
  size_t mt_rANS32x64_16w_decode_hip_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);               \
  size_t mt_rANS32x32_16w_decode_mt_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity, thread_pool *pThreadPool); \
  size_t mt_rANS32x64_16w_decode_mt_##N(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity, thread_pool *pThreadPool); \
  HSRANS_DECL_INDEXED(rANS32x32_16w, N)                                                                                                             \
  HSRANS_DECL_INDEXED(rANS32x64_16w, N)                                                                                                             \
  HSRANS_DECL_INDEXED(block_rANS32x32_16w, N)                                                                                                       \
  HSRANS_DECL_INDEXED(block_rANS32x64_16w, N)                                                                                                       \
  HSRANS_DECL_INDEXED(mt_rANS32x32_16w, N)                                                                                                          \
  HSRANS_DECL_INDEXED(mt_rANS32x64_16w, N)

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
#undef HSRANS_DECL_INDEXED

} // namespace hsrans_hip

#endif // HSRANS_DROPIN_HPP
