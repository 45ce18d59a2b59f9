// The reference's function names/signatures (see include/hsrans_dropin.hpp) bound to the C ABI.
#include "../../include/hsrans_dropin.hpp"

#include <stdlib.h>

#include <mutex>

namespace hsrans_hip
{

void make_hist(hist_t *pHist, const uint8_t *pData, const size_t size, const size_t totalSymbolCountBits)
{
  hsrans_make_hist(pHist, pData, size, (uint32_t)totalSymbolCountBits);
}

hsrans_ctx *default_context()
{
  static std::once_flag once;
  static hsrans_ctx *ctx = nullptr;
  std::call_once(once, []() {
    const char *env = getenv("HSRANS_DEVICE");
    if (hsrans_ctx_create(env ? atoi(env) : 0, &ctx) != HSRANS_OK)
      ctx = nullptr;
  });
  return ctx;
}

static size_t decode_any(int container, int states, uint32_t bits, const uint8_t *in, size_t inLength, uint8_t *out, size_t outCapacity)
{
  hsrans_ctx *ctx = default_context();
  if (ctx == nullptr) // no CPU fallback: without a gfx950 device decoding fails, like every other error (return 0)
    return 0;
  return hsrans_decode_host(ctx, container, states, bits, in, inLength, out, outCapacity, nullptr, 0);
}

size_t rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_RAW, 32, n); }
size_t rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_RAW, 64, n); }
size_t block_rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_BLOCK, 32, n); }
size_t block_rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_BLOCK, 64, n); }
size_t mt_rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_MT, 32, n); }
size_t mt_rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_MT, 64, n); }

#define HSRANS_DEF_ONE(S, N)                                                                                                                      \
  size_t rANS32x##S##_16w_encode_scalar_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, const hist_t *h)                         \
  {                                                                                                                                                \
    return hsrans_encode(HSRANS_RAW, S, N, i, l, o, c, h);                                                                                         \
  }                                                                                                                                                \
  size_t rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_any(HSRANS_RAW, S, N, i, l, o, c); } \
  size_t block_rANS32x##S##_16w_encode_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c)                                           \
  {                                                                                                                                                \
    return hsrans_encode(HSRANS_BLOCK, S, N, i, l, o, c, nullptr);                                                                                 \
  }                                                                                                                                                \
  size_t block_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c)                                       \
  {                                                                                                                                                \
    return decode_any(HSRANS_BLOCK, S, N, i, l, o, c);                                                                                             \
  }                                                                                                                                                \
  size_t mt_rANS32x##S##_16w_encode_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c)                                              \
  {                                                                                                                                                \
    return hsrans_encode(HSRANS_MT, S, N, i, l, o, c, nullptr);                                                                                    \
  }                                                                                                                                                \
  size_t mt_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_any(HSRANS_MT, S, N, i, l, o, c); }

#define HSRANS_DEF_BITS(N) HSRANS_DEF_ONE(32, N) HSRANS_DEF_ONE(64, N)
HSRANS_DEF_BITS(10)
HSRANS_DEF_BITS(11)
HSRANS_DEF_BITS(12)
HSRANS_DEF_BITS(13)
HSRANS_DEF_BITS(14)
HSRANS_DEF_BITS(15)

} // namespace hsrans_hip
