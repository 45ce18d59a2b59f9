// The reference's function names/signatures (see include/hsrans_dropin.hpp) bound to the C ABI.
#include "../../include/hsrans_dropin.hpp"

#include <stdlib.h>

#include <mutex>
#include <thread>
#include <vector>

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

// Runtime dispatch (the role of block_rANS32x64_decode_wrapper, block_rANS32x64_16w_decode.cpp:130-152): one dependent chain ->
// host SIMD decoder; independent blocks -> GPU, or every host core when there is no GPU.
static size_t decode_auto(int container, int states, uint32_t bits, const uint8_t *in, size_t inLength, uint8_t *out, size_t outCapacity)
{
  if (container != HSRANS_MT)
    return hsrans_decode_cpu(-1, 1, container, states, bits, in, inLength, out, outCapacity, nullptr, 0);
  hsrans_ctx *ctx = default_context();
  if (ctx != nullptr)
    return hsrans_decode_host(ctx, container, states, bits, in, inLength, out, outCapacity, nullptr, 0);
  const unsigned hw = std::thread::hardware_concurrency();
  return hsrans_decode_cpu(-1, hw > 1 ? hw - 1 : 1, container, states, bits, in, inLength, out, outCapacity, nullptr, 0); // main.cpp:165 sizes its pool the same way
}

// ---- sidecar index ------------------------------------------------------------------------------------------------
static const uint32_t kBlockIndexInterval = 64; // block_/mt_: checkpoints every 64 groups inside a block (16 KiB at 64 states)

static size_t index_capacity_any(int container, int states, size_t n)
{
  if (container == HSRANS_RAW)
    return hsrans_plan_capacity_chains(container, states, n, 2 * 8192 + 64, 0);
  return hsrans_plan_capacity(container, states, n, kBlockIndexInterval, 0);
}

static size_t encode_indexed_any(int container, int states, uint32_t bits, const uint8_t *in, size_t length, uint8_t *out, size_t outCapacity, uint8_t *index,
                                 size_t indexCapacity, size_t *indexLength)
{
  if (indexLength == nullptr || index == nullptr)
    return 0;
  *indexLength = 0;
  hsrans_encode_opts opts{};
  opts.plan_out = index;
  opts.plan_capacity = indexCapacity;
  std::vector<uint64_t> groups;
  hsrans_hist hist;
  if (container == HSRANS_RAW)
  {
    // one chain per resident wavefront of the default device (or of an MI355X when no device is present at encode time)
    groups.resize(2 * 8192 + 64);
    const size_t n = hsrans_index_boundaries(default_context(), states, bits, length, groups.data(), groups.size());
    if (n == 0)
      opts.index_interval = 4; // a stream too short for more than one chain: any interval gives the one-chain plan
    else
    {
      opts.index_groups = groups.data();
      opts.n_index_groups = n;
    }
    hsrans_make_hist(&hist, in, length, bits); // as main.cpp:746 does for the raw codecs
  }
  else
    opts.index_interval = kBlockIndexInterval;
  const size_t r = hsrans_encode_ex(container, states, bits, in, length, out, outCapacity, container == HSRANS_RAW ? &hist : nullptr, &opts);
  *indexLength = r ? opts.plan_size : 0;
  return r;
}

static size_t decode_indexed_any(int container, int states, uint32_t bits, const uint8_t *in, size_t inLength, uint8_t *out, size_t outCapacity,
                                 const uint8_t *index, size_t indexLength, bool pipelined)
{
  hsrans_ctx *ctx = default_context();
  if (ctx == nullptr || index == nullptr)
    return 0;
  if (pipelined)
    return hsrans_decode_host_pipelined(ctx, container, states, bits, in, inLength, out, outCapacity, index, indexLength, 0 /* slices: by size */);
  return hsrans_decode_host(ctx, container, states, bits, in, inLength, out, outCapacity, index, indexLength);
}

size_t rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_RAW, 32, n); }
size_t rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_RAW, 64, n); }
size_t block_rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_BLOCK, 32, n); }
size_t block_rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_BLOCK, 64, n); }
size_t mt_rANS32x32_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_MT, 32, n); }
size_t mt_rANS32x64_16w_capacity(const size_t n) { return hsrans_capacity(HSRANS_MT, 64, n); }

#define HSRANS_DEF_INDEXED(codec, C, S, N)                                                                                                        \
  size_t codec##_index_capacity_##N(const size_t n) { return index_capacity_any(C, S, n); }                                                        \
  size_t codec##_encode_with_index_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, uint8_t *x, const size_t xc, size_t *xl)      \
  {                                                                                                                                                \
    return encode_indexed_any(C, S, N, i, l, o, c, x, xc, xl);                                                                                     \
  }                                                                                                                                                \
  size_t codec##_decode_hip_with_index_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, const uint8_t *x, const size_t xl)        \
  {                                                                                                                                                \
    return decode_indexed_any(C, S, N, i, l, o, c, x, xl, false);                                                                                  \
  }                                                                                                                                                \
  size_t codec##_decode_hip_pipelined_with_index_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, const uint8_t *x, const size_t xl) \
  {                                                                                                                                                \
    return decode_indexed_any(C, S, N, i, l, o, c, x, xl, true);                                                                                   \
  }

#define HSRANS_DEF_ONE(S, N)                                                                                                                      \
  size_t rANS32x##S##_16w_encode_scalar_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, const hist_t *h)                         \
  {                                                                                                                                                \
    return hsrans_encode(HSRANS_RAW, S, N, i, l, o, c, h);                                                                                         \
  }                                                                                                                                                \
  size_t rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_any(HSRANS_RAW, S, N, i, l, o, c); } \
  size_t rANS32x##S##_16w_decode_auto_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_RAW, S, N, i, l, o, c); } \
  /* the reference's own decoder names (rANS32x64_16w.h:48, block_rANS32x64_16w.h:19, mt_rANS32x64_16w.h:20): the runtime-dispatch route */            \
  size_t rANS32x##S##_16w_decode_scalar_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_RAW, S, N, i, l, o, c); } \
  size_t block_rANS32x##S##_16w_decode_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_BLOCK, S, N, i, l, o, c); } \
  size_t mt_rANS32x##S##_16w_decode_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_MT, S, N, i, l, o, c); } \
  size_t block_rANS32x##S##_16w_decode_auto_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_BLOCK, S, N, i, l, o, c); } \
  size_t mt_rANS32x##S##_16w_decode_auto_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_auto(HSRANS_MT, S, N, i, l, o, c); } \
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
  size_t mt_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return decode_any(HSRANS_MT, S, N, i, l, o, c); } \
  size_t mt_rANS32x##S##_16w_decode_mt_##N(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, thread_pool *) { return decode_auto(HSRANS_MT, S, N, i, l, o, c); } \
  HSRANS_DEF_INDEXED(rANS32x##S##_16w, HSRANS_RAW, S, N)                                                                                           \
  HSRANS_DEF_INDEXED(block_rANS32x##S##_16w, HSRANS_BLOCK, S, N)                                                                                   \
  HSRANS_DEF_INDEXED(mt_rANS32x##S##_16w, HSRANS_MT, S, N)

#define HSRANS_DEF_BITS(N) HSRANS_DEF_ONE(32, N) HSRANS_DEF_ONE(64, N)
HSRANS_DEF_BITS(10)
HSRANS_DEF_BITS(11)
HSRANS_DEF_BITS(12)
HSRANS_DEF_BITS(13)
HSRANS_DEF_BITS(14)
HSRANS_DEF_BITS(15)

} // namespace hsrans_hip

// ---- include/hsrans_names.h: the same entries with C linkage, hsrans_<reference name> ------------------------------------------
#include "../../include/hsrans_names.h"
extern "C" {
size_t hsrans_rANS32x32_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_RAW, 32, n); }
size_t hsrans_rANS32x64_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_RAW, 64, n); }
size_t hsrans_block_rANS32x32_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_BLOCK, 32, n); }
size_t hsrans_block_rANS32x64_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_BLOCK, 64, n); }
size_t hsrans_mt_rANS32x32_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_MT, 32, n); }
size_t hsrans_mt_rANS32x64_16w_capacity(size_t n) { return hsrans_capacity(HSRANS_MT, 64, n); }
#define HSRANS_C_DEF(S, N)                                                                                                                        \
  size_t hsrans_rANS32x##S##_16w_encode_scalar_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c, const hsrans_hist *h)                         \
  {                                                                                                                                                \
    return hsrans_hip::rANS32x##S##_16w_encode_scalar_##N(i, l, o, c, h);                                                                           \
  }                                                                                                                                                \
  size_t hsrans_rANS32x##S##_16w_decode_scalar_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::rANS32x##S##_16w_decode_scalar_##N(i, l, o, c); } \
  size_t hsrans_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::rANS32x##S##_16w_decode_hip_##N(i, l, o, c); } \
  size_t hsrans_block_rANS32x##S##_16w_encode_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::block_rANS32x##S##_16w_encode_##N(i, l, o, c); } \
  size_t hsrans_block_rANS32x##S##_16w_decode_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::block_rANS32x##S##_16w_decode_##N(i, l, o, c); } \
  size_t hsrans_block_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::block_rANS32x##S##_16w_decode_hip_##N(i, l, o, c); } \
  size_t hsrans_mt_rANS32x##S##_16w_encode_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::mt_rANS32x##S##_16w_encode_##N(i, l, o, c); } \
  size_t hsrans_mt_rANS32x##S##_16w_decode_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::mt_rANS32x##S##_16w_decode_##N(i, l, o, c); } \
  size_t hsrans_mt_rANS32x##S##_16w_decode_hip_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c) { return hsrans_hip::mt_rANS32x##S##_16w_decode_hip_##N(i, l, o, c); } \
  size_t hsrans_mt_rANS32x##S##_16w_decode_mt_##N(const uint8_t *i, size_t l, uint8_t *o, size_t c, void *pool)                                   \
  {                                                                                                                                                \
    return hsrans_hip::mt_rANS32x##S##_16w_decode_mt_##N(i, l, o, c, (thread_pool *)pool);                                                          \
  }
#define HSRANS_C_DEF_BITS(N) HSRANS_C_DEF(32, N) HSRANS_C_DEF(64, N)
HSRANS_C_DEF_BITS(10)
HSRANS_C_DEF_BITS(11)
HSRANS_C_DEF_BITS(12)
HSRANS_C_DEF_BITS(13)
HSRANS_C_DEF_BITS(14)
HSRANS_C_DEF_BITS(15)
} // extern "C"
