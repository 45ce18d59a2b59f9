// TEST INFRASTRUCTURE ONLY — never linked into or called from the product path.
//
// extern "C" adapter over the *real* reference (rainerzufalldererste/hypersonic-rANS @ 2024_10_08),
// whose sources are compiled where they lie under /root/reference/src by oracle/Makefile into
// oracle/_ref/libhsrans_ref.so.  Nothing from the reference is copied into this repository: this file
// only *calls* the reference's public functions (declared in its own headers, found via -I) so that
// Python tests can (1) pin oracle/hsrans_oracle.c against the reference, (2) generate golden vectors,
// and (3) time the reference's AVX2 decoders as `cpu_baseline.kind == "reference"` in bench.py.
//
// Reference entry points wrapped (file:line in /root/reference/src):
//   rANS32x{32,64}_16w_{capacity,encode_scalar_N,decode_scalar_N}       rANS32x64_16w.h:6-177, rANS32x32_16w.h
//   rANS32x{32,64}_xmmShfl2_16w_decode_avx2_var{A,C}_N                  (candidateForFastest, main.cpp:202-236)
//   rANS32x{32,64}_ymmShfl2_16w_decode_avx512_var{A,C}_N                (the AVX-512 family)
//   block_rANS32x{32,64}_16w_{capacity,encode_N,decode_N}               block_rANS32x64_16w.h:6-20
//   mt_rANS32x{32,64}_16w_{capacity,encode_N,decode_N,decode_mt_N}      mt_rANS32x64_16w.h:7-28
//   make_hist                                                           hist.h:68
#include "hist.h"
#include "rANS32x32_16w.h"
#include "rANS32x64_16w.h"
#include "block_rANS32x32_16w.h"
#include "block_rANS32x64_16w.h"
#include "mt_rANS32x32_16w.h"
#include "mt_rANS32x64_16w.h"
#include "thread_pool.h"

#include <string.h>

typedef size_t (*enc_hist_fn)(const uint8_t *, const size_t, uint8_t *, const size_t, const hist_t *);
typedef size_t (*codec_fn)(const uint8_t *, const size_t, uint8_t *, const size_t);
typedef size_t (*mt_fn)(const uint8_t *, const size_t, uint8_t *, const size_t, thread_pool *);

#define PER_BITS(prefix) { prefix##10, prefix##11, prefix##12, prefix##13, prefix##14, prefix##15 }

static const enc_hist_fn raw_enc[2][6] = { PER_BITS(rANS32x32_16w_encode_scalar_), PER_BITS(rANS32x64_16w_encode_scalar_) };
static const codec_fn raw_dec_scalar[2][6] = { PER_BITS(rANS32x32_16w_decode_scalar_), PER_BITS(rANS32x64_16w_decode_scalar_) };
// fastest AVX2 entries per main.cpp's registry: varC for bits <= 12, varA for bits >= 13
static const codec_fn raw_dec_avx2[2][6] = {
  { rANS32x32_xmmShfl2_16w_decode_avx2_varC_10, rANS32x32_xmmShfl2_16w_decode_avx2_varC_11, rANS32x32_xmmShfl2_16w_decode_avx2_varC_12,
    rANS32x32_xmmShfl2_16w_decode_avx2_varA_13, rANS32x32_xmmShfl2_16w_decode_avx2_varA_14, rANS32x32_xmmShfl2_16w_decode_avx2_varA_15 },
  { rANS32x64_xmmShfl2_16w_decode_avx2_varC_10, rANS32x64_xmmShfl2_16w_decode_avx2_varC_11, rANS32x64_xmmShfl2_16w_decode_avx2_varC_12,
    rANS32x64_xmmShfl2_16w_decode_avx2_varA_13, rANS32x64_xmmShfl2_16w_decode_avx2_varA_14, rANS32x64_xmmShfl2_16w_decode_avx2_varA_15 } };
// fastest AVX-512 entries per main.cpp's registry (:209-214, the `true` flags): ymmShfl2 varC for bits <= 12, ymmShfl2 varA for
// bits >= 13 (rANS32x64_16w.cpp:2108,3666); the 32-state codec has AVX-512 decoders for bits <= 12 only
static const codec_fn raw_dec_avx512[2][6] = {
  { rANS32x32_ymmShfl2_16w_decode_avx512_varC_10, rANS32x32_ymmShfl2_16w_decode_avx512_varC_11, rANS32x32_ymmShfl2_16w_decode_avx512_varC_12, nullptr, nullptr, nullptr },
  { rANS32x64_ymmShfl2_16w_decode_avx512_varC_10, rANS32x64_ymmShfl2_16w_decode_avx512_varC_11, rANS32x64_ymmShfl2_16w_decode_avx512_varC_12,
    rANS32x64_ymmShfl2_16w_decode_avx512_varA_13, rANS32x64_ymmShfl2_16w_decode_avx512_varA_14, rANS32x64_ymmShfl2_16w_decode_avx512_varA_15 } };
static const codec_fn blk_enc[2][6] = { PER_BITS(block_rANS32x32_16w_encode_), PER_BITS(block_rANS32x64_16w_encode_) };
static const codec_fn blk_dec[2][6] = { PER_BITS(block_rANS32x32_16w_decode_), PER_BITS(block_rANS32x64_16w_decode_) };
static const codec_fn mt_enc[2][6] = { PER_BITS(mt_rANS32x32_16w_encode_), PER_BITS(mt_rANS32x64_16w_encode_) };
static const codec_fn mt_dec[2][6] = { PER_BITS(mt_rANS32x32_16w_decode_), PER_BITS(mt_rANS32x64_16w_decode_) };
static const mt_fn mt_dec_mt[2][6] = { PER_BITS(mt_rANS32x32_16w_decode_mt_), PER_BITS(mt_rANS32x64_16w_decode_mt_) };

static thread_pool *g_pool = nullptr;

static inline bool sel(int states, int bits, int *si, int *bi)
{
  if ((states != 32 && states != 64) || bits < 10 || bits > 15)
    return false;
  *si = states == 64;
  *bi = bits - 10;
  return true;
}

extern "C"
{
  // container: 0 raw, 1 block_, 2 mt_
  size_t hsref_capacity(int container, int states, size_t n)
  {
    if (states == 64)
      return container == 0 ? rANS32x64_16w_capacity(n) : container == 1 ? block_rANS32x64_16w_capacity(n) : mt_rANS32x64_16w_capacity(n);
    return container == 0 ? rANS32x32_16w_capacity(n) : container == 1 ? block_rANS32x32_16w_capacity(n) : mt_rANS32x32_16w_capacity(n);
  }

  void hsref_make_hist(const uint8_t *data, size_t n, int bits, uint16_t counts[256], uint16_t cumul[256])
  {
    hist_t h;
    make_hist(&h, data, n, (size_t)bits);
    memcpy(counts, h.symbolCount, sizeof(h.symbolCount));
    memcpy(cumul, h.cumul, sizeof(h.cumul));
  }

  // raw: histogram is make_hist(whole input), as main.cpp:746 does. counts_or_null overrides it.
  size_t hsref_encode(int container, int states, int bits, const uint8_t *in, size_t n, uint8_t *out, size_t cap, const uint16_t *counts_or_null)
  {
    int si, bi;
    if (!sel(states, bits, &si, &bi))
      return 0;
    if (container == 0)
    {
      hist_t h;
      if (counts_or_null)
      {
        uint32_t c = 0;
        for (int i = 0; i < 256; i++)
        {
          h.symbolCount[i] = counts_or_null[i];
          h.cumul[i] = (uint16_t)c;
          c += counts_or_null[i];
        }
      }
      else
        make_hist(&h, in, n, (size_t)bits);
      return raw_enc[si][bi](in, n, out, cap, &h);
    }
    return (container == 1 ? blk_enc : mt_enc)[si][bi](in, n, out, cap);
  }

  // variant: 0 scalar (raw) / runtime-dispatched (block_, mt_ single thread); 1 fastest AVX2 (raw only); 2 mt_ on a thread pool;
  // 3 fastest AVX-512 (raw only; the caller checks hsref_has_avx512 first)
  size_t hsref_decode(int container, int states, int bits, int variant, const uint8_t *in, size_t in_len, uint8_t *out, size_t cap, int threads)
  {
    int si, bi;
    if (!sel(states, bits, &si, &bi))
      return 0;
    if (container == 0)
    {
      const codec_fn f = variant == 3 ? raw_dec_avx512[si][bi] : (variant == 1 ? raw_dec_avx2 : raw_dec_scalar)[si][bi];
      return f ? f(in, in_len, out, cap) : 0;
    }
    if (container == 1)
      return blk_dec[si][bi](in, in_len, out, cap);
    if (variant == 2)
    {
      if (g_pool == nullptr)
        g_pool = thread_pool_new(threads > 0 ? (size_t)threads : (thread_pool_max_threads() > 1 ? thread_pool_max_threads() - 1 : 1));
      return mt_dec_mt[si][bi](in, in_len, out, cap, g_pool);
    }
    return mt_dec[si][bi](in, in_len, out, cap);
  }

  size_t hsref_pool_threads(void) { return g_pool ? thread_pool_thread_count(g_pool) : 0; }

  int hsref_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

  int hsref_has_avx512(void)
  {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") ? 1 : 0;
  }
}
