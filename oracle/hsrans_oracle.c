/* TEST INFRASTRUCTURE ONLY — see hsrans_oracle.h.  Plain-C restatement of the reference algorithms for the
 * rANS32x32 / rANS32x64 16w path; each function cites the reference lines it follows (under /root/reference/src).
 * Parity pinned by tests/test_oracle_vs_ref.py (real reference in oracle/_ref) and tests/golden/ vectors. */
#include "hsrans_oracle.h"

#include <string.h>

#define CONSUME_POINT16 ((uint32_t)1 << 15) /* rans.h:8 DecodeConsumePoint16 */

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

/* rANS32x64_16w.cpp:210-216: the table is the bit permutation j -> (j&0x23) | ((j&4)<<2) | ((j&0x18)>>1);
 * it is pinned through the golden streams (tests/test_oracle_golden.py): any wrong entry scrambles every decoded group. */
uint8_t orc_idx2idx(unsigned j)
{
  return (uint8_t)((j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1));
}

/* hist.cpp:8-14 */
void orc_observe_hist(uint32_t hist[256], const uint8_t *data, size_t size)
{
  memset(hist, 0, sizeof(uint32_t) * 256);
  for (size_t i = 0; i < size; i++)
    hist[data[i]]++;
}

/* hist.cpp:107-138: max-heap sift-down on an index array keyed by val[] (left child tested first, strict >) */
static void sift_down(uint8_t *idx, const uint16_t *val, int64_t n, int64_t i)
{
  for (;;)
  {
    const int64_t left = 2 * i + 1, right = 2 * i + 2;
    int64_t largest = i;
    if (left < n && val[idx[left]] > val[idx[largest]])
      largest = left;
    if (right < n && val[idx[right]] > val[idx[largest]])
      largest = right;
    if (largest == i)
      return;
    const uint8_t t = idx[i];
    idx[i] = idx[largest];
    idx[largest] = t;
    i = largest;
  }
}

/* hist.cpp:16-215.  Float arithmetic is IEEE single without contraction (the shipped oracle/_ref build uses no
 * -ffast-math either; the reference's own project file does, project.lua:81 — encoder-side only, SURVEY §7 hard part 5). */
void orc_normalize_hist(orc_hist_t *out, const uint32_t hist[256], size_t dataBytes, unsigned bits)
{
  const uint32_t total = (uint32_t)1 << bits;
  uint16_t capped[256];
  size_t cappedSum = 0;

  /* hist.cpp:60-70 */
  const float mul = (float)total / (float)dataBytes;
  for (size_t i = 0; i < 256; i++)
  {
    volatile float scaled = (float)hist[i] * mul; /* volatile: forbid fused multiply-add */
    capped[i] = (uint16_t)(scaled + 0.5f);
    if (capped[i] == 0 && hist[i])
      capped[i] = 1;
    cappedSum += capped[i];
  }

  if (cappedSum != total)
  {
    /* hist.cpp:103-140: heap sort of symbol indices, ascending by capped count */
    uint8_t sorted[256];
    for (size_t i = 0; i < 256; i++)
      sorted[i] = (uint8_t)i;
    for (int64_t i = 256 / 2 - 1; i >= 0; i--)
      sift_down(sorted, capped, 256, i);
    for (int64_t i = 255; i >= 0; i--)
    {
      const uint8_t t = sorted[0];
      sorted[0] = sorted[i];
      sorted[i] = t;
      sift_down(sorted, capped, i, 0);
    }

    /* hist.cpp:141-150 */
    size_t minTwo = 0;
    for (size_t i = 0; i < 256; i++)
      if (capped[sorted[i]] >= 2) { minTwo = i; break; }

    /* hist.cpp:152-174: steal one from every symbol with count >= 2, smallest first, round-robin */
    while (cappedSum > total)
    {
      for (size_t i = minTwo; i < 256; i++)
      {
        capped[sorted[i]]--;
        cappedSum--;
        if (cappedSum == total)
          goto ready;
      }
      for (size_t i = minTwo; i < 256; i++)
        if (capped[sorted[i]] >= 2) { minTwo = i; break; }
    }

    /* hist.cpp:176-198: give one to every symbol with count >= 2, largest first, round-robin */
    while (cappedSum < total)
    {
      for (int64_t i = 255; i >= (int64_t)minTwo; i--)
      {
        capped[sorted[i]]++;
        cappedSum++;
        if (cappedSum == total)
          goto ready;
      }
      for (size_t i = minTwo; i < 256; i++)
        if (capped[sorted[i]] >= 2) { minTwo = i; break; }
    }
  }

ready:; /* hist.cpp:201-209 */
  size_t counter = 0;
  for (size_t i = 0; i < 256; i++)
  {
    out->cumul[i] = (uint16_t)counter;
    out->symbolCount[i] = capped[i];
    counter += capped[i];
  }
}

/* hist.cpp:217-222 */
void orc_make_hist(orc_hist_t *out, const uint8_t *data, size_t size, unsigned bits)
{
  uint32_t hist[256];
  orc_observe_hist(hist, data, size);
  orc_normalize_hist(out, hist, size, bits);
}

/* hist.cpp:326-354 (uint16_t sum) / hist.cpp:308-324 (uint32_t sum) */
int orc_make_dec_table(unsigned bits, const uint16_t counts[256], uint16_t cumul[256], uint8_t *cumulInv, int wide_sum)
{
  const uint32_t total = (uint32_t)1 << bits;
  uint32_t c32 = 0;
  uint16_t c16 = 0;
  for (size_t i = 0; i < 256; i++)
  {
    cumul[i] = wide_sum ? (uint16_t)c32 : c16;
    c32 += counts[i];
    c16 = (uint16_t)(c16 + counts[i]);
  }
  if (wide_sum ? (c32 != total) : ((uint32_t)c16 != total))
    return 0;

  /* hist.cpp:343-351: slot -> symbol, skipping zero-count symbols */
  uint8_t sym = 0;
  for (uint32_t i = 0; i < total; i++)
  {
    while (sym != 0xFF && (!counts[sym] || cumul[sym + 1] <= i))
      sym++;
    cumulInv[i] = sym;
  }
  return 1;
}

size_t orc_capacity(int container, int states, size_t n)
{
  const size_t S = (size_t)states;
  if (container == ORC_RAW) /* rANS32x64_16w.cpp:10-13 */
    return n + S + 2 * 256 + 4 * S + 8 * 2;
  /* block_rANS32x64_16w_encode.cpp:47-54 / mt_rANS32x64_16w_encode.cpp:50-57 (MinMinBlockSize = 1<<15) */
  const size_t base = 2 * 8 + 256 * 2 + n + S * 4;
  const size_t blockCount = (n + ((size_t)1 << 15)) / ((size_t)1 << 15) + 1;
  const size_t perBlock = container == ORC_BLOCK ? (8 + 256 * 2) : (8 * 2 + 256 * 2 + S * 4);
  return base + blockCount * perBlock;
}

/* one encode step, rANS32x64_16w.cpp:71-97 */
static inline void enc_put(uint32_t *state, uint8_t sym, const orc_hist_t *h, unsigned bits, uint16_t **pStart)
{
  const uint32_t emitPoint = (CONSUME_POINT16 >> bits) << 16; /* :41 EncodeEmitPoint */
  const uint32_t freq = h->symbolCount[sym];
  const uint32_t max = emitPoint * freq;
  uint32_t x = *state;
  if (x >= max)
  {
    st16((uint8_t *)*pStart, (uint16_t)(x & 0xFFFF));
    (*pStart)--;
    x >>= 16;
  }
  *state = ((x / freq) << bits) + (uint32_t)h->cumul[sym] + (x % freq);
}

/* rANS32x64_16w.cpp:34-166 / rANS32x32_16w.cpp:34-159 */
size_t orc_raw_encode(int states, unsigned bits, const uint8_t *in, size_t n, uint8_t *out, size_t cap, const orc_hist_t *hist)
{
  const int64_t S = states;
  if ((S != 32 && S != 64) || bits < 10 || bits > 15 || n == 0) /* n == 0 reads in[-64..] in the reference: undefined */
    return 0;
  if (cap < orc_capacity(ORC_RAW, states, n)) /* :37 */
    return 0;

  uint32_t st[64];
  uint16_t *pEnd = (uint16_t *)(out + cap - sizeof(uint16_t)); /* :44 */
  uint16_t *pStart = pEnd;
  for (int64_t j = 0; j < S; j++)
    st[j] = CONSUME_POINT16; /* :48-49 */

  int64_t i = (int64_t)n - 1; /* :61-63 */
  i &= ~(S - 1);
  i += S;

  for (int64_t j = S - 1; j >= 0; j--) /* :65-99 final partial group */
  {
    const int64_t pos = i - S + orc_idx2idx((unsigned)j);
    if (pos < (int64_t)n)
      enc_put(&st[j], in[pos], hist, bits, &pStart);
  }
  i -= S;

  for (; i >= S; i -= S) /* :102-135 */
    for (int64_t j = S - 1; j >= 0; j--)
      enc_put(&st[j], in[i - S + orc_idx2idx((unsigned)j)], hist, bits, &pStart);

  size_t o = 0; /* :137-165 */
  st64(out + o, (uint64_t)n);
  o += 8;
  o += 8;
  for (size_t j = 0; j < 256; j++, o += 2)
    st16(out + o, hist->symbolCount[j]);
  for (int64_t j = 0; j < S; j++, o += 4)
    st32(out + o, st[j]);
  const size_t size = (size_t)(pEnd - pStart) * sizeof(uint16_t);
  memmove(out + o, pStart + 1, size);
  o += size;
  st64(out + 8, (uint64_t)o);
  return o;
}

typedef struct
{
  uint32_t states[64];
  const uint8_t *rd; /* read head into the uint16 stream */
  uint16_t counts[256], cumul[256];
  uint8_t cumulInv[1 << 15];
} dec_ctx_t;

/* decode_symbol_scalar_32x64_16w rANS32x64_16w.cpp:17-30 + renormalisation :241-245 */
static inline uint8_t dec_get(dec_ctx_t *c, unsigned j, unsigned bits)
{
  const uint32_t M = (uint32_t)1 << bits;
  uint32_t x = c->states[j];
  const uint32_t slot = x & (M - 1);
  const uint8_t sym = c->cumulInv[slot];
  x = (x >> bits) * (uint32_t)c->counts[sym] + slot - (uint32_t)c->cumul[sym];
  if (x < CONSUME_POINT16)
  {
    x = x << 16 | ld16(c->rd);
    c->rd += 2;
  }
  c->states[j] = x;
  return sym;
}

/* block_codec64.h:173-217 / block_codec32.h:162-210: whole groups from `start` until i >= end */
static size_t dec_section(dec_ctx_t *c, unsigned S, unsigned bits, uint8_t *out, size_t start, size_t end)
{
  size_t i = start;
  for (; i < end; i += S)
    for (unsigned j = 0; j < S; j++)
      out[i + orc_idx2idx(j)] = dec_get(c, j, bits);
  return i;
}

/* final partial group, rANS32x64_16w.cpp:252-280 */
static void dec_tail(dec_ctx_t *c, unsigned S, unsigned bits, uint8_t *out, size_t i, size_t outLen)
{
  for (unsigned j = 0; j < S; j++)
    if (i + orc_idx2idx(j) < outLen)
      out[i + orc_idx2idx(j)] = dec_get(c, j, bits);
}

/* rANS32x64_16w.cpp:168-283 / rANS32x32_16w.cpp:161-269 */
size_t orc_raw_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  const unsigned S = (unsigned)states;
  if ((S != 32 && S != 64) || bits < 10 || bits > 15)
    return 0;
  if (inLen < 8 * 2 + 4 * (size_t)S + 2 * 256) /* :171 */
    return 0;
  const uint64_t outLen = ld64(in); /* :176-181 */
  if (outLen > outCap)
    return 0;
  const uint64_t expIn = ld64(in + 8); /* :183-187 */
  if (inLen < expIn)
    return 0;

  static _Thread_local dec_ctx_t c;
  size_t o = 16;
  for (size_t k = 0; k < 256; k++, o += 2) /* :191-195 */
    c.counts[k] = ld16(in + o);
  if (!orc_make_dec_table(bits, c.counts, c.cumul, c.cumulInv, 0)) /* :197 */
    return 0;
  for (unsigned j = 0; j < S; j++, o += 4) /* :200-206 */
    c.states[j] = ld32(in + o);
  c.rd = in + o; /* :208 */

  /* :220 `outLen - StateCount + 1` underflows for outLen < S-1 in the reference (undefined: it walks off the
   * buffers).  The oracle defines that case the natural way: zero whole groups, then the masked tail. */
  const size_t whole = outLen + 1 >= S ? (size_t)(outLen - S + 1) : 0;
  size_t i = dec_section(&c, S, bits, out, 0, whole); /* :223-250 */
  dec_tail(&c, S, bits, out, i, (size_t)outLen);      /* :252-280 */
  return (size_t)outLen;
}

/* shared body of block_rANS32x64_16w_decode.cpp:12-126 (is_mt = 0) and mt_rANS32x64_16w_decode.cpp:12-133 (is_mt = 1).
 * The SIMD dispatch wrappers (block_…decode.cpp:130-152, mt_…decode.cpp:269-297) build their tables through
 * inplace_complete_hist (uint32_t sum), hence wide_sum = 1 here. */
static size_t container_decode(int is_mt, int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  const unsigned S = (unsigned)states;
  if ((S != 32 && S != 64) || bits < 10 || bits > 15)
    return 0;
  if (inLen < 8 * 2 + 4 * (size_t)S + 2 * 256) /* :15 */
    return 0;
  const uint64_t outLen = ld64(in); /* :22-26 */
  if (outLen > outCap)
    return 0;
  const uint64_t expIn = ld64(in + 8); /* :28-32 */
  if (inLen < expIn)
    return 0;
  /* `outLen - StateCount + 1` (block :43 / mt :37) underflows for outLen < S-1: undefined in the reference.
   * Streams that short are rejected here. */
  if (outLen + 1 < S)
    return 0;

  static _Thread_local dec_ctx_t c;
  size_t o = 16;
  if (!is_mt)
    for (unsigned j = 0; j < S; j++, o += 4) /* block :36-40 */
      c.states[j] = ld32(in + o);
  c.rd = in + o;
  memset(c.counts, 0, sizeof(c.counts)); /* `hist_t hist = {}` block :45 / mt :39 */

  const size_t whole = (size_t)(outLen - S + 1);
  size_t i = 0;
  do
  {
    const uint64_t blockSizeVal = ld64(c.rd); /* block :49-50 / mt :43-44 */
    c.rd += 8;
    if (blockSizeVal & ((uint64_t)1 << 63)) /* single-symbol block: block :52-60 / mt :46-54 */
    {
      const uint8_t symbol = (uint8_t)((blockSizeVal >> 54) & 0xFF);
      const uint64_t blockSize = blockSizeVal & (((uint64_t)1 << 54) - 1);
      if (blockSize > outCap - i) /* the reference memsets unchecked; the oracle refuses to write out of bounds */
        return 0;
      memset(out + i, symbol, (size_t)blockSize);
      i += (size_t)blockSize;
    }
    else
    {
      const uint8_t *after = NULL;
      if (is_mt) /* mt :57-66 */
      {
        const uint64_t skip = ld64(c.rd);
        c.rd += 8;
        after = c.rd + 2 * (skip + 1);
        for (unsigned j = 0; j < S; j++, c.rd += 4)
          c.states[j] = ld32(c.rd);
      }
      for (size_t k = 0; k < 256; k++, c.rd += 2) /* block :63-67 / mt :68-72 */
        c.counts[k] = ld16(c.rd);
      if (!orc_make_dec_table(bits, c.counts, c.cumul, c.cumulInv, 1)) /* block :69 / mt :74 */
        return 0;
      uint64_t blockEnd = i + blockSizeVal; /* block :72-77 / mt :77-82 */
      if (blockEnd > whole)
        blockEnd = whole;
      else if ((blockEnd & (S - 1)) != 0)
        return 0;
      i = dec_section(&c, S, bits, out, i, (size_t)blockEnd); /* block :79 / mt :84 */

      if (is_mt)
      {
        if (i > whole) /* mt :86-92 */
        {
          if (i >= outLen)
            return (size_t)outLen;
          break;
        }
        c.rd = after; /* mt :94 */
      }
    }
    if (!is_mt && i > whole) /* block :82-88 (outside the else-branch in block_, inside it in mt_) */
    {
      if (i >= outLen)
        return (size_t)outLen;
      break;
    }
  } while (i < whole);

  if (i < outLen) /* block :92-123 / mt :99-130: last partial group with the most recently read histogram */
  {
    if (!orc_make_dec_table(bits, c.counts, c.cumul, c.cumulInv, 0))
      return 0;
    dec_tail(&c, S, bits, out, i, (size_t)outLen);
  }
  return (size_t)outLen;
}

size_t orc_block_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  return container_decode(0, states, bits, in, inLen, out, outCap);
}

size_t orc_mt_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  return container_decode(1, states, bits, in, inLen, out, outCap);
}

size_t orc_decode(int container, int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  switch (container)
  {
  case ORC_RAW: return orc_raw_decode(states, bits, in, inLen, out, outCap);
  case ORC_BLOCK: return orc_block_decode(states, bits, in, inLen, out, outCap);
  case ORC_MT: return orc_mt_decode(states, bits, in, inLen, out, outCap);
  default: return 0;
  }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Plan interpreter (test infrastructure): executes a product decode plan ("HSRPLAN1", hypersonic_rans_amd/csrc/
 * hsrans_plan.h) chain by chain with the scalar step above.  It lets the CPU test-suite check the host planner and the
 * encoders' sidecar plans without a GPU; the GPU tests compare the kernels against the same oracle.
 * The layout constants are restated here on purpose (the oracle shares no code with the product).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct
{
  uint64_t words_off, out_off, hist_off, fill_len;
  uint32_t steps;
  uint16_t tail, flags;
  uint32_t state_idx, reserved;
} orc_piece_t;

size_t orc_exec_plan(const uint8_t *plan, size_t planLen, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap)
{
  if (planLen < 64 || memcmp(plan, "HSRPLAN1", 8) != 0)
    return 0;
  const uint32_t S = ld32(plan + 12), bits = ld32(plan + 16), flags = ld32(plan + 20);
  const uint64_t outLen = ld64(plan + 24);
  const uint32_t nChains = ld32(plan + 40), nPieces = ld32(plan + 44);
  if (flags & 1u) /* walk plans carry no pieces: nothing to interpret */
    return 0;
  if (outLen > outCap)
    return 0;
  const size_t piecesOff = 64 + (((size_t)nChains + 1) * 4 + 15) / 16 * 16;
  const size_t statesOff = piecesOff + (size_t)nPieces * sizeof(orc_piece_t);
  if (statesOff + (size_t)nChains * S * 4 + ((flags & 4u) ? 512 : 0) != planLen) /* optional histogram copy */
    return 0;
  static _Thread_local dec_ctx_t c;
  uint64_t haveHist = ~(uint64_t)0;
  for (uint32_t ch = 0; ch < nChains; ch++)
  {
    const uint32_t first = ld32(plan + 64 + 4 * (size_t)ch), last = ld32(plan + 64 + 4 * ((size_t)ch + 1));
    for (uint32_t pi = first; pi < last; pi++)
    {
      orc_piece_t p;
      memcpy(&p, plan + piecesOff + (size_t)pi * sizeof(p), sizeof(p));
      if (p.flags & 1u)
        for (uint32_t j = 0; j < S; j++)
          c.states[j] = ld32(plan + statesOff + ((size_t)p.state_idx * S + j) * 4);
      if (p.flags & 2u)
      {
        if (p.out_off + p.fill_len > outLen)
          return 0;
        memset(out + p.out_off, (int)(p.hist_off & 0xFF), (size_t)p.fill_len);
        continue;
      }
      if (p.hist_off + 512 > inLen || p.words_off > inLen)
        return 0;
      if (p.hist_off != haveHist)
      {
        for (size_t k = 0; k < 256; k++)
          c.counts[k] = ld16(in + p.hist_off + 2 * k);
        if (!orc_make_dec_table(bits, c.counts, c.cumul, c.cumulInv, 1))
          return 0;
        haveHist = p.hist_off;
      }
      c.rd = in + p.words_off;
      const size_t end = (size_t)p.out_off + (size_t)p.steps * S;
      if (end + p.tail > outLen)
        return 0;
      dec_section(&c, S, bits, out, (size_t)p.out_off, end);
      dec_tail(&c, S, bits, out, end, end + p.tail);
    }
  }
  return (size_t)outLen;
}
