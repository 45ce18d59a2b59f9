// Host-side format layer of the MI355X hypersonic-rANS decode path: capacity, histogram normalisation, the scalar
// encoders that produce the reference's byte formats (and, optionally, a sidecar decode plan), and the planner that
// turns a stream into the list of chains the GPU kernels execute.  No decoding happens on the host.
//
// Wire formats (little-endian, 2-byte aligned; SURVEY.md §8, reference files under /root/reference/src):
//   raw    : u64 decodedLen | u64 streamLen | u16 count[256] | u32 state[S] | u16 words...   (rANS32x64_16w.cpp:137-165)
//   block_ : u64 decodedLen | u64 streamLen | u32 state[S] | { u64 size ; u16 count[256] ; words }...   with bit 63 of
//            size marking a single-symbol block (symbol in bits 54..61, no counts/words) (block_rANS32x64_16w_encode.cpp:256-373)
//   mt_    : u64 decodedLen | u64 streamLen | { u64 size ; u64 skip ; u32 state[S] ; u16 count[256] ; words }...
//            next header = &state[0] + 2*(skip+1) bytes                                  (mt_rANS32x64_16w_encode.cpp:266-298)
#include "hsrans_host.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace hsrans
{

namespace
{
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline void wr64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

// lane -> byte position inside a group (reference table rANS32x64_16w.cpp:210-216, generated arithmetically)
inline uint32_t lane_to_byte(uint32_t j) { return (j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1); }
} // namespace

// ---------------------------------------------------------------------------------------------------------------
// capacity  (rANS32x64_16w.cpp:10-13, block_rANS32x64_16w_encode.cpp:47-54, mt_rANS32x64_16w_encode.cpp:50-57)
// ---------------------------------------------------------------------------------------------------------------
size_t capacity(int container, int states, size_t n)
{
  const size_t S = (size_t)states;
  const size_t header = 16 + 512 + 4 * S;
  if (container == HSRANS_RAW)
    return n + S + header;
  const size_t blocks = (n + 32768) / 32768 + 1; // the reference sizes for 32 KiB blocks
  const size_t per_block = container == HSRANS_BLOCK ? 8 + 512 : 16 + 512 + 4 * S;
  return header + n + blocks * per_block;
}

// ---------------------------------------------------------------------------------------------------------------
// histogram normalisation: counts scaled to sum 2^bits, every present symbol keeps >= 1 (hist.cpp:16-215)
// ---------------------------------------------------------------------------------------------------------------
namespace
{
// textbook heap sort of symbol ids by ascending count; the tie order it produces decides which symbols are adjusted
void heap_sift(uint8_t *ids, const uint16_t *key, int n, int root)
{
  while (true)
  {
    int big = root;
    const int l = 2 * root + 1, r = l + 1;
    if (l < n && key[ids[l]] > key[ids[big]])
      big = l;
    if (r < n && key[ids[r]] > key[ids[big]])
      big = r;
    if (big == root)
      break;
    std::swap(ids[root], ids[big]);
    root = big;
  }
}

int first_at_least_two(const uint8_t *ids, const uint16_t *key, int from, int fallback)
{
  for (int i = from; i < 256; i++)
    if (key[ids[i]] >= 2)
      return i;
  return fallback;
}
} // namespace

void normalize_counts(hsrans_hist *hist, const uint32_t raw[256], size_t data_bytes, uint32_t bits)
{
  const uint32_t target = 1u << bits;
  uint16_t scaled[256];
  uint64_t sum = 0;
  const float factor = (float)target / (float)data_bytes;
  for (int s = 0; s < 256; s++)
  {
    volatile float v = (float)raw[s] * factor; // single rounding per operation (no fused multiply-add)
    uint16_t c = (uint16_t)(v + 0.5f);
    if (c == 0 && raw[s] != 0)
      c = 1;
    scaled[s] = c;
    sum += c;
  }

  if (sum != target)
  {
    uint8_t order[256];
    for (int s = 0; s < 256; s++)
      order[s] = (uint8_t)s;
    for (int i = 127; i >= 0; i--)
      heap_sift(order, scaled, 256, i);
    for (int i = 255; i >= 0; i--)
    {
      std::swap(order[0], order[i]);
      heap_sift(order, scaled, i, 0);
    }

    int lo = first_at_least_two(order, scaled, 0, 0);
    while (sum > target) // take one from every symbol that can spare it, smallest first, as often as needed
    {
      bool done = false;
      for (int i = lo; i < 256 && !done; i++)
      {
        scaled[order[i]]--;
        done = --sum == target;
      }
      if (!done)
        lo = first_at_least_two(order, scaled, lo, lo);
    }
    while (sum < target) // hand one to every symbol with count >= 2, largest first
    {
      bool done = false;
      for (int i = 255; i >= lo && !done; i--)
      {
        scaled[order[i]]++;
        done = ++sum == target;
      }
      if (!done)
        lo = first_at_least_two(order, scaled, lo, lo);
    }
  }

  uint32_t run = 0;
  for (int s = 0; s < 256; s++)
  {
    hist->symbolCount[s] = scaled[s];
    hist->cumul[s] = (uint16_t)run;
    run += scaled[s];
  }
}

void make_hist(hsrans_hist *hist, const uint8_t *data, size_t size, uint32_t bits)
{
  uint32_t raw[256] = {};
  for (size_t i = 0; i < size; i++)
    raw[data[i]]++;
  normalize_counts(hist, raw, size, bits);
}

// ---------------------------------------------------------------------------------------------------------------
// scalar encoders.  rANS encodes backwards: symbols last-to-first, uint16 words written from the end of `out`
// towards the front, then the finished payload is moved behind the header (rANS32x64_16w.cpp:34-166).
// ---------------------------------------------------------------------------------------------------------------
namespace
{

struct Checkpoint
{
  uint64_t group;          // absolute group index the decoder is about to start when it is in this state
  uint64_t words_from_end; // bytes between the decoder's read cursor at that moment and the end of the stream
  uint64_t hist_from_end;  // bytes between the active histogram's counts and the end of the stream (block_/mt_)
  uint32_t states[64];
};

struct Coder
{
  uint32_t S, bits;
  uint32_t x[64];
  uint8_t *begin; // start of the output buffer
  uint8_t *end;   // one past the last byte of the output buffer
  uint8_t *p;     // lowest byte written so far
  size_t reserve; // bytes the front header will need
  bool overflow;  // the payload ran into the front of the buffer (histogram does not fit the data)
  hsrans_hist hist;

  void init(uint32_t S_, uint32_t bits_, uint8_t *out, size_t cap)
  {
    S = S_;
    bits = bits_;
    begin = out;
    end = p = out + cap;
    overflow = false;
    for (uint32_t j = 0; j < 64; j++)
      x[j] = kConsumePoint16;
  }
  size_t written() const { return (size_t)(end - p); }
  void push_bytes(const void *src, size_t n)
  {
    if ((size_t)(p - begin) < n + reserve) // keep room for the front header
    {
      overflow = true;
      return;
    }
    p -= n;
    memcpy(p, src, n);
  }
  // encode the symbols of one group that exist (pos < n), highest lane first (rANS32x64_16w.cpp:65-99)
  void put_group(const uint8_t *in, size_t group_start, size_t n)
  {
    const uint32_t emit_scale = (kConsumePoint16 >> bits) << 16;
    for (int j = (int)S - 1; j >= 0; j--)
    {
      const size_t pos = group_start + lane_to_byte((uint32_t)j);
      if (pos >= n)
        continue;
      const uint8_t sym = in[pos];
      const uint32_t freq = hist.symbolCount[sym];
      if (freq == 0) // symbol missing from the histogram: not encodable
      {
        overflow = true;
        return;
      }
      uint32_t v = x[j];
      if (v >= emit_scale * freq)
      {
        const uint16_t w = (uint16_t)v;
        push_bytes(&w, 2);
        v >>= 16;
      }
      x[j] = ((v / freq) << bits) + hist.cumul[sym] + (v % freq);
    }
  }
};

struct BlockSpan
{
  size_t begin, end;
  bool single;
  hsrans_hist hist; // histogram the block is coded with (unused for single-symbol blocks)
};

// fixed-size blocks; the last one absorbs a remainder shorter than S so that the reference decoders' loop condition
// (`i < outLen - S + 1`, block_…decode.cpp:90) always reaches its header
std::vector<BlockSpan> split_blocks(const uint8_t *in, size_t n, size_t block, uint32_t S, uint32_t bits)
{
  std::vector<BlockSpan> v;
  size_t count = (n + block - 1) / block;
  if (count > 1 && n - (count - 1) * block < S)
    count--;
  for (size_t b = 0; b < count; b++)
  {
    BlockSpan s;
    s.begin = b * block;
    s.end = b + 1 == count ? n : (b + 1) * block;
    s.single = true;
    for (size_t i = s.begin + 1; i < s.end && s.single; i++)
      s.single = in[i] == in[s.begin];
    if (!s.single)
    {
      uint32_t raw[256] = {};
      for (size_t i = s.begin; i < s.end; i++)
        raw[in[i]]++;
      normalize_counts(&s.hist, raw, s.end - s.begin, bits);
    }
    v.push_back(s);
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's adaptive block policy (block_rANS32x64_16w_encode.cpp:137-349, mt_rANS32x64_16w_encode.cpp:140-356 and
// their 32-state twins), restated: blocks are chosen back to front in units of MinBlockSize; a block keeps growing
// towards the front while a cost model (order-0 code length under the block's histogram vs. a fresh histogram for the
// next unit) says one histogram is cheaper than two; runs of one symbol become single-symbol blocks.
// Reproduces the reference's choices byte for byte, including two things that look like accidents: every block but the
// last is modelled on the bytes up to the END of the block behind it (`blockBackPoint` is updated late, :344), and the
// mt_ size cap is measured against that stale end too.  One deliberate deviation: the reference can leave a last block
// shorter than S symbols (MinBlockSize < n < MinBlockSize + S), which its own decoders then mis-decode (SURVEY.md §8
// quirks); here that remainder is merged into the block before it.
// ---------------------------------------------------------------------------------------------------------------
struct Policy
{
  uint32_t min_block_bits;
  uint32_t replace_mul; // HistReplaceMul
  size_t max_block;     // mt_: 1 << 25; block_: unlimited
};

Policy reference_policy(int container, uint32_t S, uint32_t bits)
{
  Policy p;
  if (container == HSRANS_MT) // mt_rANS32x64_16w_encode.cpp:21-48 (same in the 32-state file)
  {
    p.min_block_bits = 16;
    p.replace_mul = bits == 15 ? 50 : 500;
    p.max_block = (size_t)1 << 25;
    return p;
  }
  // block_rANS32x64_16w_encode.cpp:21-39 / block_rANS32x32_16w_encode.cpp:21-39
  static const uint32_t mul64[6] = {4000, 7730, 5600, 2500, 1500, 850}, mul32[6] = {4000, 7730, 5600, 3120, 2087, 822};
  static const uint32_t min64[6] = {20, 19, 16, 17, 17, 16}, min32[6] = {20, 19, 15, 17, 17, 18};
  p.replace_mul = (S == 64 ? mul64 : mul32)[bits - 10];
  p.min_block_bits = (S == 64 ? min64 : min32)[bits - 10];
  p.max_block = ~(size_t)0;
  return p;
}

// _CanExtendHist (mt_…encode.cpp:61-136): would coding [start, start+size) with `old` cost less than 8 x replace_mul/4096
// x 2^bits bits more than with a histogram of its own?  `counts` receives the unit's raw counts.
bool can_extend(const uint8_t *in, size_t start, size_t size, const hsrans_hist &old, uint32_t bits, const Policy &pol, uint32_t S)
{
  uint32_t counts[256] = {};
  for (size_t i = start; i < start + size; i++)
    counts[in[i]]++;
  hsrans_hist fresh;
  normalize_counts(&fresh, counts, (size_t)1 << pol.min_block_bits, bits);
  const float total = (float)(1u << bits);
  const size_t replace_point = ((size_t)(1u << bits) * pol.replace_mul) >> 12;
  float cost_old = 0;
  float cost_new = (float)(2 * 256 + S * 4 + 8 * 2) * 0.5f;
  for (int s = 0; s < 256; s++)
  {
    if (counts[s] == 0)
      continue;
    const float before = (float)(counts[s] - 1) * log2f((float)old.symbolCount[s] / total);
    const float after = (float)counts[s] * log2f((float)fresh.symbolCount[s] / total);
    cost_old -= before;
    cost_new -= after;
  }
  return cost_old - cost_new < (float)replace_point;
}

void count_range(uint32_t counts[256], const uint8_t *in, size_t begin, size_t end, uint32_t *distinct, uint8_t *last_symbol)
{
  memset(counts, 0, 256 * sizeof(uint32_t));
  for (size_t i = begin; i < end; i++)
    counts[in[i]]++;
  *distinct = 0;
  for (int s = 0; s < 256; s++)
    if (counts[s])
    {
      (*distinct)++;
      *last_symbol = (uint8_t)s;
    }
}

std::vector<BlockSpan> reference_blocks(int container, const uint8_t *in, size_t n, uint32_t S, uint32_t bits)
{
  const Policy pol = reference_policy(container, S, bits);
  const size_t unit = (size_t)1 << pol.min_block_bits;
  std::vector<BlockSpan> back_to_front;
  uint32_t counts[256];
  uint32_t distinct;
  uint8_t symbol = 0;

  size_t target = (((n - 1) & ~(size_t)(S - 1)) & ~(unit - 1));
  if (target > unit)
    target -= unit;
  if (target > 0 && n - target < S) // see the note above: never leave a last block shorter than one group
    target -= unit;
  size_t stale_end = n; // `blockBackPoint`: the end of the PREVIOUSLY chosen block while the next one is being chosen
  size_t block_end = n;
  bool first = true;
  while (true)
  {
    BlockSpan b{};
    count_range(counts, in, target, block_end, &distinct, &symbol);
    if (distinct == 1)
    {
      // single-symbol block: swallow the whole run, then give back what does not start on a group boundary
      size_t run = target;
      while (run > 0 && in[run - 1] == symbol)
        run--;
      target = (run + S - 1) & ~(size_t)(S - 1);
      b.single = true;
    }
    else
    {
      size_t extra = 0;
      for (int s = 0; s < 256; s++)
        if (counts[s] == 0)
        {
          counts[s] = 1; // "safe" histogram: every symbol stays codable while the block is extended
          extra++;
        }
      normalize_counts(&b.hist, counts, first ? block_end - target + extra : unit, bits);
      while (target > 0 && stale_end - target < pol.max_block && can_extend(in, target - unit, unit, b.hist, bits, pol, S))
        target -= unit;
      // the histogram that is actually used: counts of [target, stale_end) — for every block but the last that range
      // reaches into the block behind it
      uint32_t d2;
      uint8_t s2;
      count_range(counts, in, target, stale_end, &d2, &s2);
      normalize_counts(&b.hist, counts, stale_end - target, bits);
      b.single = false;
    }
    b.begin = target;
    b.end = block_end;
    back_to_front.push_back(b);
    if (target == 0)
      break;
    // next block towards the front
    stale_end = block_end;
    block_end = target;
    first = false;
    target = (target - 1) & ~(unit - 1);
    if (target > 0 && block_end - target < unit * 2 / 3)
      target -= unit;
  }
  std::reverse(back_to_front.begin(), back_to_front.end());
  return back_to_front;
}

} // namespace

size_t encode(int container, int states, uint32_t bits, const uint8_t *in, size_t n, uint8_t *out, size_t cap, const hsrans_hist *hist,
              hsrans_encode_opts *opts)
{
  if (!valid_codec(container, states, bits) || n == 0 || in == nullptr || out == nullptr)
    return 0;
  if (cap < capacity(container, states, n))
    return 0;
  const uint32_t S = (uint32_t)states;
  // checkpoints of the sidecar plan: every `interval` groups, or at the explicit group indices of opts->index_groups
  const uint64_t *ig = opts && opts->n_index_groups ? opts->index_groups : nullptr;
  size_t ig_left = ig ? opts->n_index_groups : 0; // entries not yet passed (the encoder walks the groups back to front)
  const uint32_t interval = ig ? 0 : (opts ? opts->index_interval : 0);
  const bool want_plan = interval != 0 || ig != nullptr;
  if (want_plan && ((interval % 4) != 0 || opts->plan_out == nullptr))
    return 0;
  for (size_t k = 0; k < ig_left; k++)
    if (ig[k] == 0 || (ig[k] % 4) != 0 || (k > 0 && ig[k] <= ig[k - 1]))
      return 0;
  auto wanted = [&](uint64_t g) { // is group g a checkpoint?  (called with descending g)
    if (interval != 0)
      return g % interval == 0;
    while (ig_left > 0 && ig[ig_left - 1] > g)
      ig_left--;
    return ig_left > 0 && ig[ig_left - 1] == g;
  };
  // block_/mt_: block_size == 0 selects the reference's adaptive block policy (byte-identical streams), anything else
  // fixed blocks of that many symbols
  const bool fixed_blocks = opts && opts->block_size != 0;
  const bool independent = opts && (opts->flags & HSRANS_ENC_INDEPENDENT_BLOCKS) != 0;
  if (independent && (container != HSRANS_MT || !fixed_blocks))
    return 0;
  const size_t block = fixed_blocks ? opts->block_size : 65536;
  if (block % 64 != 0)
    return 0;

  // decoder view of the whole file: T whole groups then `tail` symbols (rANS32x64_16w.cpp:220-250)
  const uint64_t T = n + 1 >= S ? (n - S + 1 + S - 1) / S : 0;
  const uint32_t tail = (uint32_t)(n - T * S);
  const size_t last_group_start = (n - 1) / S * S;

  const size_t header_bytes = container == HSRANS_RAW ? 16 + 512 + 4 * (size_t)S : container == HSRANS_BLOCK ? 16 + 4 * (size_t)S : 16;
  Coder c;
  c.init(S, bits, out, cap);
  c.reserve = header_bytes;
  std::vector<Checkpoint> cps; // recorded back to front

  auto checkpoint = [&](uint64_t group, uint64_t hist_from_end) {
    Checkpoint cp;
    cp.group = group;
    cp.words_from_end = c.written();
    cp.hist_from_end = hist_from_end;
    memcpy(cp.states, c.x, sizeof(cp.states));
    cps.push_back(cp);
  };

  struct BlockMeta
  {
    BlockSpan span;
    uint64_t header_from_end; // bytes from the block header to the end of the stream
    uint64_t hist_from_end;
    uint64_t words_from_end; // decoder cursor at block start
    uint32_t start_states[64];
  };
  std::vector<BlockMeta> metas; // back to front

  if (container == HSRANS_RAW)
  {
    hsrans_hist own;
    if (hist == nullptr)
    {
      make_hist(&own, in, n, bits);
      hist = &own;
    }
    c.hist = *hist;
    for (size_t g = last_group_start / S + 1; g-- > 0;)
    {
      c.put_group(in, g * S, n);
      if (want_plan && g != 0 && g < T && wanted(g))
        checkpoint(g, 0);
    }
  }
  else
  {
    const std::vector<BlockSpan> spans = fixed_blocks ? split_blocks(in, n, block, S, bits) : reference_blocks(container, in, n, S, bits);
    uint64_t next_header_from_end = 0;
    for (size_t b = spans.size(); b-- > 0;)
    {
      const BlockSpan &sp = spans[b];
      BlockMeta m;
      m.span = sp;
      m.hist_from_end = 0;
      const uint64_t size = sp.end - sp.begin;
      if (sp.single)
      {
        // single-symbol block: only the marker word, states untouched (mt_rANS32x64_16w_encode.cpp:289-295)
        const uint64_t marker = size | ((uint64_t)1 << 63) | ((uint64_t)in[sp.begin] << 54);
        m.words_from_end = c.written();
        memcpy(m.start_states, c.x, sizeof(m.start_states));
        c.push_bytes(&marker, 8);
      }
      else
      {
        c.hist = sp.hist;
        if (independent)
          for (uint32_t j = 0; j < 64; j++)
            c.x[j] = kConsumePoint16; // HSRANS_ENC_INDEPENDENT_BLOCKS: every block starts from fresh states
        const size_t g_first = sp.begin / S;
        const size_t g_last = (sp.end - 1) / S; // inclusive; may be the file's partial group
        // positions of this block's counts are only known once its words are written: checkpoints inside the
        // block are patched below
        const size_t cp_mark = cps.size();
        for (size_t g = g_last + 1; g-- > g_first;)
        {
          c.put_group(in, g * S, n);
          if (want_plan && g != g_first && g < T && (interval != 0 ? (g - g_first) % interval == 0 : wanted(g)))
            checkpoint(g, 0);
        }
        m.words_from_end = c.written();
        memcpy(m.start_states, c.x, sizeof(m.start_states));
        c.push_bytes(c.hist.symbolCount, 512);
        m.hist_from_end = c.written();
        for (size_t k = cp_mark; k < cps.size(); k++)
          cps[k].hist_from_end = m.hist_from_end;
        if (container == HSRANS_MT)
        {
          c.push_bytes(c.x, 4 * (size_t)S);
          const uint64_t states_from_end = c.written();
          // skip: uint16 units from the state array to the next block header, minus one (mt_…decode.cpp:59)
          // (the last block's field is never read; the reference measures it from the stream's last word instead of its
          // end, mt_…encode.cpp:149,277, hence one less there)
          const uint64_t skip = (states_from_end - next_header_from_end) / 2 - 1 - (next_header_from_end == 0 ? 1 : 0);
          c.push_bytes(&skip, 8);
        }
        c.push_bytes(&size, 8);
      }
      m.header_from_end = c.written();
      next_header_from_end = m.header_from_end;
      metas.push_back(m);
    }
  }

  if (c.overflow)
    return 0;
  // front header, then the payload moves up behind it
  const size_t payload = c.written();
  const size_t total = header_bytes + payload;
  uint8_t *w = out;
  wr64(w, (uint64_t)n);
  wr64(w + 8, (uint64_t)total);
  w += 16;
  if (container == HSRANS_RAW)
  {
    memcpy(w, c.hist.symbolCount, 512);
    w += 512;
  }
  if (container != HSRANS_MT)
  {
    memcpy(w, c.x, 4 * (size_t)S);
    w += 4 * (size_t)S;
  }
  memmove(w, c.p, payload);

  if (!want_plan)
    return total;

  // ---- sidecar plan: chains in output order ----
  PlanBuilder pb;
  pb.begin(container, states, bits, n, total);
  pb.hdr.interval = interval;
  auto rans_piece = [&](uint64_t g_begin, uint64_t g_end_excl, uint64_t words_from_end, uint64_t hist_off) {
    // groups [g_begin, g_end_excl) of the file; only those below T are whole groups (the caller adds the tail)
    Piece p{};
    p.words_off = total - words_from_end;
    p.out_off = g_begin * S;
    p.hist_off = hist_off;
    const uint64_t whole_end = std::min<uint64_t>(g_end_excl, T);
    p.steps = (uint32_t)(whole_end > g_begin ? whole_end - g_begin : 0);
    return p;
  };
  std::reverse(cps.begin(), cps.end()); // now ascending by group
  if (container == HSRANS_RAW)
  {
    std::vector<uint64_t> ck_group(cps.size()), ck_wfe(cps.size());
    std::vector<uint32_t> ck_states(cps.size() * (size_t)S);
    for (size_t k = 0; k < cps.size(); k++)
    {
      ck_group[k] = cps[k].group;
      ck_wfe[k] = cps[k].words_from_end;
      memcpy(&ck_states[k * S], cps[k].states, 4 * (size_t)S);
    }
    opts->plan_size = raw_plan_from_checkpoints(states, bits, n, total, c.hist.symbolCount, c.x, cps.size(), ck_group.data(), ck_wfe.data(), ck_states.data(), interval,
                                                opts->plan_out, opts->plan_capacity);
    return opts->plan_size ? total : 0;
  }
  {
    size_t k = 0;
    for (size_t b = metas.size(); b-- > 0;) // metas are back to front: walk in output order
    {
      const BlockMeta &m = metas[b];
      if (m.span.single)
      {
        Piece p{};
        p.out_off = m.span.begin;
        p.hist_off = in[m.span.begin];
        p.fill_len = m.span.end - m.span.begin;
        p.flags = kPieceChainStart | kPieceFill;
        pb.add_chain(p, nullptr);
        continue;
      }
      const uint64_t g_first = m.span.begin / S;
      const uint64_t g_last_excl = (m.span.end - 1) / S + 1;
      const bool is_last_block = m.span.end == n;
      uint64_t g = g_first;
      const uint32_t *st = m.start_states;
      uint64_t wfe = m.words_from_end;
      while (true)
      {
        const bool more = k < cps.size() && cps[k].group < g_last_excl && cps[k].group > g_first;
        const uint64_t g_next = more ? cps[k].group : g_last_excl;
        Piece p = rans_piece(g, g_next, wfe, total - m.hist_from_end);
        p.tail = (uint16_t)((!more && is_last_block) ? tail : 0);
        p.flags = kPieceChainStart;
        pb.add_chain(p, st);
        if (!more)
          break;
        g = g_next;
        st = cps[k].states;
        wfe = cps[k].words_from_end;
        k++;
      }
    }
  }
  opts->plan_size = pb.serialize(opts->plan_out, opts->plan_capacity);
  return opts->plan_size ? total : 0;
}

// The sidecar plan of a raw stream from what its encoder recorded (the host encoder above, or the gfx950 one: hsrans_capi.cpp
// hsrans_encode_device_raw): chain 0 from the header's states, then one chain per checkpoint, ascending by group.
size_t raw_plan_from_checkpoints(int states, uint32_t bits, uint64_t n, uint64_t total, const uint16_t counts[256], const uint32_t *start_states, size_t n_ck,
                                 const uint64_t *ck_group, const uint64_t *ck_words_from_end, const uint32_t *ck_states, uint32_t interval, uint8_t *plan_out,
                                 size_t plan_capacity)
{
  const uint32_t S = (uint32_t)states;
  const uint64_t T = n + 1 >= S ? (n - S + 1 + S - 1) / S : 0; // whole groups (rANS32x64_16w.cpp:220-250)
  const uint32_t tail = (uint32_t)(n - T * S);
  const uint64_t G_total = (n - 1) / S + 1; // groups incl. a partial one
  const uint64_t header_bytes = 16 + 512 + 4 * (uint64_t)S;
  if (n == 0 || total < header_bytes)
    return 0;
  PlanBuilder pb;
  pb.begin(HSRANS_RAW, states, bits, n, total);
  pb.hdr.interval = interval;
  pb.set_hist(counts);
  uint64_t g = 0;
  size_t k = 0;
  const uint32_t *st = start_states;
  uint64_t wfe = total - header_bytes;
  while (true)
  {
    const uint64_t g_next = k < n_ck ? ck_group[k] : G_total;
    if (g_next <= g && k < n_ck)
      return 0;
    Piece p{};
    p.words_off = total - wfe;
    p.out_off = g * S;
    p.hist_off = 16;
    const uint64_t whole_end = std::min<uint64_t>(g_next, T); // only groups below T are whole (the last chain adds the tail)
    p.steps = (uint32_t)(whole_end > g ? whole_end - g : 0);
    p.tail = (uint16_t)(g_next == G_total ? tail : 0);
    p.flags = kPieceChainStart;
    pb.add_chain(p, st);
    if (k == n_ck)
      break;
    g = g_next;
    st = ck_states + k * S;
    wfe = ck_words_from_end[k];
    k++;
  }
  return pb.serialize(plan_out, plan_capacity);
}

// ---------------------------------------------------------------------------------------------------------------
// plan builder / serialisation
// ---------------------------------------------------------------------------------------------------------------
void PlanBuilder::begin(int container, int states, uint32_t bits, uint64_t decoded_len, uint64_t stream_len)
{
  hdr = PlanHeader{};
  memcpy(hdr.magic, "HSRPLAN1", 8);
  hdr.container = (uint32_t)container;
  hdr.states = (uint32_t)states;
  hdr.bits = bits;
  hdr.decoded_len = decoded_len;
  hdr.stream_len = stream_len;
  chain_first.clear();
  pieces.clear();
  this->states.clear();
  has_hist = false;
}

void PlanBuilder::add_chain(const Piece &p, const uint32_t *st)
{
  Piece q = p;
  q.flags |= kPieceChainStart;
  q.state_idx = (uint32_t)chain_first.size();
  chain_first.push_back((uint32_t)pieces.size());
  pieces.push_back(q);
  if (st)
    states.insert(states.end(), st, st + hdr.states);
  else
    states.resize(states.size() + hdr.states, 0u);
}

void PlanBuilder::reserve(size_t chains)
{
  chain_first.reserve(chains + 1);
  pieces.reserve(chains);
  states.reserve(chains * (size_t)hdr.states);
}

void PlanBuilder::add_piece(const Piece &p)
{
  Piece q = p;
  q.flags &= (uint16_t)~kPieceChainStart;
  pieces.push_back(q);
}

void PlanBuilder::set_hist(const uint16_t counts[256])
{
  memcpy(hist_counts, counts, sizeof(hist_counts));
  has_hist = true;
}

size_t PlanBuilder::serialized_size() const
{
  return (size_t)plan_size((uint32_t)chain_first.size(), (uint32_t)pieces.size(), hdr.states, kPlanHasHist); // upper bound
}

size_t PlanBuilder::serialize(uint8_t *out, size_t cap)
{
  const uint32_t nc = (uint32_t)chain_first.size(), np = (uint32_t)pieces.size();
  hdr.n_chains = nc;
  hdr.n_pieces = np;
  if (!(hdr.flags & kPlanWalk))
  {
    bool any = false, same = true;
    uint64_t h = 0;
    for (const Piece &p : pieces)
    {
      if (p.flags & kPieceFill)
        continue;
      if (!any)
      {
        h = p.hist_off;
        any = true;
      }
      else if (p.hist_off != h)
        same = false;
    }
    hdr.shared_hist = any && same ? 1 : 0;
    hdr.aux_off = hdr.shared_hist ? h : 0;
    // mergeable: single-piece rANS chains, one histogram, back-to-back in the output, only the last one has a tail.
    // Set for raw streams only: there the word stream is one run, so back-to-back output implies back-to-back words.
    bool merge = hdr.shared_hist && hdr.container == HSRANS_RAW && np == nc && nc > 1;
    for (uint32_t i = 0; merge && i < np; i++)
    {
      const Piece &p = pieces[i];
      if ((p.flags & kPieceFill) || !(p.flags & kPieceChainStart))
        merge = false;
      else if (i + 1 < np && (p.tail != 0 || p.out_off + (uint64_t)p.steps * hdr.states != pieces[i + 1].out_off || p.words_off > pieces[i + 1].words_off))
        merge = false;
    }
    hdr.flags = merge ? (hdr.flags | kPlanMergeable) : (hdr.flags & ~kPlanMergeable);
  }
  const bool with_hist = has_hist && hdr.shared_hist && !(hdr.flags & kPlanWalk);
  hdr.flags = with_hist ? (hdr.flags | kPlanHasHist) : (hdr.flags & ~kPlanHasHist);
  const size_t need = (size_t)plan_size(nc, np, hdr.states, hdr.flags);
  if (need > cap)
    return 0;
  memset(out, 0, need);
  memcpy(out, &hdr, sizeof(hdr));
  uint32_t *cf = (uint32_t *)(out + plan_chain_first_off());
  for (uint32_t i = 0; i < nc; i++)
    cf[i] = chain_first[i];
  cf[nc] = np;
  if (np)
    memcpy(out + plan_pieces_off(nc), pieces.data(), (size_t)np * sizeof(Piece));
  if (!states.empty())
    memcpy(out + plan_states_off(nc, np), states.data(), states.size() * 4);
  if (with_hist)
    memcpy(out + plan_hist_off(nc, np, hdr.states), hist_counts, 512);
  return need;
}

size_t plan_capacity_chains(int container, int states, size_t decoded_size, size_t extra_chains, uint32_t block_size)
{
  size_t chains = 2 + extra_chains;
  if (container != HSRANS_RAW)
  {
    const size_t b = block_size ? block_size : 32768;
    chains += decoded_size / b + 2;
  }
  return (size_t)plan_size((uint32_t)chains, (uint32_t)chains + 2, (uint32_t)states, kPlanHasHist);
}

size_t plan_capacity(int container, int states, size_t decoded_size, uint32_t interval, uint32_t block_size)
{
  const size_t S = (size_t)states;
  const size_t groups = decoded_size / S + 2;
  size_t chains = 2;
  if (container != HSRANS_RAW)
  {
    const size_t b = block_size ? block_size : 32768; // hsrans_plan_build on foreign streams: reference blocks are >= 32 KiB
    chains += decoded_size / b + 2;
  }
  if (interval)
    chains += groups / interval + 1;
  return (size_t)plan_size((uint32_t)chains, (uint32_t)chains + 2, (uint32_t)S, kPlanHasHist);
}

// ---------------------------------------------------------------------------------------------------------------
// planner: derive the chains from the stream alone
// ---------------------------------------------------------------------------------------------------------------
static bool plan_collect(int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, size_t out_cap, PlanBuilder &pb)
{
  if (!valid_codec(container, states, bits) || in == nullptr)
    return false;
  const uint32_t S = (uint32_t)states;
  // the checks every reference decoder opens with (rANS32x64_16w.cpp:171-187, block_…decode.cpp:15-32, mt_…decode.cpp:15-32)
  if (in_len < 16 + 4 * (size_t)S + 512)
    return false;
  const uint64_t out_len = rd64(in);
  if (out_len > out_cap)
    return false;
  const uint64_t stored = rd64(in + 8);
  if (in_len < stored)
    return false;
  if (out_len == 0)
    return false;

  pb.begin(container, states, bits, out_len, in_len);
  uint32_t st[64];

  if (container == HSRANS_RAW)
  {
    // one chain: there is no restart point in the raw format (rANS32x64_16w.cpp:223-250)
    for (uint32_t j = 0; j < S; j++)
      st[j] = rd32(in + 528 + 4 * j);
    Piece p{};
    p.hist_off = 16;
    p.words_off = 528 + 4 * (uint64_t)S;
    p.out_off = 0;
    const uint64_t T = out_len + 1 >= S ? (out_len - S + 1 + S - 1) / S : 0; // trip count of the loop at :223
    if (T > 0xFFFFFFFFull)
      return false;
    p.steps = (uint32_t)T;
    p.tail = (uint16_t)(out_len - T * S);
    pb.add_chain(p, st);
    uint16_t counts[256];
    memcpy(counts, in + 16, 512);
    pb.set_hist(counts);
    return true;
  }

  if (out_len + 1 < S) // `outLen - StateCount + 1` underflows in the reference (block_…decode.cpp:43): undefined there, rejected here
    return false;
  const uint64_t whole = out_len - S + 1;

  if (container == HSRANS_BLOCK)
  {
    // inline headers are only discoverable by decoding (block_…decode.cpp:47-90): the kernel walks them
    for (uint32_t j = 0; j < S; j++)
      st[j] = rd32(in + 16 + 4 * j);
    pb.hdr.flags |= kPlanWalk;
    pb.hdr.aux_off = 16 + 4 * (uint64_t)S;
    Piece p{};
    pb.add_chain(p, st);
    return true;
  }

  // mt_: follow the header chain exactly like mt_rANS32x64_16w_decode.cpp:41-96
  uint64_t pos = 16, i = 0;
  bool last_is_rans = false;
  do
  {
    if (pos + 8 > in_len)
      return false;
    const uint64_t size_val = rd64(in + pos);
    pos += 8;
    if (size_val >> 63)
    {
      const uint64_t len = size_val & (((uint64_t)1 << 54) - 1);
      if (len == 0 || len > out_len - std::min(i, out_len) || i > out_len)
        return false;
      Piece p{};
      p.flags = kPieceFill;
      p.out_off = i;
      p.fill_len = len;
      p.hist_off = (size_val >> 54) & 0xFF;
      pb.add_chain(p, nullptr);
      i += len;
      last_is_rans = false;
    }
    else
    {
      if (pos + 8 + 4 * (uint64_t)S + 512 > in_len)
        return false;
      const uint64_t skip = rd64(in + pos);
      pos += 8;
      if (skip > in_len) // keeps `after` from wrapping
        return false;
      const uint64_t after = pos + 2 * (skip + 1);
      for (uint32_t j = 0; j < S; j++)
        st[j] = rd32(in + pos + 4 * j);
      pos += 4 * (uint64_t)S;
      uint32_t sum = 0;
      for (uint32_t s = 0; s < 256; s++)
        sum += rd16(in + pos + 2 * s);
      if (sum != (1u << bits)) // inplace_complete_hist, hist.cpp:308-324
        return false;
      Piece p{};
      p.hist_off = pos;
      pos += 512;
      p.words_off = pos;
      p.out_off = i;
      uint64_t end = i + size_val;
      if (end > whole || end < i)
        end = whole;
      else if (end & (S - 1))
        return false;
      const uint64_t steps = end > i ? (end - i + S - 1) / S : 0; // decode_section: `for (; i < end; i += S)`
      if (steps > 0xFFFFFFFFull || size_val == 0)
        return false;
      p.steps = (uint32_t)steps;
      pb.add_chain(p, st);
      i += steps * S;
      last_is_rans = true;
      if (i > whole)
        break; // both outcomes of mt_…decode.cpp:86-92 leave the loop
      pos = after;
    }
  } while (i < whole);

  if (i < out_len)
  {
    // final partial group: decoded with the most recent histogram and the states the last block left behind
    // (mt_…decode.cpp:99-130) == a tail on the last chain.  A trailing single-symbol block followed by a partial group
    // cannot be produced by the encoder ("unreachable", :102) and is rejected.
    if (!last_is_rans || out_len - i >= S)
      return false;
    pb.pieces.back().tail = (uint16_t)(out_len - i);
  }
  return true;
}

size_t plan_build(int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, size_t out_cap, uint8_t *plan_out, size_t plan_cap)
{
  PlanBuilder pb;
  if (plan_out == nullptr || !plan_collect(container, states, bits, in, in_len, out_cap, pb))
    return 0;
  return pb.serialize(plan_out, plan_cap);
}

// Same planner into a buffer sized by what the stream turned out to hold: the one-shot decode entries use this, so that a
// stream whose blocks are smaller than the 32 KiB plan_capacity() assumes for foreign streams (block_size option of
// hsrans_encode) still decodes without a caller-made plan.  The chain count is bounded by the stream (>= 8 bytes per header).
bool plan_build_vec(int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, size_t out_cap, std::vector<uint8_t> *plan)
{
  PlanBuilder pb;
  if (plan == nullptr || !plan_collect(container, states, bits, in, in_len, out_cap, pb))
    return false;
  plan->resize(pb.serialized_size());
  const size_t n = pb.serialize(plan->data(), plan->size());
  plan->resize(n);
  return n != 0;
}

// Plans can come from anywhere (files, other processes): nothing a kernel derives an address from is taken on trust.
// Every bound is written in subtraction form (off > len || len - off < need) so that offsets near 2^64 cannot wrap.
bool plan_validate(const uint8_t *plan, size_t size, uint64_t stream_len, uint64_t out_cap)
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return false;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  if (memcmp(h.magic, "HSRPLAN1", 8) != 0 || !valid_codec((int)h.container, (int)h.states, h.bits))
    return false;
  if (h.flags & ~(kPlanWalk | kPlanMergeable | kPlanHasHist))
    return false;
  if (h.n_chains == 0 || h.n_pieces < h.n_chains || plan_size(h.n_chains, h.n_pieces, h.states, h.flags) != size)
    return false;
  if (h.decoded_len > out_cap || h.stream_len > stream_len)
    return false;
  auto fits = [](uint64_t off, uint64_t need, uint64_t len) { return off <= len && len - off >= need; };
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  if (cf[0] != 0 || cf[h.n_chains] != h.n_pieces)
    return false;
  for (uint32_t c = 0; c < h.n_chains; c++)
  {
    if (cf[c] >= cf[c + 1] || cf[c + 1] > h.n_pieces)
      return false;
    if (!(pc[cf[c]].flags & kPieceChainStart) || pc[cf[c]].state_idx >= h.n_chains)
      return false;
  }
  if (h.flags & kPlanWalk)
    return h.container == HSRANS_BLOCK && h.n_chains == 1 && !(h.flags & (kPlanMergeable | kPlanHasHist)) && fits(h.aux_off, 8, stream_len) &&
           h.decoded_len + 1 >= h.states;
  bool any_rans = false;
  for (uint32_t i = 0; i < h.n_pieces; i++)
  {
    const Piece &p = pc[i];
    if (p.flags & kPieceFill)
    {
      if (!fits(p.out_off, p.fill_len, h.decoded_len))
        return false;
      continue;
    }
    any_rans = true;
    const uint64_t syms = (uint64_t)p.steps * h.states + p.tail;
    if (p.tail >= h.states || !fits(p.out_off, syms, h.decoded_len))
      return false;
    if ((p.words_off & 1) || p.words_off > stream_len || !fits(p.hist_off, 512, stream_len))
      return false;
    if ((p.out_off % 4) != 0)
      return false;
    // the kernels address a chain's words with 32-bit offsets from its first word: a piece whose words could span 4 GiB
    // (it consumes at most one word per symbol) is refused instead of decoded wrongly
    const uint64_t span = std::min<uint64_t>(syms * 2, stream_len - p.words_off);
    if (span >= 0xFFFF0000ull)
      return false;
    if (h.shared_hist && p.hist_off != h.aux_off) // one table per workgroup is built from aux_off
      return false;
  }
  if (h.shared_hist && (!any_rans || !fits(h.aux_off, 512, stream_len)))
    return false;
  if ((h.flags & kPlanHasHist) && !h.shared_hist)
    return false;
  if (h.flags & kPlanMergeable)
  {
    // what PlanBuilder::serialize checks before it sets the flag, re-derived: the persistent launches compute output and
    // stream positions of whole runs of chains from the first chain of the run, so the chains must really be back to back
    if (!h.shared_hist || h.container != HSRANS_RAW || h.n_pieces != h.n_chains || h.n_chains < 2)
      return false;
    for (uint32_t i = 0; i < h.n_pieces; i++)
    {
      const Piece &p = pc[i];
      if ((p.flags & kPieceFill) || p.state_idx != i || p.steps == 0)
        return false;
      if (i + 1 < h.n_pieces && (p.tail != 0 || p.out_off + (uint64_t)p.steps * h.states != pc[i + 1].out_off || p.words_off > pc[i + 1].words_off))
        return false;
      if (h.interval != 0 && i + 1 < h.n_pieces && p.steps != h.interval) // interval 0: chains of any length (hsrans_plan_thin)
        return false;
    }
    if (h.interval != 0 && pc[h.n_pieces - 1].steps > h.interval)
      return false;
  }
  return true;
}

size_t plan_slice(const uint8_t *plan, size_t size, uint32_t first, uint32_t count, uint8_t *out, size_t cap)
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return 0;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  if (memcmp(h.magic, "HSRPLAN1", 8) != 0 || (h.flags & kPlanWalk) || count == 0 || first >= h.n_chains || count > h.n_chains - first)
    return 0;
  if (plan_size(h.n_chains, h.n_pieces, h.states, h.flags) != size)
    return 0;
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  const uint32_t *st = (const uint32_t *)(plan + plan_states_off(h.n_chains, h.n_pieces));
  PlanBuilder pb;
  pb.hdr = h;
  if (h.flags & kPlanHasHist)
    pb.set_hist((const uint16_t *)(plan + plan_hist_off(h.n_chains, h.n_pieces, h.states)));
  for (uint32_t c = first; c < first + count; c++)
  {
    for (uint32_t i = cf[c]; i < cf[c + 1]; i++)
    {
      if (i == cf[c])
        pb.add_chain(pc[i], st + (size_t)pc[i].state_idx * h.states);
      else
        pb.add_piece(pc[i]);
    }
  }
  return pb.serialize(out, cap);
}

// Thin a mergeable (raw) plan to the chains that start at the given group indices (see hsrans_plan_thin in the C header):
// a boundary that is not a chain start snaps to the last chain start before it; kept chains swallow the chains behind them.
size_t plan_thin(const uint8_t *plan, size_t size, const uint64_t *groups, size_t n_groups, uint8_t *out, size_t cap)
{
  if (plan == nullptr || size < sizeof(PlanHeader) || out == nullptr || (n_groups != 0 && groups == nullptr))
    return 0;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  if (!plan_validate(plan, size, h.stream_len, h.decoded_len) || !(h.flags & kPlanMergeable))
    return 0;
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  const uint32_t *st = (const uint32_t *)(plan + plan_states_off(h.n_chains, h.n_pieces));
  const uint64_t S = h.states;
  std::vector<uint32_t> keep;
  keep.push_back(0);
  for (size_t k = 0; k < n_groups; k++)
  {
    // last chain whose first group is <= groups[k]
    uint32_t lo = 0, hi = h.n_chains; // invariant: start(lo) <= g < start(hi)
    while (hi - lo > 1)
    {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (pc[mid].out_off / S <= groups[k])
        lo = mid;
      else
        hi = mid;
    }
    if (lo > keep.back())
      keep.push_back(lo);
  }
  PlanBuilder pb;
  pb.hdr = h;
  pb.hdr.interval = 0;
  pb.hdr.flags &= ~(kPlanMergeable | kPlanHasHist); // re-derived by serialize()
  if (h.flags & kPlanHasHist)
    pb.set_hist((const uint16_t *)(plan + plan_hist_off(h.n_chains, h.n_pieces, h.states)));
  for (size_t j = 0; j < keep.size(); j++)
  {
    const uint32_t a = keep[j], b = j + 1 < keep.size() ? keep[j + 1] : h.n_chains;
    Piece p = pc[a];
    const Piece &last = pc[b - 1];
    const uint64_t steps = (last.out_off - p.out_off) / S + last.steps;
    if (steps > 0xFFFFFFFFull)
      return 0;
    p.steps = (uint32_t)steps;
    p.tail = last.tail;
    pb.add_chain(p, st + (size_t)pc[a].state_idx * h.states);
  }
  return pb.serialize(out, cap);
}

bool plan_chain_range(const uint8_t *plan, size_t size, uint32_t first, uint32_t count, uint64_t *begin, uint64_t *end)
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return false;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  if (memcmp(h.magic, "HSRPLAN1", 8) != 0 || (h.flags & kPlanWalk) || count == 0 || first >= h.n_chains || count > h.n_chains - first)
    return false;
  if (plan_size(h.n_chains, h.n_pieces, h.states, h.flags) != size)
    return false;
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  uint64_t lo = ~(uint64_t)0, hi = 0;
  for (uint32_t i = cf[first]; i < cf[first + count]; i++)
  {
    const uint64_t len = (pc[i].flags & kPieceFill) ? pc[i].fill_len : (uint64_t)pc[i].steps * h.states + pc[i].tail;
    lo = std::min(lo, pc[i].out_off);
    hi = std::max(hi, pc[i].out_off + len);
  }
  *begin = lo;
  *end = hi;
  return true;
}

// Stream bytes the chains [first, first + count) can touch: `head` = the histogram every chain shares (shared_hist
// plans: raw streams), `body` = from the first of their own histograms / words to where the next chain's data starts (the
// kernels never request stream bytes beyond the next chain's first word).  Chains are in stream order in every plan this
// library builds.
bool plan_stream_ranges(const uint8_t *plan, size_t size, uint32_t first, uint32_t count, uint64_t out[4])
{
  if (plan == nullptr || size < sizeof(PlanHeader))
    return false;
  PlanHeader h;
  memcpy(&h, plan, sizeof(h));
  if (memcmp(h.magic, "HSRPLAN1", 8) != 0 || (h.flags & kPlanWalk) || count == 0 || first >= h.n_chains || count > h.n_chains - first)
    return false;
  if (plan_size(h.n_chains, h.n_pieces, h.states, h.flags) != size)
    return false;
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h.n_chains));
  uint64_t lo = ~(uint64_t)0, hi = 0;
  bool any = false;
  for (uint32_t i = cf[first]; i < cf[first + count]; i++)
  {
    if (pc[i].flags & kPieceFill)
      continue;
    any = true;
    lo = std::min(lo, pc[i].words_off);
    if (!h.shared_hist)
      lo = std::min(lo, pc[i].hist_off);
    hi = std::max(hi, pc[i].words_off);
  }
  // the end: the first word of the next chain that has words, or the end of the stream
  uint64_t next = h.stream_len;
  for (uint32_t i = cf[first + count]; i < h.n_pieces; i++)
    if (!(pc[i].flags & kPieceFill))
    {
      next = pc[i].words_off;
      break;
    }
  out[0] = 0;
  out[1] = h.shared_hist && any ? h.aux_off + 512 : 0;
  out[2] = any ? lo : 0;
  out[3] = any ? std::max(hi, next) : 0;
  if (out[3] > h.stream_len || out[1] > h.stream_len)
    return false;
  return true;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Sharding one stream over the ranks of a communicator (hsrans_shard_layout; the exchange itself is hsrans_comm.cpp)
// ---------------------------------------------------------------------------------------------------------------------------
// decoded bytes up to and including chain c (chains are in output order in every plan this library builds)
static bool chain_ends(const uint8_t *plan, size_t plan_size, PlanHeader *h, std::vector<uint64_t> *ends)
{
  if (plan == nullptr || plan_size < sizeof(PlanHeader) || (memcpy(h, plan, sizeof(PlanHeader)), memcmp(h->magic, "HSRPLAN1", 8) != 0) || !plan_validate(plan, plan_size, h->stream_len, h->decoded_len))
    return false;
  const uint32_t *cf = (const uint32_t *)(plan + plan_chain_first_off());
  const Piece *pc = (const Piece *)(plan + plan_pieces_off(h->n_chains));
  ends->resize(h->n_chains);
  uint64_t acc = 0;
  uint32_t pi = 0;
  for (uint32_t c = 0; c < h->n_chains; c++)
  {
    for (; pi < cf[c + 1]; pi++)
      acc += (pc[pi].flags & kPieceFill) ? pc[pi].fill_len : (uint64_t)pc[pi].steps * h->states + pc[pi].tail;
    (*ends)[c] = acc;
  }
  return true;
}

// Cuts chains [first, first + count) into n contiguous runs whose decoded bytes follow `shares` (null = equal): run r ends at the
// first chain whose end lies beyond lo + (hi - lo) * (shares[0] + .. + shares[r]) / sum — sums taken in index order, in double.
static void cut(const std::vector<uint64_t> &ends, uint32_t first, uint32_t count, const double *shares, uint32_t n, uint32_t *run_first, uint32_t *run_count)
{
  const uint64_t lo = first > 0 ? ends[first - 1] : 0, hi = count ? ends[first + count - 1] : lo;
  double sum = 0;
  for (uint32_t r = 0; r < n; r++)
    sum += shares ? shares[r] : 1.0;
  double cum = 0;
  uint32_t prev = first;
  for (uint32_t r = 0; r < n; r++)
  {
    uint32_t b = first + count;
    if (r + 1 < n)
    {
      cum += shares ? shares[r] : 1.0;
      const uint64_t target = lo + (uint64_t)((double)(hi - lo) * (cum / sum));
      b = first + (uint32_t)(std::upper_bound(ends.begin() + first, ends.begin() + first + count, target) - (ends.begin() + first));
      b = std::min(std::max(b, prev), first + count);
    }
    run_first[r] = prev;
    run_count[r] = b - prev;
    prev = b;
  }
}


int shard_layout(const uint8_t *plan, size_t plan_size, uint32_t world, uint32_t parts, const double *weights, hsrans_shard *shards, uint64_t *windows)
{
  if (plan == nullptr || shards == nullptr || world == 0 || world > 1024 || parts == 0 || parts > 64)
    return HSRANS_E_ARG;
  if (weights != nullptr)
  {
    double sum = 0;
    for (uint32_t r = 0; r < world; r++)
    {
      if (!(weights[r] >= 0) || !(weights[r] <= 1e300)) // (NaN fails both comparisons)
        return HSRANS_E_ARG;
      sum += weights[r];
    }
    if (!(sum > 0) || !(sum <= 1e300))
      return HSRANS_E_ARG;
  }
  PlanHeader h;
  std::vector<uint64_t> ends;
  if (!chain_ends(plan, plan_size, &h, &ends))
    return HSRANS_E_FORMAT;
  std::vector<uint32_t> rf(world), rc(world), sf(parts), sc(parts);
  cut(ends, 0, h.n_chains, weights, world, rf.data(), rc.data());
  for (uint32_t r = 0; r < world; r++)
  {
    cut(ends, rf[r], rc[r], nullptr, parts, sf.data(), sc.data());
    for (uint32_t k = 0; k < parts; k++)
    {
      hsrans_shard &s = shards[(size_t)r * parts + k];
      s.first_chain = sf[k];
      s.chain_count = sc[k];
      s.out_begin = s.out_end = 0;
      if (sc[k] != 0 && !plan_chain_range(plan, plan_size, sf[k], sc[k], &s.out_begin, &s.out_end))
        return HSRANS_E_FORMAT;
    }
    if (windows != nullptr)
    {
      windows[2 * r] = windows[2 * r + 1] = 0;
      if (rc[r] != 0)
      {
        uint64_t ranges[4];
        if (!plan_stream_ranges(plan, plan_size, rf[r], rc[r], ranges))
          return HSRANS_E_FORMAT;
        windows[2 * r] = ranges[2] & ~(uint64_t)15; // 16-byte aligned start: hsrans_decode_device_window
        windows[2 * r + 1] = ranges[3];
      }
    }
  }
  return HSRANS_OK;
}

} // namespace hsrans
