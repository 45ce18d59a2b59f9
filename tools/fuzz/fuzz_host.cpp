// Sanitizer fuzz of the host-side code that handles UNTRUSTED input (streams, decode plans): the planner (plan_build), the plan
// validator (plan_validate / plan_slice / plan_thin / plan_stream_ranges) and the host SIMD decoder (cpu::decode, cpu::exec_plan,
// cpu::index_build).  Built with -fsanitize=address,undefined (tools/fuzz/Makefile) — no GPU, no HIP runtime: the same plan
// validator guards every GPU entry (hsrans_dplan_create, hsrans_decode_host), so what gets past it here is what a kernel would
// take addresses from.  Deterministic: xorshift from a seed; every iteration mutates a valid stream or plan (byte flips, 16/32/64
// bit fields set to edge values, truncation) and runs it through all entries; any sanitizer report aborts.
//   make -C tools/fuzz && tools/fuzz/fuzz_host [seconds] [seed]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "../../hypersonic_rans_amd/csrc/hsrans_batch.h"
#include "../../hypersonic_rans_amd/csrc/hsrans_cpu.h"
#include "../../hypersonic_rans_amd/csrc/hsrans_host.h"

using namespace hsrans;

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
  g_rng ^= g_rng >> 12, g_rng ^= g_rng << 25, g_rng ^= g_rng >> 27;
  return g_rng * 0x2545F4914F6CDD1Dull;
}

static void mutate(std::vector<uint8_t> &v)
{
  if (v.empty())
    return;
  const int kind = (int)(rnd() % 6);
  const size_t at = (size_t)(rnd() % v.size());
  static const uint64_t edges[] = {0, 1, 0x7FFF, 0x8000, 0xFFFF, 0x7FFFFFFF, 0x80000000ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0x7FFFFFFFFFFFFFFFull, ~0ull, ~0ull - 255};
  switch (kind)
  {
  case 0: v[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
  case 1: v[at] = (uint8_t)rnd(); break;
  case 2:
  {
    const uint64_t e = edges[rnd() % (sizeof(edges) / sizeof(edges[0]))];
    const size_t w = (size_t)1 << (1 + rnd() % 3); // 2, 4 or 8 bytes
    const size_t a = at / w * w;
    if (a + w <= v.size())
      memcpy(v.data() + a, &e, w);
    break;
  }
  case 3: v.resize(at); break;                                       // truncate
  case 4: for (int k = 0; k < 8; k++) v[(size_t)(rnd() % v.size())] = (uint8_t)rnd(); break;
  default:
  {
    uint64_t x;
    const size_t a = at / 8 * 8;
    if (a + 8 <= v.size())
    {
      memcpy(&x, v.data() + a, 8);
      x += (rnd() & 1) ? 64 : (uint64_t)-64; // nudge an offset / length
      memcpy(v.data() + a, &x, 8);
    }
  }
  }
}

int main(int argc, char **argv)
{
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  if (argc > 2)
    g_rng ^= strtoull(argv[2], nullptr, 10) * 0x100000001B3ull;
  const size_t n = 150000;
  std::vector<uint8_t> data(n);
  for (size_t i = 0; i < n; i++)
  {
    const uint64_t r = rnd();
    data[i] = (uint8_t)((r & 0xFF) < 200 ? (r >> 8) % 7 : (r >> 8) % 256); // skewed, all symbols present
  }
  for (size_t i = 60000; i < 100000; i++) // a run that becomes a single-symbol block
    data[i] = 42;
  struct Case
  {
    int container, states;
    uint32_t bits;
    std::vector<uint8_t> stream, plan;
  };
  std::vector<Case> cases;
  for (int container = HSRANS_RAW; container <= HSRANS_MT; container++)
    for (int states : {32, 64})
      for (uint32_t bits : {10u, 13u, 15u})
      {
        Case c{container, states, bits, {}, {}};
        c.stream.resize(capacity(container, states, n));
        c.plan.resize(plan_capacity(container, states, n, 16, 16384));
        hsrans_encode_opts o{};
        o.block_size = container == HSRANS_RAW ? 0 : 16384;
        o.index_interval = 16;
        o.plan_out = c.plan.data();
        o.plan_capacity = c.plan.size();
        const size_t m = encode(container, states, bits, data.data(), n, c.stream.data(), c.stream.size(), nullptr, &o);
        if (m == 0)
        {
          fprintf(stderr, "encode failed (%d %d %u)\n", container, states, bits);
          return 2;
        }
        c.stream.resize(m);
        c.plan.resize(o.plan_size);
        cases.push_back(c);
      }
  std::vector<uint8_t> out(n + 64), scratch(1 << 20);
  // every unmodified case decodes to the data, with and without its plan, at every level
  for (const Case &c : cases)
    for (int level = 0; level <= cpu::best_level(); level++)
    {
      if (cpu::decode(level, 1, c.container, c.states, c.bits, c.stream.data(), c.stream.size(), out.data(), n) != n || memcmp(out.data(), data.data(), n) != 0)
        return fprintf(stderr, "baseline decode failed\n"), 2;
      if (cpu::exec_plan(level, 2, c.plan.data(), c.plan.size(), c.stream.data(), c.stream.size(), out.data(), n) != n || memcmp(out.data(), data.data(), n) != 0)
        return fprintf(stderr, "baseline plan decode failed\n"), 2;
    }
  const clock_t t0 = clock();
  uint64_t iters = 0, accepted = 0;
  while ((double)(clock() - t0) / CLOCKS_PER_SEC < seconds)
  {
    const Case &c = cases[(size_t)(rnd() % cases.size())];
    std::vector<uint8_t> stream = c.stream, plan = c.plan;
    const int what = (int)(rnd() % 3);
    const int rounds = 1 + (int)(rnd() % 3);
    for (int r = 0; r < rounds; r++)
    {
      if (what != 1)
        mutate(plan);
      if (what != 0)
        mutate(stream);
    }
    const int level = (int)(rnd() % (uint64_t)(cpu::best_level() + 1));
    // the stream alone: planner + decoder
    std::vector<uint8_t> built(plan_capacity(c.container, c.states, n, 0, 0));
    const size_t bl = plan_build(c.container, c.states, c.bits, stream.data(), stream.size(), n, built.data(), built.size());
    (void)bl;
    (void)cpu::decode(level, 1 + (uint32_t)(rnd() % 3), c.container, c.states, c.bits, stream.data(), stream.size(), out.data(), n);
    // the (mutated) plan: validator first — whatever it lets through must be safe to run, slice, thin and query
    if (plan_validate(plan.data(), plan.size(), stream.size(), n))
    {
      accepted++;
      (void)cpu::exec_plan(level, 1 + (uint32_t)(rnd() % 3), plan.data(), plan.size(), stream.data(), stream.size(), out.data(), n);
      PlanHeader h;
      memcpy(&h, plan.data(), sizeof(h));
      const uint32_t first = (uint32_t)(rnd() % h.n_chains), count = 1 + (uint32_t)(rnd() % (h.n_chains - first));
      std::vector<uint8_t> sl(plan.size());
      const size_t sn = plan_slice(plan.data(), plan.size(), first, count, sl.data(), sl.size());
      if (sn != 0 && plan_validate(sl.data(), sn, stream.size(), n))
        (void)cpu::exec_plan(level, 1, sl.data(), sn, stream.data(), stream.size(), out.data(), n);
      uint64_t ranges[4], b, e;
      (void)plan_stream_ranges(plan.data(), plan.size(), first, count, ranges);
      (void)plan_chain_range(plan.data(), plan.size(), first, count, &b, &e);
      uint64_t groups[3] = {64, 256, 1024};
      (void)plan_thin(plan.data(), plan.size(), groups, 3, sl.data(), sl.size());
      // the sharding arithmetic of the multi-GPU entries (hsrans_shard_layout): any world / sub-run count / weights, hostile ones too
      {
        const uint32_t world = 1 + (uint32_t)(rnd() % 9), parts = 1 + (uint32_t)(rnd() % 5);
        std::vector<hsrans_shard> shards((size_t)world * parts);
        std::vector<uint64_t> windows(2 * (size_t)world);
        std::vector<double> w(world);
        static const double odd[] = {0.0, 1.0, 1e-300, 1e300, -1.0, 0.5, 3.0};
        for (double &x : w)
          x = (rnd() & 3) ? (double)(rnd() % 1000) / 100.0 : odd[rnd() % 7];
        if (rnd() % 16 == 0)
          w[0] = 0.0 / (double)(rnd() % 2); // NaN (0/0) or 0/1
        const int rc = shard_layout(plan.data(), plan.size(), world, parts, (rnd() & 1) ? w.data() : nullptr, shards.data(), (rnd() & 1) ? windows.data() : nullptr);
        if (rc == HSRANS_OK)
        {
          // (a mutated plan that passes the validator may have chains whose outputs overlap: memory-safe, wrong bytes — so the ranges
          // need not tile the output, but every one must lie inside it and the sub-runs must tile the CHAINS)
          uint32_t next = 0;
          for (const hsrans_shard &sh : shards)
          {
            if (sh.first_chain != next || sh.first_chain + sh.chain_count > h.n_chains || sh.out_end < sh.out_begin || sh.out_end > h.decoded_len)
              return fprintf(stderr, "shard_layout: a sub-run is out of range\n"), 3;
            next += sh.chain_count;
          }
          if (next != h.n_chains)
            return fprintf(stderr, "shard_layout: %u of %u chains\n", next, h.n_chains), 3;
        }
      }
    }
    else
      (void)cpu::exec_plan(level, 1, plan.data(), plan.size(), stream.data(), stream.size(), out.data(), n); // must refuse, not crash
    if (c.container != HSRANS_BLOCK)
    {
      uint64_t groups[4] = {8, 64, 512, 2000};
      (void)cpu::index_build(level, 1, c.container, c.states, c.bits, stream.data(), stream.size(), groups, 4, scratch.data(), scratch.size());
    }
    // the batch launch's dealing (hsrans_batch_deal / hsrans_index_boundaries_batch): random members, random chain lengths
    if (rnd() % 8 == 0)
    {
      const uint32_t M = 1 + (uint32_t)(rnd() % 6), grid = 2 + (uint32_t)(rnd() % 40), waves = (rnd() & 1) ? 16 : 1 + (uint32_t)(rnd() % 16);
      std::vector<std::vector<uint64_t>> starts(M);
      std::vector<BatchDealMember> in(M);
      std::vector<uint64_t> totals(M);
      for (uint32_t m = 0; m < M; m++)
      {
        const uint32_t nc = 1 + (uint32_t)(rnd() % 3000);
        starts[m].resize(nc + 1);
        uint64_t g = 0;
        for (uint32_t c2 = 0; c2 < nc; c2++)
        {
          starts[m][c2] = g;
          g += (rnd() % 5 == 0) ? 0 : 1 + rnd() % ((rnd() & 7) ? 300 : 100000);
        }
        starts[m][nc] = g;
        in[m] = BatchDealMember{starts[m].data(), nc, g};
        totals[m] = g;
      }
      uint32_t w8[8];
      for (uint32_t &x : w8)
        x = (rnd() % 10 == 0) ? 1 : 200 + (uint32_t)(rnd() % 1500);
      const BatchDeal deal = batch_deal(in, grid, waves, w8);
      std::vector<uint32_t> covered(M, 0);
      for (size_t wv = 0; wv < deal.slots.size(); wv++)
      {
        const BatchSlot &sl2 = deal.slots[wv];
        if (sl2.member >= M || sl2.begin > sl2.end || sl2.end > in[sl2.member].n_chains)
          return fprintf(stderr, "batch_deal: slot out of range\n"), 3;
        if (sl2.member != deal.slots[wv / waves * waves].member)
          return fprintf(stderr, "batch_deal: a workgroup mixes members\n"), 3;
        covered[sl2.member] += sl2.end - sl2.begin;
      }
      for (uint32_t m = 0; m < M; m++)
        if (deal.wg_count[m] != 0 && covered[m] != in[m].n_chains)
          return fprintf(stderr, "batch_deal: member %u: %u of %u chains dealt\n", m, covered[m], in[m].n_chains), 3;
      std::vector<uint64_t> bounds(1 << 16);
      const size_t nb = batch_boundaries(totals.data(), M, (uint32_t)(rnd() % M), grid, waves, w8, bounds.data(), bounds.size());
      for (size_t k = 1; k + 1 < nb; k++)
        if (bounds[k] <= bounds[k - 1])
          return fprintf(stderr, "batch_boundaries: not ascending\n"), 3;
    }
    iters++;
  }
  printf("fuzz_host: %llu iterations, %llu mutated plans passed the validator, no sanitizer report\n", (unsigned long long)iters, (unsigned long long)accepted);
  return 0;
}
