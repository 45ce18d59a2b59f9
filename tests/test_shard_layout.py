"""hsrans_shard_layout (csrc/hsrans_comm.cpp): which chains / output bytes / stream bytes each rank of a sharded decode owns.  Pure host
arithmetic in the C library; checked here against an independent restatement in plain Python (sequential double sums, bisect) for
worlds 2, 3, 4, 8, sub-run counts 1..4, equal and weighted shares, all three containers — and for the properties every exchange
relies on: the sub-runs tile the chains and the output without gaps or overlaps, and a rank's stream window covers what its chains read.
The reference's counterpart: blocks handed to pool threads one by one (src/mt_rANS32x64_16w_decode.cpp:217-220)."""
import bisect

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, sharded, synth


def _chain_ends(plan):
    hdr, cf, pieces = api.plan_tables(plan)
    S = hdr["states"]
    size = np.where(pieces["flags"] & 2, pieces["fill_len"], pieces["steps"].astype(np.uint64) * S + pieces["tail"]).astype(np.uint64)
    per_piece_end = np.cumsum(size)
    return [int(x) for x in per_piece_end[cf[1:].astype(np.int64) - 1]], hdr["n_chains"]


def _cut(ends, first, count, shares):
    lo = ends[first - 1] if first > 0 else 0
    hi = ends[first + count - 1] if count else lo
    total = 0.0
    for x in shares:
        total += float(x)
    cum, prev, out = 0.0, first, []
    for r in range(len(shares)):
        b = first + count
        if r + 1 < len(shares):
            cum += float(shares[r])
            target = lo + int(float(hi - lo) * (cum / total))
            b = first + bisect.bisect_right(ends, target, first, first + count) - first
            b = min(max(b, prev), first + count)
        out.append((prev, b - prev))
        prev = b
    return out


CASES = [(H.MT, 65536, 32), (H.MT, 1 << 18, 0), (H.RAW, 0, 16), (H.BLOCK, 65536, 64)]


@pytest.fixture(scope="module")
def plans():
    d = synth.nonstationary(4_000_037, seed=21)
    out = []
    for container, block, interval in CASES:
        if interval:
            s, p = H.encode(container, 64, 11, d, block_size=block, index_interval=interval)
        else:
            s = H.encode(container, 64, 11, d, block_size=block)
            p = H.plan_build(container, 64, 11, s)
        out.append((s, p, d.size))
    return out


@pytest.mark.parametrize("world", (2, 3, 4, 8))
@pytest.mark.parametrize("parts", (1, 3, 4))
def test_c_layout_equals_the_restatement_and_tiles_the_output(plans, world, parts):
    rng = np.random.default_rng(world * 10 + parts)
    for s, plan, n in plans:
        ends, n_chains = _chain_ends(plan)
        for weights in (None, sharded.root_weights(world, world - 1, 0.55), list(rng.uniform(0.2, 3.0, world)), [0.0] + [1.0] * (world - 1)):
            shards, windows = api.shard_layout(plan, world, parts, weights)
            runs = _cut(ends, 0, n_chains, [1.0] * world if weights is None else weights)
            prev_end, prev_chain = 0, 0
            for r in range(world):
                subs = _cut(ends, runs[r][0], runs[r][1], [1.0] * parts)
                assert [(f, c) for f, c, _, _ in shards[r]] == subs, (world, parts, weights, r)
                for (f, c, b, e) in shards[r]:
                    assert f == prev_chain
                    prev_chain += c
                    if c == 0:
                        assert (b, e) == (0, 0)
                        continue
                    assert (b, e) == api.plan_chain_range(plan, f, c)
                    assert b == prev_end and e == ends[f + c - 1]
                    prev_end = e
                f0, cnt = runs[r]
                if cnt:
                    (_, _), (bb, be) = api.plan_stream_ranges(plan, f0, cnt)
                    assert windows[r] == (bb & ~15, be) and windows[r][1] <= s.size
                else:
                    assert windows[r] == (0, 0)
            assert prev_chain == n_chains and prev_end == n
            # the Python wrapper bench.py and the tests read is the same arithmetic
            lay = sharded.ShardLayout(plan, world, parts, weights)
            assert lay.sub_runs == [[(f, c) for f, c, _, _ in subs] for subs in shards] and lay.windows == windows


def test_layout_rejects_bad_arguments(plans):
    _, plan, _ = plans[0]
    for world, parts, weights in ((0, 1, None), (2, 0, None), (2, 1, [1.0, -1.0]), (2, 1, [0.0, 0.0]), (2, 65, None)):
        with pytest.raises(H.HsransError):
            api.shard_layout(plan, world, parts, weights)
    with pytest.raises(H.HsransError):
        api.shard_layout(plan[:100], 2, 1, None)


def test_comm_entries_fail_loudly_without_a_device():
    """No GPU here: a context cannot be made, so no communicator; the unique id only needs an RCCL to bind (present in this image)."""
    L = H.load_library()
    assert L.hsrans_comm_world(None) == 0 and L.hsrans_comm_rank(None) == -1
    assert L.hsrans_decode_sharded(None, None, None, 1, None) != 0
