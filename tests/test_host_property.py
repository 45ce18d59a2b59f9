"""Property tests of the host-side format layer (CPU only): whatever bytes go in, the product's encoders write streams the
oracle's restatement of the reference decoders reads back, and every plan the library builds for them — from the stream
alone, from the encoder's sidecar, sliced — reproduces the same bytes under the oracle's plan interpreter."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import hypersonic_rans_amd as H
from oracle_lib import BLOCK, MT, RAW

SETTINGS = dict(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture, HealthCheck.data_too_large])


def _data(draw):
    kind = draw(st.integers(0, 3))
    n = draw(st.integers(63, 6000))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return rng.choice(np.array([3, 200], np.uint8), n, p=[0.9, 0.1])
    if kind == 2:
        a = rng.zipf(1.3, n)
        return (a % 251).astype(np.uint8)
    runs = rng.integers(0, 256, max(1, n // 500), dtype=np.uint8)
    return np.repeat(runs, 500)[:n].copy() if n // 500 else rng.integers(0, 4, n, dtype=np.uint8)


@st.composite
def case(draw):
    d = _data(draw)
    return dict(d=d, container=draw(st.sampled_from((RAW, BLOCK, MT))), states=draw(st.sampled_from((32, 64))), bits=draw(st.integers(10, 15)),
                interval=draw(st.sampled_from((0, 4, 8, 32))), block=draw(st.sampled_from((0, 32768, 65536))))


@settings(**SETTINGS)
@given(case())
def test_streams_and_plans_round_trip_through_the_oracle(oracle, c):
    d, container, S, bits = c["d"], c["container"], c["states"], c["bits"]
    if np.unique(d).size == 1 and container != RAW:
        return  # the reference rejects its own one-symbol block_/mt_ files (tests/test_oracle_vs_ref.py)
    block = c["block"] if container != RAW else 0
    if c["interval"]:
        s, plan = H.encode(container, S, bits, d, index_interval=c["interval"], block_size=block)
    else:
        s, plan = (H.encode(container, S, bits, d, block_size=block) if block else H.encode(container, S, bits, d)), None
    r, out = oracle.decode(container, S, bits, s, d.size)
    assert r == d.size and np.array_equal(out, d)
    built = H.plan_build(container, S, bits, s)
    for p in (built, plan):
        if p is None or (H.api.plan_tables(p)[0]["flags"] & 1):  # walk plans are interpreted by the kernel only
            continue
        r, out = oracle.exec_plan(p, s, d.size)
        assert r == d.size and np.array_equal(out, d)
        n = H.plan_chain_count(p)
        if n >= 2:  # any split into two slices covers the output, and each slice only needs its stream ranges
            cut = 1 + (int(d[0]) % (n - 1))
            got = np.zeros(d.size, np.uint8)
            for first, count in ((0, cut), (cut, n - cut)):
                masked = np.full_like(s, 0xEE)
                for lo, hi in H.plan_stream_ranges(p, first, count):
                    masked[lo:hi] = s[lo:hi]
                r, part = oracle.exec_plan(H.plan_slice(p, first, count), masked, d.size)
                b, e = H.plan_chain_range(p, first, count)
                assert r == d.size
                got[b:e] = part[b:e]
            assert np.array_equal(got, d)


@settings(**SETTINGS)
@given(case())
def test_independent_block_layout_round_trips(oracle, c):
    d, S, bits = c["d"], c["states"], c["bits"]
    if np.unique(d).size == 1:
        return
    block = 64 * (1 + int(d[0]) % 8) if d.size < 2000 else 1024
    try:
        s = H.encode(H.MT, S, bits, d, block_size=block, independent_blocks=True)
    except H.HsransError:
        return  # capacity() is sized for 32 KiB blocks: tiny blocks of incompressible data may not fit (documented)
    r, out = oracle.decode(MT, S, bits, s, d.size)
    assert r == d.size and np.array_equal(out, d)
