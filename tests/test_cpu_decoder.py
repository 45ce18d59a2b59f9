"""The library's own host SIMD decoders (csrc/hsrans_cpu.cpp: scalar / AVX2 / AVX-512 by runtime dispatch) against the CPU
oracle and the golden vectors of the real reference — the same bar as the HIP path, bit-exact, every dispatch level this
host has.  They serve single-chain streams in the `*_decode_auto_N` drop-in entries, build indexes of foreign streams and
are bench.py's in-run CPU comparator; the GPU entries never use them."""
import json
import os

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, synth
from oracle_lib import BLOCK, MT, RAW

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = {"raw": RAW, "block": BLOCK, "mt": MT}
LEVELS = [lv for lv in (0, 1, 2) if lv <= api.cpu_level()]


@pytest.fixture(scope="module")
def zipf():
    return synth.enwik8_shaped(1 << 20, seed=11)


@pytest.fixture(scope="module")
def nonstat():
    return synth.nonstationary(1_500_000)


def test_dispatch_reports_a_level():
    assert api.cpu_level() in (0, 1, 2)
    assert 0 in LEVELS


@pytest.mark.parametrize("level", LEVELS)
def test_golden_vectors_from_the_real_reference(level):
    small = np.load(os.path.join(HERE, "golden", "small_vectors.npz"))
    manifest = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
    keys = [k[:-7] for k in small.files if k.endswith("_stream") and not k.startswith("quirk_")]
    assert len(keys) >= 100
    for k in keys:
        cont, s, b, _tag = k.split("_", 3)
        stream, want = small[k + "_stream"], small[k + "_in"]
        r, got = api.decode_cpu(NAMES[cont], int(s[1:]), int(b[1:]), stream, want.size, level=level)
        assert r == want.size and np.array_equal(got, want), (k, level)
    # MinBlockSize < n < MinBlockSize + S: the reference decodes its own streams wrongly; so do we, byte for byte
    for q in manifest["quirks"]:
        k = q["key"]
        _, cont, s, b, n = k.split("_")
        r, got = api.decode_cpu(NAMES[cont], int(s[1:]), int(b[1:]), small[k + "_stream"], int(n[1:]), level=level)
        assert r == q["returned"] and np.array_equal(got, small[k + "_decoded"]), (k, level)


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("states", (32, 64))
def test_every_container_bits_and_edge_length(oracle, zipf, nonstat, level, states):
    S = states
    for bits in range(10, 16):
        for n in (S - 1, S, S + 1, 2 * S - 1, 2 * S, 1000, 4096, 65536 + S, 100_003):
            for container, src in ((RAW, zipf), (BLOCK, nonstat), (MT, nonstat)):
                d = src[:n]
                s = H.encode(container, S, bits, d, block_size=0 if container == RAW else 32768)
                r0, want = oracle.decode(container, S, bits, s, n)
                r, got = api.decode_cpu(container, S, bits, s, n, level=level)
                assert r == r0 and (r == 0 or np.array_equal(got, want)), (container, S, bits, n, level)


@pytest.mark.parametrize("level", LEVELS)
def test_plans_threads_and_truncated_streams(oracle, zipf, nonstat, level):
    for container, d, kw in ((RAW, zipf, dict(index_interval=32)), (MT, nonstat, dict(index_interval=64, block_size=65536)), (BLOCK, nonstat, dict(index_interval=100))):
        s, plan = H.encode(container, 64, 12, d, **kw)
        for threads in (1, 3):
            r, got = api.decode_cpu(container, 64, 12, s, d.size, plan=plan, level=level, threads=threads)
            assert r == d.size and np.array_equal(got, d), (container, threads)
    # mt_ without a plan on several threads (one block per task, like the reference's thread pool)
    s = H.encode(MT, 32, 11, nonstat)
    r, got = api.decode_cpu(MT, 32, 11, s, nonstat.size, level=level, threads=4)
    assert r == nonstat.size and np.array_equal(got, nonstat)
    # a stream cut short decodes (the reference never checks its cursor either) without reading past the buffer: the same
    # bytes as the oracle gives for the zero-padded stream, at every level
    s = H.encode(RAW, 64, 11, zipf)
    cut = s[: s.size - 3000].copy()
    cut[8:16] = np.frombuffer(np.uint64(cut.size).tobytes(), np.uint8)
    r0, want = api.decode_cpu(RAW, 64, 11, cut, zipf.size, level=0)
    r, got = api.decode_cpu(RAW, 64, 11, cut, zipf.size, level=level)
    assert r == r0 == zipf.size and np.array_equal(got, want)
    # malformed: histogram sum off by one -> 0, as hist.cpp:308-324
    bad = s.copy()
    bad[16] ^= 1
    assert api.decode_cpu(RAW, 64, 11, bad, zipf.size, level=level)[0] == 0


def test_host_index_builder_matches_the_encoders_plan(oracle, zipf, nonstat):
    """Checkpoints recovered from the stream alone by one host decode pass = the plan the encoder writes for the same positions."""
    for S, bits in ((64, 11), (32, 13)):
        groups = np.arange(1, zipf.size // S // 40) * 40
        s, plan = H.encode(RAW, S, bits, zipf, index_groups=groups)
        built = api.index_build_host(RAW, S, bits, s, groups)
        assert np.array_equal(built, plan)
        r, got = oracle.exec_plan(built, s, zipf.size)
        assert r == zipf.size and np.array_equal(got, zipf)
    # the one-chain-per-wave positions of hsrans_index_boundaries, and an mt_ stream (checkpoints inside the blocks, in parallel)
    g = H.index_boundaries(64, 11, zipf.size)
    s, plan = H.encode(RAW, 64, 11, zipf, index_groups=g)
    assert np.array_equal(api.index_build_host(RAW, 64, 11, s, g, threads=2), plan)
    s = H.encode(MT, 64, 11, nonstat)
    g = np.arange(1, nonstat.size // 64 // 128) * 128
    built = api.index_build_host(MT, 64, 11, s, g, threads=3)
    r, got = oracle.exec_plan(built, s, nonstat.size)
    assert r == nonstat.size and np.array_equal(got, nonstat)
    assert H.plan_chain_count(built) > H.plan_chain_count(H.plan_build(MT, 64, 11, s))


@pytest.mark.parametrize("states", (32, 64))
def test_hostile_start_states_decode_the_same_at_every_level(states):
    """start states come from the (untrusted) plan blob.  A state >= 2^31 must renormalise the same way — never — at
    every dispatch level (the AVX2 level used a signed compare): same bytes, whatever they are, from scalar, AVX2 and AVX-512."""
    data = synth.enwik8_shaped(200_000, seed=33)
    stream, plan = H.encode(H.RAW, states, 11, data, index_interval=64)
    hdr, cf, pieces = api.plan_tables(plan)
    so = 64 + ((hdr["n_chains"] + 1) * 4 + 15) // 16 * 16 + 48 * hdr["n_pieces"]
    bad = plan.copy()
    st = bad[so:so + 4 * states * hdr["n_chains"]].view("<u4")
    st[3 * states + 5] = 0x80001234          # chain 3, state 5: above 2^31
    st[7 * states + states - 1] = 0xFFFFFFFF
    outs = []
    for level in range(api.cpu_level() + 1):
        r, out = api.decode_cpu(H.RAW, states, 11, stream, data.size, plan=bad, level=level)
        assert r == data.size
        outs.append(out.copy())
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("container", [H.BLOCK, H.MT])
@pytest.mark.parametrize("block_size", [4096, 16384])
def test_blocks_smaller_than_the_reference_minimum_decode_without_a_caller_plan(container, block_size):
    """hsrans_plan_capacity assumes the reference's >= 32 KiB blocks for foreign streams; the one-shot decode entries size their
    own plan by the chains the stream holds instead (a GPU fuzz run found the 16 KiB case returning 0)."""
    from hypersonic_rans_amd import api
    data = synth.nonstationary(200_000, seed=17)
    stream = H.encode(container, 64, 11, data, block_size=block_size)
    r, out = api.decode_cpu(container, 64, 11, stream)
    assert r == data.size and np.array_equal(out[:r], data)
    plan = H.plan_build(container, 64, 11, stream)
    r, out = api.decode_cpu(container, 64, 11, stream, plan=plan)
    assert r == data.size and np.array_equal(out[:r], data)
