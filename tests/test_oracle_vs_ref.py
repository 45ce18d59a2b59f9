"""Differential test: oracle restatement vs the REAL reference compiled into oracle/_ref (skipped where it is absent)."""
import numpy as np
import pytest

from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT, RAW


@pytest.fixture(scope="module")
def data():
    return synth.enwik8_shaped(300_000, seed=21), synth.nonstationary(1_200_000, seed=5)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_raw(oracle, ref, data, states, bits):
    zipf, _ = data
    for n in (1, 31, 62, 63, 64, 65, 127, 128, 129, 1000, 65536, 65600, 131073, 300_000):
        d = zipf[:n]
        counts, _ = ref.make_hist(d, bits)
        assert list(oracle.make_hist(d, bits).symbolCount) == list(counts)
        s_ref = ref.encode(RAW, states, bits, d)
        assert np.array_equal(oracle.raw_encode(states, bits, d), s_ref)
        r, got = oracle.decode(RAW, states, bits, s_ref, n)
        assert r == n and np.array_equal(got, d)
        if n >= states - 1:  # below that the reference walks off its buffers (rANS32x64_16w.cpp:220)
            for variant in (0, 1):
                r2, got2 = ref.decode(RAW, states, bits, s_ref, n, variant=variant)
                assert r2 == n and np.array_equal(got2, got)


@pytest.mark.parametrize("container", (BLOCK, MT))
@pytest.mark.parametrize("states", (32, 64))
def test_containers_including_quirk_lengths(oracle, ref, data, container, states):
    zipf, nonstat = data
    for bits in (10, 11, 12, 13, 14, 15):
        for src, n in ((zipf, 64), (zipf, 1000), (zipf, 65536), (zipf, 65537), (zipf, 65560), (zipf, 65599), (zipf, 65600), (zipf, 131073),
                       (zipf, 131100), (zipf, 300_000), (nonstat, 1_200_000), (nonstat, 700_001)):
            if n < states:
                continue
            d = src[:n]
            s = ref.encode(container, states, bits, d)
            r1, o1 = ref.decode(container, states, bits, s, n)
            r2, o2 = oracle.decode(container, states, bits, s, n)
            assert r1 == r2 and np.array_equal(o1, o2), (container, states, bits, n)


@pytest.mark.parametrize("container", (BLOCK, MT))
def test_one_symbol_files_do_not_round_trip_in_the_reference(oracle, ref, container):
    """Reference quirk: a file of ONE repeated byte encodes to a lone single-symbol block (280 / 24 bytes), which the
    reference's own decoders reject because every stream must be at least 16 + 4*S + 512 bytes long
    (block_rANS32x64_16w_decode.cpp:15-32, mt_rANS32x64_16w_decode.cpp:15-32).  The product encoder writes the same bytes,
    the oracle returns the same 0."""
    import hypersonic_rans_amd as H
    for n in (1000, 65536, 67230, 200_000):
        d = np.full(n, 77, np.uint8)
        s = ref.encode(container, 64, 11, d)
        mine = H.encode(container, 64, 11, d)
        assert np.array_equal(s, mine) and s.size == (280 if container == BLOCK else 24)
        assert ref.decode(container, 64, 11, s, n)[0] == 0
        assert oracle.decode(container, 64, 11, s, n)[0] == 0
        with pytest.raises(H.HsransError):
            H.plan_build(container, 64, 11, s)


def test_error_returns(oracle, ref, data):
    zipf, _ = data
    d = zipf[:5000]
    for container in (RAW, BLOCK, MT):
        s = ref.encode(container, 64, 11, d)
        for cap, in_len in ((4999, None), (5000, s.size - 1), (5000, 100)):
            r1, _ = oracle.decode(container, 64, 11, s, cap, in_len=in_len)
            assert r1 == 0
        bad = s.copy()
        off = 16 if container == RAW else (16 + 256 + 8 if container == BLOCK else 16 + 16 + 256)
        bad[off] ^= 1
        assert oracle.decode(container, 64, 11, bad, 5000)[0] == 0
        assert ref.decode(container, 64, 11, bad, 5000)[0] == 0


@pytest.mark.parametrize("container", (BLOCK, MT))
def test_product_container_encoders_match_the_real_reference(ref, data, container):
    """The product's block_/mt_ encoders (reference block policy) against the real reference's encoders, byte for byte."""
    import hypersonic_rans_amd as H

    zipf, nonstat = data
    for states in (32, 64):
        for bits in (10, 11, 12, 13, 14, 15):
            for src, n in ((zipf, 64), (zipf, 1000), (zipf, 65536), (zipf, 65600), (zipf, 262144), (zipf, 300_000), (nonstat, 1_200_000), (nonstat, 700_001)):
                d = src[:n]
                assert np.array_equal(H.encode(container, states, bits, d), ref.encode(container, states, bits, d)), (container, states, bits, n)


@pytest.mark.parametrize("states", (32, 64))
def test_independent_block_streams_are_plain_mt_streams_for_the_real_reference(oracle, ref, data, states):
    """The layout the GPU encoder writes (HSRANS_ENC_INDEPENDENT_BLOCKS: every block restarts from states 2^15) is decoded by
    the real reference — single-threaded and through its thread pool — and by the oracle; each block's histogram is the
    reference's make_hist of that block, byte for byte."""
    import hypersonic_rans_amd as H
    zipf, nonstat = data
    for bits in (10, 11, 13, 15):
        for src, n, block in ((zipf, 64, 64), (zipf, 65536, 65536), (zipf, 65536 + 31, 65536), (zipf, 299_999, 32768), (nonstat, 1_200_000, 65536),
                              (nonstat, 700_001, 1 << 18)):
            if n < states:
                continue
            d = src[:n]
            assert d.size == n
            s = H.encode(H.MT, states, bits, d, block_size=block, independent_blocks=True)
            for variant in (0, 2):
                r, out = ref.decode(MT, states, bits, s, n, variant=variant)
                assert r == n and np.array_equal(out, d), (states, bits, n, block, variant)
            r, out = oracle.decode(MT, states, bits, s, n)
            assert r == n and np.array_equal(out, d)
            # first block: [size][skip][states][counts]
            first = d[: min(n, block)] if n - block >= states or n <= block else d
            if np.unique(first).size > 1:
                counts = s[16 + 16 + 4 * states: 16 + 16 + 4 * states + 512].view("<u2")
                want = np.asarray(ref.make_hist(first, bits)[0], dtype=np.uint16)
                assert np.array_equal(counts, want), (states, bits, n, block)
                st = s[32: 32 + 4 * states].view("<u4")
                assert st.min() >= 1 << 15  # decoder start states are normalised
