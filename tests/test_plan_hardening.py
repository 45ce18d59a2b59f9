"""Decode plans are untrusted input (they travel as files): what hsrans_plan_validate must refuse, checked through the entries a
crafted plan would reach without a GPU (hsrans_decode_cpu runs the same validator as hsrans_dplan_create / hsrans_decode_host),
plus the one-chain-per-wave index (hsrans_index_boundaries, explicit checkpoints, hsrans_plan_thin)."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, synth
from oracle_lib import RAW, MT

PIECES_OFF = lambda n_chains: 64 + ((n_chains + 1) * 4 + 15) // 16 * 16


@pytest.fixture(scope="module")
def zipf():
    return synth.enwik8_shaped(400_000, seed=21)


def _ok(container, stream, plan, n):
    return api.decode_cpu(container, 64, 11, stream, n, plan=plan, level=0)[0]


def test_mergeable_flag_is_rederived(zipf):
    """kPlanMergeable came from the blob unchecked; a plan whose chains are not back to back would make the persistent
    launch write (n_chains - 1) * interval * S bytes past the output."""
    s, plan = H.encode(RAW, 64, 11, zipf, index_interval=32)
    n = zipf.size
    assert _ok(RAW, s, plan, n) == n
    hdr, cf, pieces = api.plan_tables(plan)
    assert hdr["flags"] & 2
    po = PIECES_OFF(hdr["n_chains"])

    def mutated(fn):
        p = plan.copy()
        view = p[po:po + 48 * hdr["n_pieces"]].view(api.PIECE_DTYPE)
        fn(view)
        return p

    # first chain moved to the end of the output: every single piece still fits decoded_len, the run as a whole does not
    assert _ok(RAW, s, mutated(lambda v: v["out_off"].__setitem__(0, n - 32 * 64)), n) == 0
    # a gap / overlap between two chains
    assert _ok(RAW, s, mutated(lambda v: v["out_off"].__setitem__(5, v["out_off"][5] + 64)), n) == 0
    # words going backwards
    assert _ok(RAW, s, mutated(lambda v: v["words_off"].__setitem__(7, v["words_off"][6] - 2)), n) == 0
    # a tail in the middle, a chain of another length, a second histogram, a state index out of order
    assert _ok(RAW, s, mutated(lambda v: v["tail"].__setitem__(3, 5)), n) == 0
    assert _ok(RAW, s, mutated(lambda v: v["steps"].__setitem__(3, 31)), n) == 0
    assert _ok(RAW, s, mutated(lambda v: v["hist_off"].__setitem__(3, 18)), n) == 0
    assert _ok(RAW, s, mutated(lambda v: v["state_idx"].__setitem__(3, 4)), n) == 0
    # offsets near 2^64 must not wrap the bounds checks
    for field in ("hist_off", "words_off", "out_off"):
        assert _ok(RAW, s, mutated(lambda v: v[field].__setitem__(2, np.uint64(2**64 - 256))), n) == 0
    p = plan.copy()
    p[56:64] = np.frombuffer(np.uint64(2**64 - 100).tobytes(), np.uint8)  # PlanHeader::aux_off
    assert _ok(RAW, s, p, n) == 0
    p = plan.copy()
    p[20:24] = np.frombuffer(np.uint32(hdr["flags"] | 0x80).tobytes(), np.uint8)  # unknown flag bit
    assert _ok(RAW, s, p, n) == 0


def test_oversized_chain_span_is_refused():
    """The kernels address a chain's words with 32-bit offsets: a piece that could span 4 GiB of stream is refused, not mis-decoded
    (checked on the header arithmetic alone: the stream itself is not needed to be that long for the validator to look)."""
    d = synth.enwik8_shaped(70_000, seed=3)
    s, plan = H.encode(RAW, 64, 11, d, index_interval=32)
    hdr, cf, pieces = api.plan_tables(plan)
    p = plan.copy()
    p[24:32] = np.frombuffer(np.uint64(2**40).tobytes(), np.uint8)  # decoded_len
    p[32:40] = np.frombuffer(np.uint64(2**40).tobytes(), np.uint8)  # stream_len
    view = p[PIECES_OFF(hdr["n_chains"]):][: 48 * hdr["n_pieces"]].view(api.PIECE_DTYPE)
    view["steps"][-1] = 2**31 - 1
    assert H.load_library().hsrans_decode_cpu(0, 1, RAW, 64, 11, api._p(s), 2**40, api._p(np.zeros(16, np.uint8)), 2**40, api._p(p), p.size) == 0


@pytest.mark.parametrize("states", (32, 64))
def test_index_boundaries_cover_the_machine(states):
    S = states
    for bits in (11, 13, 15):
        n = 100_000_000
        g = H.index_boundaries(S, bits, n)
        assert np.all(g % 4 == 0) and np.all(np.diff(g.astype(np.int64)) > 0)
        total = (n - S + 1 + S - 1) // S
        assert 0 < g[0] and g[-1] < total
        lengths = np.diff(np.concatenate([[0], g.astype(np.int64), [total]]))
        # one chain per resident wave (two per wave for 32-state pairs) + at most half as many short tail chains
        assert 4096 * (2 if S == 32 else 1) <= lengths.size <= 1.5 * 8192 * (2 if S == 32 else 1) + 1
        assert lengths.min() >= 4
    assert H.index_boundaries(S, 11, 1000).size == 0  # too short for a second chain


def test_explicit_checkpoints_thin_plans_and_slices(oracle, zipf):
    n = zipf.size
    for S in (32, 64):
        groups = np.array([8, 40, 44, 1000, 4000], np.uint64)
        s, plan = H.encode(RAW, S, 11, zipf, index_groups=groups)
        hdr, cf, pieces = api.plan_tables(plan)
        assert hdr["n_chains"] == 6 and hdr["interval"] == 0 and hdr["flags"] & 2
        assert list(pieces["out_off"] // S) == [0, 8, 40, 44, 1000, 4000]
        r, got = oracle.exec_plan(plan, s, n)
        assert r == n and np.array_equal(got, zipf)
        # the same plan by thinning a fine-grained one; boundaries that are not checkpoints snap down
        s2, fine = H.encode(RAW, S, 11, zipf, index_interval=4)
        assert np.array_equal(s2, s) and np.array_equal(H.plan_thin(fine, groups), plan)
        snapped = H.plan_thin(fine, np.array([9, 43, 46, 1003], np.uint64))
        assert list(api.plan_tables(snapped)[2]["out_off"] // S) == [0, 8, 40, 44, 1000]
        r, got = oracle.exec_plan(snapped, s, n)
        assert r == n and np.array_equal(got, zipf)
        # slices of a plan with chains of any length stay valid plans
        part = H.plan_slice(plan, 2, 3)
        b, e = H.plan_chain_range(plan, 2, 3)
        r, got = oracle.exec_plan(part, s, n)
        assert r == n and np.array_equal(got[b:e], zipf[b:e])
    # bad checkpoint lists are refused
    for bad in ([0, 8], [6], [16, 8], [8, 8]):
        with pytest.raises(H.HsransError):
            H.encode(RAW, 64, 11, zipf, index_groups=np.array(bad, np.uint64))
    # mt_: explicit checkpoints inside the blocks
    s, plan = H.encode(MT, 64, 11, zipf, index_groups=np.arange(1, 60, dtype=np.uint64) * 100)
    r, got = oracle.exec_plan(plan, s, n)
    assert r == n and np.array_equal(got, zipf)
