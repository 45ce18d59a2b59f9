"""Pins the CPU oracle (oracle/hsrans_oracle.c) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  No GPU, no reference needed at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT, RAW

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = {"raw": RAW, "block": BLOCK, "mt": MT}


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(HERE, "golden", "small_vectors.npz"))


@pytest.fixture(scope="module")
def manifest():
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        return json.load(f)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _inputs():
    return {"zipf1M_seed1": synth.enwik8_shaped(1 << 20, seed=1), "uniform1M_seed1": synth.uniform_bytes(1 << 20, seed=1),
            "nonstat1M_seed99": synth.nonstationary(1 << 20, seed=99), "two3000": synth.two_symbol(3000, seed=5)}


def test_small_vectors_decode(oracle, small):
    keys = [k[:-7] for k in small.files if k.endswith("_stream") and not k.startswith("quirk_")]
    assert len(keys) >= 100
    for k in keys:
        cont, s, b, _tag = k.split("_", 3)
        stream, want = small[k + "_stream"], small[k + "_in"]
        r, got = oracle.decode(NAMES[cont], int(s[1:]), int(b[1:]), stream, want.size)
        assert r == want.size and np.array_equal(got, want), k


def test_small_vectors_raw_encode_is_byte_identical(oracle, small):
    for k in [k[:-7] for k in small.files if k.endswith("_stream") and k.startswith("raw_")]:
        _, s, b, _tag = k.split("_", 3)
        got = oracle.raw_encode(int(s[1:]), int(b[1:]), small[k + "_in"])
        assert np.array_equal(got, small[k + "_stream"]), k


def test_quirk_lengths_reproduce_the_reference_output(oracle, small, manifest):
    """MinBlockSize < n < MinBlockSize + S: the reference's own round trip is wrong; the oracle must return the same bytes."""
    assert any(not q["round_trip_ok"] for q in manifest["quirks"])
    for q in manifest["quirks"]:
        k = q["key"]
        _, cont, s, b, n = k.split("_")
        stream, want = small[k + "_stream"], small[k + "_decoded"]
        r, got = oracle.decode(NAMES[cont], int(s[1:]), int(b[1:]), stream, int(n[1:]))
        assert r == q["returned"] and np.array_equal(got, want), k


def test_large_manifest(oracle, manifest):
    inputs = _inputs()
    checked = 0
    for e in manifest["large"]:
        if e["container"] != "raw":
            continue  # block_/mt_ streams need the reference's encoder heuristics; covered by tests/test_oracle_vs_ref.py
        data = inputs[e["input"]]
        stream = oracle.raw_encode(e["states"], e["bits"], data)
        assert stream.size == e["stream_len"] and _sha(stream) == e["stream_sha256"], e
        r, got = oracle.decode(RAW, e["states"], e["bits"], stream, data.size)
        assert r == data.size and _sha(got) == e["decoded_sha256"], e
        checked += 1
    assert checked == 36


def test_hist_and_capacity(oracle, manifest):
    inputs = _inputs()
    for e in manifest["hist"]:
        h = oracle.make_hist(inputs[e["input"]], e["bits"])
        assert list(h.symbolCount) == e["counts"], (e["input"], e["bits"])
    for e in manifest["capacity"]:
        assert oracle.capacity(NAMES[e["container"]], e["states"], e["n"]) == e["capacity"], e


def test_idx2idx_is_a_permutation(oracle):
    for S in (32, 64):
        assert sorted(oracle.idx2idx(j) for j in range(S)) == list(range(S))


def test_table_rejects_bad_sums(oracle):
    counts = np.zeros(256, np.uint16)
    counts[0] = 1000
    assert oracle.make_dec_table(11, counts)[0] == 0
    counts[1] = 1048
    ok, cumul, inv = oracle.make_dec_table(11, counts)
    assert ok == 1 and inv[999] == 0 and inv[1000] == 1 and cumul[1] == 1000
    # uint16 wrap-around quirk of the scalar builder (hist.cpp:332) vs the uint32 sum (hist.cpp:310)
    counts[:] = 0
    counts[:33] = 2048
    counts[32] = 2048  # 33 * 2048 = 67584 = 65536 + 2048
    assert oracle.make_dec_table(11, counts, wide_sum=0)[0] == 1
    assert oracle.make_dec_table(11, counts, wide_sum=1)[0] == 0
