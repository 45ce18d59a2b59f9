"""Host-side logic of the product, no GPU: C-ABI surface, encoders, planner, plan slicing — checked with the CPU oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, sharded, synth
from oracle_lib import BLOCK, MT, RAW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hsrans_hip.h")).read()
    names = set(re.findall(r"\b(hsrans_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    lib = H.load_library()
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/hsrans_hip.h but not exported"


def test_dropin_cpp_names_are_exported():
    out = subprocess.run(["nm", "-D", "--demangle", H.lib_path()], capture_output=True, text=True, check=True).stdout
    for cont in ("", "block_", "mt_"):
        for S in (32, 64):
            for bits in range(10, 16):
                assert f"hsrans_hip::{cont}rANS32x{S}_16w_decode_hip_{bits}(" in out
                enc = f"hsrans_hip::rANS32x{S}_16w_encode_scalar_{bits}(" if cont == "" else f"hsrans_hip::{cont}rANS32x{S}_16w_encode_{bits}("
                assert enc in out
            assert f"hsrans_hip::{cont}rANS32x{S}_16w_capacity(" in out
            for bits in range(10, 16):  # round 2: runtime dispatch, sidecar index, the reference's thread-pool signature
                for suffix in ("decode_auto", "index_capacity", "encode_with_index", "decode_hip_with_index", "decode_hip_pipelined_with_index"):
                    assert f"hsrans_hip::{cont}rANS32x{S}_16w_{suffix}_{bits}(" in out, (cont, S, suffix, bits)
            assert f"hsrans_hip::mt_rANS32x{S}_16w_decode_mt_11(unsigned char const*, unsigned long, unsigned char*, unsigned long, thread_pool*)" in out


def test_auto_entries_decode_without_a_gpu(oracle):
    """`*_decode_auto_N` (runtime dispatch, the role of block_rANS32x64_decode_wrapper): a single-chain stream goes to this library's
    host SIMD decoder, so the drop-in works — and is not slower than the reference's CPU path — with or without a GPU; without one
    mt_ streams go to the host decoder on all cores.  Called through the C++ names exactly as main.cpp would (mangled symbols)."""
    import ctypes

    import torch

    out = subprocess.run(["nm", "-D", H.lib_path()], capture_output=True, text=True, check=True).stdout
    lib = ctypes.CDLL(H.lib_path())
    d = synth.enwik8_shaped(300_000, seed=5)
    for cont, container in (("", H.RAW), ("block_", H.BLOCK), ("mt_", H.MT)):
        if container == H.MT and torch.cuda.is_available():
            continue  # with a GPU that entry launches kernels: covered by the -m gpu harness test
        sym = next(l.split()[-1] for l in out.splitlines() if f"{cont}rANS32x64_16w_decode_auto_12E" in l and (cont != "" or "block_" not in l and "mt_" not in l))
        fn = getattr(lib, sym)
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        s = H.encode(container, 64, 12, d)
        got = np.full(d.size, 0xCC, np.uint8)
        assert fn(s.ctypes.data, s.size, got.ctypes.data, d.size) == d.size
        assert np.array_equal(got, d), cont
        assert fn(s.ctypes.data, s.size, got.ctypes.data, d.size - 1) == 0  # outCapacity too small: 0, like the reference


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(H.HsransError):
        H.Context(0)


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hypersonic_rans_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "hsrans_oracle" not in text and "libhsrans_ref" not in text, f


@pytest.fixture(scope="module")
def zipf():
    return synth.enwik8_shaped(400_000, seed=31)


@pytest.fixture(scope="module")
def nonstat():
    return synth.nonstationary(1_500_000, seed=77)


@pytest.mark.parametrize("states", (32, 64))
def test_capacity_hist_and_raw_encode_match_the_oracle(oracle, zipf, states):
    for n in (0, 1, 63, 64, 1000, 65536, 100_000_000):
        for c in (RAW, BLOCK, MT):
            assert H.capacity(c, states, n) == oracle.capacity(c, states, n)
    for bits in range(10, 16):
        for n in (1, 31, 63, 64, 65, 127, 128, 1000, 65600, 400_000):
            d = zipf[:n]
            assert list(H.make_hist(d, bits).symbolCount) == list(oracle.make_hist(d, bits).symbolCount)
            assert np.array_equal(H.encode(H.RAW, states, bits, d), oracle.raw_encode(states, bits, d)), (bits, n)


@pytest.mark.parametrize("container", (RAW, BLOCK, MT))
@pytest.mark.parametrize("states", (32, 64))
def test_streams_and_plans(oracle, zipf, nonstat, container, states):
    for bits in (10, 11, 13, 15):
        for src, n in ((zipf, 1), (zipf, 63), (zipf, 64), (zipf, 65), (zipf, 1000), (zipf, 65536), (zipf, 65560), (zipf, 65599), (zipf, 65600),
                       (zipf, 131073), (zipf, 400_000), (nonstat, 1_500_000)):
            d = src[:n]
            for interval in (0, 4, 64):
                if interval:
                    s, plan = H.encode(container, states, bits, d, index_interval=interval)
                else:
                    s, plan = H.encode(container, states, bits, d), None
                if n >= states - 1 or container == RAW:
                    r, got = oracle.decode(container, states, bits, s, n)  # the reference's decoder accepts our stream
                    assert r == n and np.array_equal(got, d), (bits, n, interval)
                if plan is not None:
                    r, got = oracle.exec_plan(plan, s, n)
                    assert r == n and np.array_equal(got, d), ("plan", bits, n, interval)
            if container != BLOCK and (n >= states - 1 or container == RAW):
                plan = H.plan_build(container, states, bits, s)
                r, got = oracle.exec_plan(plan, s, n)
                assert r == n and np.array_equal(got, d), ("plan_build", bits, n)


def test_plan_build_on_reference_streams_reproduces_reference_output(oracle):
    """Golden mt_ streams written by the real reference, incl. the quirk lengths where its own round trip is wrong."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_vectors.npz"))
    n_checked = 0
    for k in [k[:-7] for k in g.files if k.endswith("_stream")]:
        parts = k.split("_")
        if parts[0] == "quirk":
            cont, S, bits, n = parts[1], int(parts[2][1:]), int(parts[3][1:]), int(parts[4][1:])
            want = g[k + "_decoded"]
        else:
            cont, S, bits = parts[0], int(parts[1][1:]), int(parts[2][1:])
            want = g[k + "_in"]
            n = want.size
        if cont == "block":
            continue  # inline headers: walked on the device
        s = g[k + "_stream"]
        plan = H.plan_build({"raw": RAW, "mt": MT}[cont], S, bits, s)
        r, got = oracle.exec_plan(plan, s, n)
        assert r == n and np.array_equal(got, want), k
        n_checked += 1
    assert n_checked > 60


def test_malformed_streams_are_rejected(zipf):
    d = zipf[:10_000]
    for container in (RAW, MT, BLOCK):
        s = H.encode(container, 64, 11, d)
        with pytest.raises(H.HsransError):
            H.plan_build(container, 64, 11, s[:100])  # shorter than a header
        with pytest.raises(H.HsransError):
            H.plan_build(container, 64, 11, s, out_capacity=9_999)  # outCapacity < decodedLength
        t = s.copy()
        t[8:16] = np.frombuffer(np.uint64(s.size + 1).tobytes(), np.uint8)
        with pytest.raises(H.HsransError):
            H.plan_build(container, 64, 11, t)  # inLength < stored length
    s = H.encode(H.MT, 64, 11, d)
    t = s.copy()
    t[16 + 16 + 256] ^= 1  # a count of the first block: sum != 2^11
    with pytest.raises(H.HsransError):
        H.plan_build(H.MT, 64, 11, t)
    with pytest.raises(H.HsransError):
        H.plan_build(H.MT, 64, 12, s)  # wrong bits


def test_plan_slices_cover_the_output(oracle, zipf):
    d = zipf[:300_001]
    for container in (RAW, MT, BLOCK):
        s, plan = H.encode(container, 64, 11, d, index_interval=16)
        for world in (1, 2, 3, 8):
            runs = sharded.shard_chains(plan, world)
            assert sum(c for _, c in runs) == H.plan_chain_count(plan)
            out = np.zeros(d.size, np.uint8)
            pos = 0
            for first, count in runs:
                if count == 0:
                    continue
                b, e = H.plan_chain_range(plan, first, count)
                assert b == pos
                pos = e
                r, part = oracle.exec_plan(H.plan_slice(plan, first, count), s, d.size)
                assert r == d.size
                out[b:e] = part[b:e]
            assert pos == d.size and np.array_equal(out, d)


def test_container_encoders_are_byte_identical_to_the_reference_golden():
    """block_/mt_ encoders with the default (reference) block policy reproduce the reference's streams byte for byte:
    small golden streams + SHA-256 of 1 MiB streams from tests/golden (generated by the real reference)."""
    import hashlib
    import json

    g = np.load(os.path.join(ROOT, "tests", "golden", "small_vectors.npz"))
    n_small = 0
    for k in [k[:-7] for k in g.files if k.endswith("_stream") and not k.startswith("quirk_")]:
        cont, S, bits, _ = k.split("_", 3)
        if cont == "raw":
            continue
        got = H.encode({"block": BLOCK, "mt": MT}[cont], int(S[1:]), int(bits[1:]), g[k + "_in"])
        assert np.array_equal(got, g[k + "_stream"]), k
        n_small += 1
    assert n_small >= 50
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    inputs = {"zipf1M_seed1": synth.enwik8_shaped(1 << 20, seed=1), "uniform1M_seed1": synth.uniform_bytes(1 << 20, seed=1),
              "nonstat1M_seed99": synth.nonstationary(1 << 20, seed=99)}
    n_large = 0
    for e in manifest["large"]:
        if e["container"] == "raw":
            continue
        s = H.encode({"block": BLOCK, "mt": MT}[e["container"]], e["states"], e["bits"], inputs[e["input"]])
        assert s.size == e["stream_len"] and hashlib.sha256(s.tobytes()).hexdigest() == e["stream_sha256"], e
        n_large += 1
    assert n_large >= 40


def test_container_encoders_fix_the_short_last_block(oracle):
    """MinBlockSize < n < MinBlockSize + S: the reference leaves a last block shorter than one group, which its decoders
    mis-decode; our encoder merges it, so the stream round-trips through the (restated) reference decoder."""
    d = synth.enwik8_shaped(140_000, seed=8)
    for container, S, bits, n in ((MT, 64, 11, 65560), (MT, 32, 12, 65550), (BLOCK, 64, 12, 65560), (BLOCK, 64, 13, 131100)):
        s = H.encode(container, S, bits, d[:n])
        r, got = oracle.decode(container, S, bits, s, n)
        assert r == n and np.array_equal(got, d[:n]), (container, S, bits, n)


def test_empty_input_is_a_clean_failure():
    """length 0: the reference's encoders run off their buffers (a segmentation fault in the compiled reference, so it is not
    called here); the product returns 0 / raises, for every container, and no stream claims decodedLength 0."""
    e = np.zeros(0, np.uint8)
    for container in (H.RAW, H.BLOCK, H.MT):
        for states in (32, 64):
            with pytest.raises(H.HsransError):
                H.encode(container, states, 11, e)
            with pytest.raises(H.HsransError):
                H.encode(container, states, 11, e, index_interval=32)
    s = H.encode(H.RAW, 64, 11, np.arange(200, dtype=np.uint8))
    zero_len = s.copy()
    zero_len[:8] = 0  # a stream that says it decodes to nothing
    with pytest.raises(H.HsransError):
        H.plan_build(H.RAW, 64, 11, zero_len)
