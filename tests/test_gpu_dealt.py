"""k_decode_dealt (round 6): block_/mt_ plans with checkpoints decoded in ONE round with host-dealt shares — the launch a rank of a
sharded decode runs (128 MiB of BASELINE config 4's stream) and 100 MB-class mt_ streams in >= 128 KiB blocks at any index interval.
Every case asserts that the launch under test is the one that ran (hsrans_launch_info: spread == 2) and compares the GPU's bytes
with the oracle's decode of the same stream (reference semantics: /root/reference/src/mt_rANS32x64_16w_decode.cpp:41-133,
block_rANS32x64_16w_decode.cpp:36-126)."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT

pytestmark = pytest.mark.gpu


def shifting(n: int, seed: int, period: int) -> np.ndarray:
    """enwik8-shaped bytes whose alphabet is re-mapped every `period` bytes: every block gets a histogram of its own, so a workgroup
    whose share straddles two blocks really needs its two tables (and no single-symbol blocks: those keep the older launches)"""
    d = synth.enwik8_shaped(n, seed=seed).copy()
    for k, lo in enumerate(range(0, n, period)):
        d[lo:lo + period] ^= np.uint8((k * 37) & 0xFF)
    return d


CASES = (
    # container, bits, size, block, interval
    (H.MT, 11, 24_000_000, 1 << 18, 8),
    (H.MT, 11, 24_000_001, 1 << 18, 16),      # a final partial group
    (H.MT, 10, 40_000_000, 1 << 20, 16),
    (H.MT, 11, 33_554_432, 1 << 17, 16),      # 256 blocks for 512 workgroups: most shares inside one block
    (H.MT, 11, 50_000_000, 3 * 65536, 4),     # many chains per wave
    (H.BLOCK, 11, 30_000_000, 1 << 18, 16),   # block_: states carried over the block boundaries, checkpoints at them
    (H.MT, 11, 20_000_000, 0, 16),            # the reference's adaptive block policy (blocks of >= 64 KiB where the histogram holds)
    (H.MT, 14, 24_000_001, 1 << 18, 8),       # 13 / 14 bits: two RANK tables per workgroup, the second one's address in the gather's offset field
    (H.MT, 13, 30_000_000, 1 << 18, 16),
    (H.BLOCK, 14, 26_000_000, 1 << 19, 8),
)


@pytest.mark.parametrize("container,bits,size,block,interval", CASES)
def test_dealt_launch_against_the_oracle(gpu_ctx, oracle, container, bits, size, block, interval):
    import torch
    d = shifting(size, seed=bits + interval, period=block if block else 150_000)
    stream, plan = H.encode(container, 64, bits, d, index_interval=interval, block_size=block)
    r, want = oracle.decode(MT if container == H.MT else BLOCK, 64, bits, stream, size)
    assert r == size and np.array_equal(want, d)
    dplan = gpu_ctx.make_device_plan(plan)
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16 + 16, np.uint8)])).cuda()
    d_out = torch.full((size + 64,), 0xCC, dtype=torch.uint8, device="cuda")
    for _ in range(2):  # (the dealing is cached in the device plan: the second launch reuses it)
        gpu_ctx.decode_device(dplan, d_in, d_out[:size], stream_length=stream.size)
    assert gpu_ctx.status(dplan) == 0
    info = dplan.launch_info()
    assert info["spread"] == 2, info
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:size], want), int(np.argmax(got[:size] != want))
    assert (got[size:] == 0xCC).all()  # nothing written behind the output


def test_dealt_launch_of_a_gpu_encoded_stream_and_its_device_built_plan(gpu_ctx, oracle):
    """encode -> decode without leaving HBM: the plan is written by the encoder's kernels, the dealing reads the block list back (32
    bytes a group) when the plan is made"""
    import torch
    n = 48_000_000
    d = shifting(n, seed=3, period=1 << 18)
    d_in = torch.from_numpy(d).cuda()
    enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    for interval in (8, 32):
        m, dplan = gpu_ctx.encode_device(H.MT, 64, 11, d_in, enc, block_size=1 << 18, index_interval=interval, want_plan=True)
        r, want = oracle.decode(MT, 64, 11, enc[:m].cpu().numpy(), n)
        assert r == n and np.array_equal(want, d)
        out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, enc, out, stream_length=m)
        assert gpu_ctx.status(dplan) == 0 and dplan.launch_info()["spread"] == 2
        assert torch.equal(out, d_in)


def test_plans_the_dealt_launch_does_not_take_keep_their_launch(gpu_ctx, monkeypatch):
    """single-symbol blocks, 12 bits, too few chains, HSRANS_DEALT=0: the older launches, same bytes"""
    import torch

    def launch_of(data, bits, block, interval):
        stream, plan = H.encode(H.MT, 64, bits, data, index_interval=interval, block_size=block)
        dplan = gpu_ctx.make_device_plan(plan)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16 + 16, np.uint8)])).cuda()
        out = torch.zeros(data.size, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_in, out, stream_length=stream.size)
        assert gpu_ctx.status(dplan) == 0 and np.array_equal(out.cpu().numpy(), data)
        return dplan.launch_info()["spread"]

    d = shifting(24_000_000, seed=9, period=1 << 18)
    assert launch_of(d, 11, 1 << 18, 8) == 2
    assert launch_of(d, 12, 1 << 18, 8) != 2          # two 32 KiB tables do not fit beside the rings twice per CU
    assert launch_of(d, 15, 1 << 18, 8) != 2          # nor two 34 KiB rank tables
    assert launch_of(d, 14, 1 << 18, 8) == 2
    assert launch_of(d[:3_000_000], 11, 1 << 18, 8) != 2  # fewer chains than 1.4 per wave of the device
    runs = synth.nonstationary(24_000_000, seed=2)     # holds single-symbol blocks
    assert launch_of(runs, 11, 1 << 16, 8) != 2
    monkeypatch.setenv("HSRANS_DEALT", "0")
    assert launch_of(d, 11, 1 << 18, 8) != 2


def test_dealt_launch_on_damaged_streams_and_plans(gpu_ctx):
    """Streams and plans are untrusted input: the dealt launch with a valid plan over DAMAGED stream bytes (bit flips in block
    headers, histograms, words; truncation) and with damaged plans that still pass validation must neither hang nor write outside the
    output (guard bytes behind it) — a wrong histogram is reported (status) or decodes to other bytes, like the reference's `return 0` /
    garbage-in-garbage-out (mt_rANS32x64_16w_decode.cpp:68-70 checks the histogram sum only); the undamaged stream decodes again after."""
    import torch
    n = 24_000_000
    d = shifting(n, seed=21, period=1 << 18)
    stream, plan = H.encode(H.MT, 64, 11, d, index_interval=16, block_size=1 << 18)
    dplan = gpu_ctx.make_device_plan(plan)
    pad = (-stream.size) % 16 + 16
    d_out = torch.full((n + 4096,), 0xCC, dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(5)
    reported = 0
    for it in range(40):
        s = stream.copy()
        kind = it % 4
        if kind == 0:    # single bits anywhere
            for p in rng.integers(16, s.size, 8):
                s[p] ^= np.uint8(1 << rng.integers(0, 8))
        elif kind == 1:  # a block's histogram (the first 512 bytes behind a header's 272)
            off = int(rng.integers(0, s.size - 4096))
            s[off:off + 600] = rng.integers(0, 256, 600, dtype=np.uint8)
        elif kind == 2:  # a long run of junk
            off = int(rng.integers(0, s.size - (1 << 20)))
            s[off:off + (1 << 20)] = 0xFF
        else:            # zeros over the first block
            s[16:16 + 70_000] = 0
        d_in = torch.from_numpy(np.concatenate([s, np.zeros(pad, np.uint8)])).cuda()
        gpu_ctx.decode_device(dplan, d_in, d_out[:n], stream_length=s.size)
        torch.cuda.synchronize()
        reported += gpu_ctx.status(dplan) != 0
        assert dplan.launch_info()["spread"] == 2
        assert bool((d_out[n:] == 0xCC).all()), "wrote behind the output"
    assert reported >= 10  # the damaged histograms were noticed
    # damaged plans: whatever hsrans_dplan_create still accepts must be safe to launch
    accepted = 0
    for it in range(60):
        p = plan.copy()
        for q in rng.integers(64, p.size, 4):
            p[q] ^= np.uint8(1 << rng.integers(0, 8))
        try:
            dp = gpu_ctx.make_device_plan(p)
        except H.HsransError:
            continue
        accepted += 1
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros(pad, np.uint8)])).cuda()
        gpu_ctx.decode_device(dp, d_in, d_out[:n], stream_length=stream.size)
        torch.cuda.synchronize()
        gpu_ctx.status(dp)
        assert bool((d_out[n:] == 0xCC).all()), "a damaged plan wrote behind the output"
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros(pad, np.uint8)])).cuda()
    gpu_ctx.decode_device(dplan, d_in, d_out[:n], stream_length=stream.size)
    assert gpu_ctx.status(dplan) == 0 and np.array_equal(d_out[:n].cpu().numpy(), d)


def _awkward(n: int, kind: str) -> np.ndarray:
    rng = np.random.default_rng(len(kind))
    if kind == "all_equal":      # 256 symbols, equal counts: every symbol's first slot 2^bits / 256 behind the last one's
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "ends":           # symbols 0 and 255 only: 254 zero counts between two marks
        return np.where(rng.random(n) < 0.3, 0, 255).astype(np.uint8)
    if kind == "one_heavy":      # one symbol nearly always, every other one rarely: 255 counts of 1 — a mark in every one of 255 neighbouring slots
        d = np.full(n, 7, np.uint8)
        pos = rng.choice(n, n // 400, replace=False)
        d[pos] = rng.integers(0, 256, pos.size, dtype=np.uint8)
        return d
    if kind == "top_only":       # symbols 200 .. 255: no mark before slot 0's own symbol but 200 zero counts
        return rng.integers(200, 256, n, dtype=np.uint8)
    raise ValueError(kind)


@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
@pytest.mark.parametrize("kind", ("all_equal", "ends", "one_heavy", "top_only"))
def test_table_built_from_marks_on_awkward_histograms(gpu_ctx, oracle, bits, kind):
    """The 8-byte table of a block is built on the device from MARKS (kernels_common.h pack64_from_marks: a symbol marks its first slot, a
    slot's symbol is the largest mark at or before it) — 4 slots a thread in the grouped launch at 11 bits, 8 at 12 bits, 256 of its 512
    threads at 10 bits, half a 1,024-thread workgroup per table in the dealt launch; from 13 bits the byte-per-slot table of the rank loop, a
    mark per DWORD and the symbols that begin inside one patched in (rank_from_marks).  Histograms that stress the marks, against the oracle's
    decode (hist.cpp:291-306 make_dec_pack_hist is what both restate): small streams take the grouped launch, the long ones the dealt one."""
    import torch
    for container, n, block, interval, dealt in ((H.MT, 2_000_003, 1 << 16, 16, False), (H.BLOCK, 1_500_000, 1 << 17, 8, False), (H.MT, 24_000_000, 1 << 18, 8, bits <= 11 or bits in (13, 14))):
        d = _awkward(n, kind)
        stream, plan = H.encode(container, 64, bits, d, index_interval=interval, block_size=block)
        r, want = oracle.decode(MT if container == H.MT else BLOCK, 64, bits, stream, n)
        assert r == n and np.array_equal(want, d)
        dplan = gpu_ctx.make_device_plan(plan)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16 + 16, np.uint8)])).cuda()
        d_out = torch.full((n + 64,), 0xCC, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_in, d_out[:n], stream_length=stream.size)
        assert gpu_ctx.status(dplan) == 0
        info = dplan.launch_info()
        assert (info["spread"] == 2) == dealt, (info, n)
        got = d_out.cpu().numpy()
        assert np.array_equal(got[:n], want), (kind, bits, n, int(np.argmax(got[:n] != want)))
        assert (got[n:] == 0xCC).all()
