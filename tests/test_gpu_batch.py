"""K independent streams decoded by ONE launch (include/hsrans_hip.h: hsrans_dplan_batch_*, hsrans_decode_device_batch) — the
counterpart of the reference's pool of independent work items (src/mt_rANS32x64_16w_decode.cpp:182-224: a task per block;
src/main.cpp:841-898: file after file).  Every member's output is compared with the scalar CPU oracle's decode of the same
stream; the dealing of the wave slots is checked on the host without a GPU."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT, RAW

GRID, WAVES = 512, 16


def _starts_wave(n, states=64, bits=11):
    g = H.index_boundaries(states, bits, n, None)
    return np.concatenate([[0], g, [n // states]]).astype(np.uint64)


def _starts_uniform(n, interval, states=64):
    total = n // states
    s = np.arange(0, total, interval, dtype=np.uint64)
    return np.concatenate([s, [total]]).astype(np.uint64)


@pytest.mark.parametrize("members", (
    [("wave", 100_000_000)],
    [("wave", 100_000_000)] * 2,
    [("wave", 100_000_000)] * 4,
    [("wave", 100_000_000)] * 5,
    [(32, 100_000_000)] * 3,
    [("wave", 100_000_000), (32, 100_000_000), (16, 3_000_000), ("wave", 1_000_000), (64, 64 * 1024)],
    [(8, 1 << 20)] * 32,
))
def test_dealing_covers_every_chain_once_and_never_mixes_members_in_a_workgroup(members):
    """hsrans_batch_deal (no device): every member's chains are dealt exactly once, in contiguous runs; a workgroup's 16 slots
    name one member (a workgroup holds one decode table); exactly one workgroup per member checks the histogram."""
    starts = [_starts_wave(n) if kind == "wave" else _starts_uniform(n, kind) for kind, n in members]
    imbalance, slots = H.batch_deal(starts, GRID, WAVES)
    assert slots.shape == (GRID * WAVES, 4) and imbalance >= 0.99
    per_wg_member = slots[:, 0].reshape(GRID, WAVES)
    assert (per_wg_member == per_wg_member[:, :1]).all(), "a workgroup mixes members"
    for m, st in enumerate(starts):
        nc = st.size - 1
        mine = slots[slots[:, 0] == m]
        runs = mine[mine[:, 1] < mine[:, 2]]
        order = np.argsort(runs[:, 1], kind="stable")
        runs = runs[order]
        assert runs.shape[0] >= 1 and runs[0, 1] == 0 and runs[-1, 2] == nc
        assert (runs[1:, 1] == runs[:-1, 2]).all(), "runs overlap or leave a gap"
        idle = mine[mine[:, 1] >= mine[:, 2]]
        assert (idle[:, 1] == nc).all() and (idle[:, 2] == nc).all()
        wgs_checking = np.unique(np.nonzero((slots[:, 0] == m) & (slots[:, 3] & 1 != 0))[0] // WAVES)
        assert wgs_checking.size == 1, "exactly one workgroup per member compares the histograms"
    # members of one size and the one-chain-per-wave index at 1, 1/2, 1/4 of the device: an exact fit
    if all(kind == "wave" for kind, _ in members) and len(members) in (1, 2, 4) and len({n for _, n in members}) == 1:
        assert imbalance < 1.05, imbalance


def test_dealing_rejects_bad_arguments():
    st = _starts_uniform(1 << 20, 32)
    with pytest.raises(H.HsransError):
        H.batch_deal([st], 1, 16)
    with pytest.raises(H.HsransError):
        H.batch_deal([st[::-1].copy()], GRID, WAVES)


# ---- on the GPU ---------------------------------------------------------------------------------------------------------------
def _member(ctx, oracle, container, states, bits, n, seed, index):
    """(stream, plan, expected bytes per the oracle, device tensors, device plan)"""
    import torch

    data = synth.enwik8_shaped(n, seed=seed)
    if index == "wave":
        stream, plan = H.encode(container, states, bits, data, index_groups=H.index_boundaries(states, bits, n, ctx))
    elif index == "none":
        stream = H.encode(container, states, bits, data)
        plan = H.plan_build(container, states, bits, stream)
    elif container == RAW:
        stream, plan = H.encode(container, states, bits, data, index_interval=index)
    else:
        stream, plan = H.encode(container, states, bits, data, block_size=1 << 16, index_interval=index)
    r, want = oracle.decode(container, states, bits, stream, n)
    assert r == n and np.array_equal(want, data)
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    return {"stream": stream, "plan": plan, "want": want, "d_in": d_in, "d_out": d_out, "dplan": ctx.make_device_plan(plan), "n": n}


MIX = (  # container, states, bits, decoded bytes, index
    (RAW, 64, 11, 3_000_001, "wave"),
    (RAW, 64, 12, 1_000_000, 32),
    (MT, 64, 11, 2_500_000, 32),
    (RAW, 32, 11, 777_777, "wave"),
    (RAW, 64, 14, 1_500_000, "wave"),
    (RAW, 64, 10, 5_000_000, 16),
    (BLOCK, 64, 12, 600_000, 64),
    (RAW, 64, 11, 200_000, "none"),
    (MT, 64, 11, 1_200_000, 64),
    (BLOCK, 64, 12, 2_000_000, 32),
)


@pytest.mark.gpu
@pytest.mark.parametrize("k", (1, 2, 5, 8, 10))
def test_batch_of_mixed_streams_matches_the_oracle_per_stream(gpu_ctx, oracle, k):
    """K = 1, 2, 5, 8, 10 members of mixed container / width / state count / length / index kind in one call: every member's output
    equals the oracle's decode of its stream, its status word is clean, and the members the shared launches take are counted
    (raw 64-state <= 12 bits with an index: the one-chain-per-wave launch; block_/mt_ 64-state with checkpoints, two or more of one
    width: the grouped launch; everything else a launch of its own)."""
    import torch

    ms = [_member(gpu_ctx, oracle, MIX[i][0], MIX[i][1], MIX[i][2], MIX[i][3], 100 + i, MIX[i][4]) for i in range(k)]
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = batch.info()
    assert info["members"] == k
    direct = 0
    for states in (64, 32):  # (a launch per state count; a lone member of its kind keeps its own launch)
        of_kind = sum(1 for i in range(k) if MIX[i][0] == RAW and MIX[i][1] == states and MIX[i][2] <= 12 and MIX[i][4] != "none")
        direct += of_kind if of_kind >= 2 else 0
    assert info["direct_members"] == direct
    grouped = 0
    for bits in (10, 11, 12):
        of_width = sum(1 for i in range(k) if MIX[i][0] in (MT, BLOCK) and MIX[i][1] == 64 and MIX[i][2] == bits and MIX[i][4] != "none")
        grouped += of_width if of_width >= 2 else 0
    assert info["grouped_members"] == grouped
    assert info["direct_members"] + info["grouped_members"] + info["solo_members"] == k
    for rep in range(2):  # (a batch is launched again and again)
        for m in ms:
            m["d_out"].zero_()
        gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
        torch.cuda.synchronize()
        assert gpu_ctx.batch_status(batch) == [0] * k
        for i, m in enumerate(ms):
            assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"member {i} {MIX[i]} differs from the oracle (repeat {rep})"


@pytest.mark.gpu
@pytest.mark.parametrize("k,index", ((4, "wave"), (2, "wave"), (3, 32), (4, 32)))
def test_batch_of_equal_streams_one_launch(gpu_ctx, oracle, k, index):
    """The benchmark's shape at a size the oracle decodes in seconds: K streams of one size share ONE launch; bit-exact per stream."""
    import torch

    ms = [_member(gpu_ctx, oracle, RAW, 64, 11, 16_000_000, 7 + i, index) for i in range(k)]
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = batch.info()
    assert info["launches"] == 1 and info["direct_members"] == k and info["solo_members"] == 0
    gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
    torch.cuda.synchronize()
    assert gpu_ctx.batch_status(batch) == [0] * k
    for i, m in enumerate(ms):
        assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"member {i} differs from the oracle"


@pytest.mark.gpu
def test_batch_of_32_state_streams_and_both_state_counts_together(gpu_ctx, oracle):
    """rANS32x32 members share a launch of their own kind (k_decode_batch_pair: a slot's run decoded as two halves side by side, lanes
    0..31 / 32..63); together with rANS32x64 members a call makes two launches.  Index kinds: one chain per wave of a launch of its
    own, uniform, and shaped for this batch (hsrans_index_boundaries_batch: two chains per wave slot); every member against the oracle."""
    import torch

    sizes32 = [4_000_000, 4_000_000, 2_500_003, 700_001]
    ms = []
    for i, n in enumerate(sizes32):
        data = synth.enwik8_shaped(n, seed=500 + i)
        if i == 0:
            stream, plan = H.encode(RAW, 32, 11, data, index_groups=H.index_boundaries_batch(32, 11, sizes32, i, gpu_ctx))
        elif i == 1:
            stream, plan = H.encode(RAW, 32, 11, data, index_groups=H.index_boundaries(32, 11, n, gpu_ctx))
        else:
            stream, plan = H.encode(RAW, 32, (12, 10)[i % 2], data, index_interval=(16, 64)[i % 2])
        bits = 11 if i < 2 else (12, 10)[i % 2]
        r, want = oracle.decode(RAW, 32, bits, stream, n)
        assert r == n and np.array_equal(want, data)
        ms.append({"stream": stream, "want": want, "d_in": torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda(),
                   "d_out": torch.zeros(n, dtype=torch.uint8, device="cuda"), "dplan": gpu_ctx.make_device_plan(plan)})
    only32 = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = only32.info()
    assert info["launches"] == 1 and info["direct_members"] == 4 and info["solo_members"] == 0
    for rep in range(2):
        for m in ms:
            m["d_out"].fill_(0xA5 if rep else 0)
        gpu_ctx.decode_device_batch(only32, [m["d_in"] for m in ms], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
        torch.cuda.synchronize()
        assert gpu_ctx.batch_status(only32) == [0] * 4
        for i, m in enumerate(ms):
            assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"32-state member {i} differs from the oracle (repeat {rep})"
    only32.close()
    # 64- and 32-state members in ONE call: a launch per state count
    members = [_member(gpu_ctx, oracle, RAW, 64, 11, 3_000_000, 600 + i, ("wave", 32)[i]) for i in range(2)] + ms[:3]
    two = gpu_ctx.make_batch([m["dplan"] for m in members])
    info = two.info()
    assert info["launches"] == 2 and info["direct_members"] == 5
    for m in members:
        m["d_out"].zero_()
    gpu_ctx.decode_device_batch(two, [m["d_in"] for m in members], [m["d_out"] for m in members], stream_lengths=[m["stream"].size for m in members])
    torch.cuda.synchronize()
    assert gpu_ctx.batch_status(two) == [0] * 5
    for i, m in enumerate(members):
        assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"member {i} of the two-launch call differs from the oracle"


@pytest.mark.gpu
def test_batch_of_wide_histogram_streams(gpu_ctx, oracle):
    """13-15-bit members (BASELINE config 3's widths) share launches of k_decode_dual's kind — two chains per wave in one instruction
    stream, the 8-byte table at 13 bits, the rank table at 14 / 15 (a launch for each of the two table kinds; 14- and 15-bit members
    together in one).  Index kinds: one chain per wave of a launch of their own, uniform, shaped for the batch; each against the oracle."""
    import torch

    spec = [(14, 5_000_000, "batch"), (15, 5_000_000, "wave"), (14, 1_200_007, 16), (15, 2_000_000, 64), (13, 3_000_000, "wave"), (13, 999_999, 32)]
    rank_sizes = [n for b, n, _ in spec if b >= 14]
    ms = []
    for i, (bits, n, index) in enumerate(spec):
        data = synth.enwik8_shaped(n, seed=700 + i)
        if index == "batch":
            stream, plan = H.encode(RAW, 64, bits, data, index_groups=H.index_boundaries_batch(64, bits, rank_sizes, 0, gpu_ctx))
        elif index == "wave":
            stream, plan = H.encode(RAW, 64, bits, data, index_groups=H.index_boundaries(64, bits, n, gpu_ctx))
        else:
            stream, plan = H.encode(RAW, 64, bits, data, index_interval=index)
        r, want = oracle.decode(RAW, 64, bits, stream, n)
        assert r == n and np.array_equal(want, data)
        ms.append({"stream": stream, "want": want, "d_in": torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda(),
                   "d_out": torch.zeros(n, dtype=torch.uint8, device="cuda"), "dplan": gpu_ctx.make_device_plan(plan)})
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = batch.info()
    assert info["launches"] == 2 and info["direct_members"] == 6 and info["solo_members"] == 0, info
    for rep in range(2):
        for m in ms:
            m["d_out"].fill_(0x5A if rep else 0)
        gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
        torch.cuda.synchronize()
        assert gpu_ctx.batch_status(batch) == [0] * 6
        for i, m in enumerate(ms):
            assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"wide member {i} {spec[i]} differs from the oracle (repeat {rep})"


@pytest.mark.gpu
@pytest.mark.parametrize("container", (MT, BLOCK))
def test_many_small_block_streams_share_one_launch(gpu_ctx, oracle, container):
    """The many-small-files case: 12 block_/mt_ streams of 0.3 .. 3 MB (block sizes 32 .. 256 KiB, non-stationary data with single-symbol
    blocks among them) decoded by ONE grouped launch — a workgroup per block and round over all members' blocks — each against the oracle;
    then one member's block header is corrupted: that member alone reports it."""
    import torch

    ms = []
    for i in range(12):
        n = 300_000 + i * 233_333
        data = synth.nonstationary(n, seed=300 + i)
        stream, plan = H.encode(container, 64, 11, data, block_size=(1 << 15) << (i % 4), index_interval=(16, 32, 64)[i % 3])
        r, want = oracle.decode(container, 64, 11, stream, n)
        assert r == n and np.array_equal(want, data)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
        ms.append({"stream": stream, "want": want, "d_in": d_in, "d_out": torch.zeros(n, dtype=torch.uint8, device="cuda"), "dplan": gpu_ctx.make_device_plan(plan)})
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = batch.info()
    assert info["launches"] == 1 and info["grouped_members"] == 12 and info["solo_members"] == 0 and info["direct_members"] == 0
    for rep in range(3):
        for m in ms:
            m["d_out"].fill_(0xEE if rep else 0)
        gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
        torch.cuda.synchronize()
        assert gpu_ctx.batch_status(batch) == [0] * 12
        for i, m in enumerate(ms):
            assert np.array_equal(m["d_out"].cpu().numpy(), m["want"]), f"member {i} differs from the oracle (repeat {rep})"
    # a histogram count of member 5's first block changed: its sum check fails (the reference's decoder returns 0), nobody else's
    bad = ms[5]["d_in"].clone()
    bad[400] ^= 0x11  # inside the first block's 256 uint16 counts (mt_: bytes 288..799 = [n][total] [size][skip][states] then counts; block_: 280..791)
    gpu_ctx.decode_device_batch(batch, [bad if i == 5 else m["d_in"] for i, m in enumerate(ms)], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
    torch.cuda.synchronize()
    codes = gpu_ctx.batch_status(batch)
    assert codes[5] != 0 and all(c == 0 for i, c in enumerate(codes) if i != 5), codes
    for i, m in enumerate(ms):
        if i != 5:
            assert np.array_equal(m["d_out"].cpu().numpy(), m["want"])


@pytest.mark.gpu
def test_batch_reports_a_members_bad_histogram_and_decodes_the_others(gpu_ctx, oracle):
    """A member whose stream does not carry the histogram its plan's table was built from (the reference's sum check, hist.cpp:308-324,
    returns false -> its decoder returns 0): that member's status says so, the other members' outputs and statuses are untouched."""
    import torch

    ms = [_member(gpu_ctx, oracle, RAW, 64, 11, 2_000_000, 40 + i, 32) for i in range(3)]
    bad = ms[1]["d_in"].clone()
    bad[16 + 40] ^= 0x5A  # one count of the stream's histogram (raw header: [n][total][counts...])
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    gpu_ctx.decode_device_batch(batch, [ms[0]["d_in"], bad, ms[2]["d_in"]], [m["d_out"] for m in ms], stream_lengths=[m["stream"].size for m in ms])
    torch.cuda.synchronize()
    codes = gpu_ctx.batch_status(batch)
    assert codes[0] == 0 and codes[2] == 0 and codes[1] != 0, codes
    for i in (0, 2):
        assert np.array_equal(ms[i]["d_out"].cpu().numpy(), ms[i]["want"])
    assert gpu_ctx.batch_status(batch) == [0, 0, 0]  # (reported once, then cleared, like hsrans_dplan_status)


@pytest.mark.gpu
def test_batch_argument_checks(gpu_ctx, oracle):
    import torch

    ms = [_member(gpu_ctx, oracle, RAW, 64, 11, 300_000, 60 + i, 32) for i in range(2)]
    with pytest.raises(H.HsransError):
        gpu_ctx.make_batch([ms[0]["dplan"], ms[0]["dplan"]])  # a plan belongs to one member
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    small = torch.zeros(100, dtype=torch.uint8, device="cuda")
    with pytest.raises(H.HsransError):
        gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [ms[0]["d_out"], small], stream_lengths=[m["stream"].size for m in ms])
    torch.cuda.synchronize()
    assert int(ms[0]["d_out"].sum().item()) == 0, "nothing may have been launched when one member's arguments fail"


@pytest.mark.gpu
def test_queue_members_arriving_in_three_flushes(gpu_ctx, oracle):
    """hsrans_queue (round 6): streams submitted ONE BY ONE, decoded by one batch launch per flush — the reference's pool takes tasks as they
    arrive (src/thread_pool.cpp:124-133) and its loop hands it file after file (src/main.cpp:841-898, :163-170).  Ten streams of mixed kinds
    arrive in three flushes (a flush by hand, one forced by `max_members`, a last one by hand); then the queue's three ways of finding a
    batch: the very same plans again (reused as it is), OTHER streams of the same shapes (its dealing reused, only the member table
    rewritten — also after the first plans were destroyed), and a plan submitted twice (flushes first).  Every member against the oracle."""
    import torch
    ctx = gpu_ctx

    def member(container, states, bits, n, seed, **kw):
        d = synth.enwik8_shaped(n, seed=seed)
        if kw:
            s, plan = H.encode(container, states, bits, d, **kw)
        else:  # a stream without an index: a member with a launch of its own behind the shared ones
            s = H.encode(container, states, bits, d)
            plan = H.plan_build(container, states, bits, s)
        r, want = oracle.decode({H.RAW: RAW, H.MT: MT, H.BLOCK: BLOCK}[container], states, bits, s, n)
        assert r == n and np.array_equal(want, d)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16 + 16, np.uint8)])).cuda()
        return {"want": want, "len": s.size, "d_in": d_in, "out": torch.zeros(n, dtype=torch.uint8, device="cuda"), "dplan": ctx.make_device_plan(plan)}

    def check(ms):
        torch.cuda.synchronize()
        for m in ms:
            assert ctx.status(m["dplan"]) == 0
            assert np.array_equal(m["out"].cpu().numpy(), m["want"])
            m["out"].zero_()

    mixed = [member(H.RAW, 64, 11, 3_000_000, 1, index_interval=16), member(H.RAW, 64, 12, 1_500_000, 2, index_interval=8),
             member(H.MT, 64, 11, 2_000_000, 3, index_interval=16, block_size=65536), member(H.RAW, 32, 11, 2_000_000, 4, index_interval=16),
             member(H.RAW, 64, 14, 2_500_000, 5, index_interval=16), member(H.RAW, 64, 11, 700_001, 6, index_interval=4),
             member(H.MT, 64, 11, 1_000_000, 7, index_interval=8, block_size=65536), member(H.RAW, 64, 11, 5_000_000, 8, index_interval=32),
             member(H.RAW, 64, 10, 900_000, 9), member(H.RAW, 64, 11, 2_000_000, 10, index_interval=16)]
    q = H.api.Queue(ctx, max_members=4)
    for m in mixed[:3]:
        q.submit(m["dplan"], m["d_in"], m["out"], stream_length=m["len"])
    assert q.pending() == 3 and q.stats()["flushes"] == 0  # nothing launched yet
    q.flush()
    for m in mixed[3:7]:  # the fourth submission flushes by itself
        q.submit(m["dplan"], m["d_in"], m["out"], stream_length=m["len"])
    assert q.pending() == 0 and q.stats()["flushes"] == 2
    for m in mixed[7:]:
        q.submit(m["dplan"], m["d_in"], m["out"], stream_length=m["len"])
    q.flush()
    q.flush()  # nothing pending: nothing happens
    st = q.stats()
    assert st["submitted"] == 10 and st["flushes"] == 3 and st["batches_made"] == 3
    check(mixed)
    # the same plans in the same order: their batch as it is
    for m in mixed[:3]:
        q.submit(m["dplan"], m["d_in"], m["out"], stream_length=m["len"])
    q.flush()
    assert q.stats()["batches_reused"] == 1 and q.stats()["retargeted"] == 0 and q.stats()["batches_made"] == 3
    check(mixed[:3])
    q.close()

    # a loop over files of one size class: every flush brings OTHER streams of the same shapes
    q = H.api.Queue(ctx, max_members=8)
    for rnd in range(3):
        files = [member(H.RAW, 64, 11, 4_000_000, 100 + 4 * rnd + k, index_interval=16) for k in range(4)]
        for m in files:
            q.submit(m["dplan"], m["d_in"], m["out"], stream_length=m["len"])
        q.flush()
        check(files)
        del files  # (the plans are destroyed: the next round's may sit at the same addresses)
    st = q.stats()
    assert st["batches_made"] == 1 and st["batches_reused"] == 2 and st["retargeted"] == 2, st
    # a plan that is still pending when it is submitted again: the first batch goes out first
    a, b = member(H.RAW, 64, 11, 1_000_000, 200, index_interval=8), member(H.RAW, 64, 11, 1_000_000, 201, index_interval=8)
    out2 = torch.zeros_like(a["out"])
    q.submit(a["dplan"], a["d_in"], a["out"], stream_length=a["len"])
    q.submit(b["dplan"], b["d_in"], b["out"], stream_length=b["len"])
    flushes = q.stats()["flushes"]
    q.submit(a["dplan"], a["d_in"], out2, stream_length=a["len"])
    assert q.stats()["flushes"] == flushes + 1 and q.pending() == 1
    q.flush()
    check([a, b])
    assert np.array_equal(out2.cpu().numpy(), a["want"])
    # arguments
    with pytest.raises(H.HsransError):
        q.submit(a["dplan"], a["d_in"][1:], a["out"], stream_length=a["len"])  # misaligned stream
    with pytest.raises(H.HsransError):
        q.submit(a["dplan"], a["d_in"], a["out"][: 1000], stream_length=a["len"])  # output too small
    q.close()
