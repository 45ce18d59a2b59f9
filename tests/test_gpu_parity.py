"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT, RAW

pytestmark = pytest.mark.gpu

LENS_SMALL = (1, 31, 62, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4096, 65536, 65600, 131073)


@pytest.fixture(scope="module")
def zipf():
    return synth.enwik8_shaped(1 << 20, seed=11)


@pytest.fixture(scope="module")
def nonstat():
    return synth.nonstationary(3_000_000)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_raw_single_chain_every_length(gpu_ctx, oracle, zipf, states, bits):
    for n in LENS_SMALL + (1 << 20,):
        d = zipf[:n]
        s = H.encode(H.RAW, states, bits, d)
        r0, want = oracle.decode(RAW, states, bits, s, n)
        assert r0 == n and np.array_equal(want, d)
        r, got = gpu_ctx.decode_host(H.RAW, states, bits, s, n)
        assert r == n, (states, bits, n)
        assert np.array_equal(got, want), (states, bits, n, int(np.argmax(got != want)))


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
@pytest.mark.parametrize("interval", (4, 64, 256))
def test_raw_indexed(gpu_ctx, oracle, zipf, states, bits, interval):
    for n in (63, 64, 65, 1000, 65600, 300_001, 1 << 20):
        d = zipf[:n]
        s, plan = H.encode(H.RAW, states, bits, d, index_interval=interval)
        r, got = gpu_ctx.decode_host(H.RAW, states, bits, s, n, plan=plan)
        assert r == n and np.array_equal(got, d), (states, bits, interval, n)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_raw_one_chain_per_wave_odd_shapes(gpu_ctx, oracle, nonstat, states, bits):
    """The one-chain-per-wave launches (k_decode_direct, the two-chain kernel at 13-15 bits, the 32-state pair loop): their
    hand-scheduled loops re-base the stream cursor every 4 groups and wrap the LDS ring — so: few LONG chains (every ring
    wraps many times), chains of unequal length (checkpoints sit on multiples of 4 groups: the API's rule), partial last groups, a
    single chain pair, and the device's own boundaries at sizes that leave most waves without a chain."""
    rng = np.random.default_rng(1000 * states + bits)
    for n in (70_001, 1_000_003, 3_000_000):
        d = nonstat[:n]
        groups_total = n // states
        cases = [H.index_boundaries(states, bits, n, gpu_ctx)]
        # a handful of long chains of unequal length
        for k in (1, 2, 5, 33):
            if groups_total > 8 * (k + 1):
                cuts = (np.unique(rng.integers(1, groups_total // 4 - 1, size=k)) * 4).astype(np.uint64)
                cases.append(cuts)
        cases.append((np.arange(1, min(64, groups_total // 8)) * 8).astype(np.uint64))  # many tiny chains at the front, one long one behind
        if n == 3_000_000:
            cases.append((np.arange(1, groups_total // 4) * 4).astype(np.uint64))  # more chains than the device has waves: every wave takes several
        for g in cases:
            if g.size == 0:
                continue
            s, plan = H.encode(H.RAW, states, bits, d, index_groups=g)
            r0, want = oracle.decode(RAW, states, bits, s, n)
            assert r0 == n and np.array_equal(want, d)
            got = gpu_ctx.decode(H.RAW, states, bits, s, plan=plan)
            assert got.size == n and np.array_equal(got, want), (states, bits, n, g[:6], int(np.argmax(got != want)) if got.size == n else got.size)


@pytest.mark.parametrize("container", (BLOCK, MT))
@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_containers_own_encoder(gpu_ctx, oracle, zipf, nonstat, container, states, bits):
    for src, n in ((zipf, 64), (zipf, 1000), (zipf, 65536), (zipf, 65600), (zipf, 200_001), (nonstat, 3_000_000), (nonstat, 1_000_003)):
        d = src[:n]
        s = H.encode(container, states, bits, d)
        r0, want = oracle.decode(container, states, bits, s, n)
        assert r0 == n and np.array_equal(want, d)
        r, got = gpu_ctx.decode_host(container, states, bits, s, n)
        assert r == n and np.array_equal(got, want), (container, states, bits, n)
        s2, plan = H.encode(container, states, bits, d, index_interval=64, block_size=1 << 16)
        r, got = gpu_ctx.decode_host(container, states, bits, s2, n, plan=plan)
        assert r == n and np.array_equal(got, d), ("indexed", container, states, bits, n)


@pytest.mark.parametrize("container", (RAW, BLOCK, MT))
@pytest.mark.parametrize("states", (32, 64))
def test_reference_encoded_streams(gpu_ctx, oracle, ref, zipf, nonstat, container, states):
    """Streams written by the REAL reference encoder (only where oracle/_ref is present), incl. the quirk lengths."""
    for bits in (10, 11, 12, 13, 14, 15):
        for src, n in ((zipf, 65537), (zipf, 65560), (zipf, 65599), (zipf, 65600), (zipf, 131073), (zipf, 524300), (nonstat, 3_000_000)):
            d = src[:n]
            s = ref.encode(container, states, bits, d)
            r0, want = oracle.decode(container, states, bits, s, n)
            r, got = gpu_ctx.decode_host(container, states, bits, s, n)
            assert r == r0 and np.array_equal(got, want), (container, states, bits, n)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (11, 13, 15))
def test_streams_that_renormalise_on_nearly_every_step(gpu_ctx, oracle, states, bits):
    """The most stream a group can consume: a histogram in which one symbol holds nearly all the probability and data made of the
    255 others only (frequency 1 of 2^bits each: `bits` bits per symbol, so nearly every lane reads a word in nearly every group).
    The hand-scheduled loops re-base their cursors and look for chunk crossings once per 4 groups (<= 512 stream bytes) and time their
    waits by crossings: this is the input that crosses a chunk in almost every iteration, from the first one on."""
    rng = np.random.default_rng(5 + bits)
    n = 3_000_003
    d = rng.integers(0, 255, n, dtype=np.uint8)
    counts = np.ones(256, np.int64)
    counts[255] = (1 << bits) - 255
    hist = H.api.hist_from_counts(counts)
    room = 2 * n + 4096  # (the data expands: `bits` bits per byte)
    plans = [H.encode(H.RAW, states, bits, d, hist=hist, index_interval=g, out_capacity=room) for g in (4, 32)]
    plans.append(H.encode(H.RAW, states, bits, d, hist=hist, index_groups=H.index_boundaries(states, bits, n, gpu_ctx), out_capacity=room))
    for k, (s, plan) in enumerate(plans):
        assert s.size > n * bits // 8  # (it really is that incompressible)
        r0, want = oracle.decode(RAW, states, bits, s, n)
        assert r0 == n and np.array_equal(want, d)
        r, got = gpu_ctx.decode_host(H.RAW, states, bits, s, n, plan=plan)
        assert r == n and np.array_equal(got, d), (k,)


def test_failure_modes(gpu_ctx, zipf):
    d = zipf[:10000]
    s = H.encode(H.RAW, 64, 11, d)
    assert gpu_ctx.decode_host(H.RAW, 64, 11, s, 9999)[0] == 0  # outCapacity < decodedLength (rANS32x64_16w.cpp:180)
    assert gpu_ctx.decode_host(H.RAW, 64, 11, s, 10000, in_length=s.size - 1)[0] == 0  # inLength < stored (:186)
    assert gpu_ctx.decode_host(H.RAW, 64, 11, s[:100], 10000)[0] == 0  # shorter than a header (:171)
    assert gpu_ctx.decode_host(H.RAW, 64, 12, s, 10000)[0] == 0  # wrong bits: histogram sum check (hist.cpp:340)
    bad = s.copy()
    bad[16] ^= 1  # corrupt one count -> sum != 2^bits
    assert gpu_ctx.decode_host(H.RAW, 64, 11, bad, 10000)[0] == 0
    for c in (H.BLOCK, H.MT):
        s = H.encode(c, 64, 11, d)
        assert gpu_ctx.decode_host(c, 64, 12, s, 10000)[0] == 0


def test_index_build_on_gpu(gpu_ctx, oracle, zipf):
    d = zipf
    s = H.encode(H.RAW, 64, 11, d)
    plan = gpu_ctx.index_build(H.RAW, 64, 11, s, 64)
    s2, plan2 = H.encode(H.RAW, 64, 11, d, index_interval=64)
    assert np.array_equal(s, s2)
    assert np.array_equal(plan, plan2), "GPU-built index differs from the encoder's"
    r, got = gpu_ctx.decode_host(H.RAW, 64, 11, s, d.size, plan=plan)
    assert r == d.size and np.array_equal(got, d)


def test_device_entry_and_graph(gpu_ctx, zipf):
    import torch

    d = zipf
    s, plan = H.encode(H.RAW, 64, 11, d, index_interval=64)
    pad = (-s.size) % 16
    d_in = torch.from_numpy(np.concatenate([s, np.zeros(pad, np.uint8)])).cuda()
    d_out = torch.zeros(d.size, dtype=torch.uint8, device="cuda")
    dp = gpu_ctx.make_device_plan(plan)
    gpu_ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
    assert gpu_ctx.status(dp) == 0
    assert np.array_equal(d_out.cpu().numpy(), d)
    info = dp.launch_info()
    assert info["shared_table"] == 1 and info["chains"] == H.plan_chain_count(plan)
    # hipGraph capture of the launch (no allocation / sync inside hsrans_decode_device)
    d_out.zero_()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            gpu_ctx.decode_device(dp, d_in, d_out, stream=side, stream_length=s.size)
    torch.cuda.synchronize()
    d_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), d)


@pytest.mark.parametrize("bits", (11, 12, 14, 15))
def test_full_size_100mb_round_trip(gpu_ctx, bits):
    """BASELINE.json configs[1] and [2] at full size: encode -> decode equals the input (size-independent property);
    the oracle itself is compared on a 1 MiB prefix-stream of the same generator in the tests above."""
    import hashlib

    import torch

    n = 100_000_000
    data = synth.enwik8_shaped(n, seed=20241008)
    s, plan = H.encode(H.RAW, 64, bits, data, index_interval=32)
    pad = (-s.size) % 16
    d_in = torch.from_numpy(np.concatenate([s, np.zeros(pad, np.uint8)])).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dp = gpu_ctx.make_device_plan(plan)
    gpu_ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
    assert gpu_ctx.status(dp) == 0
    got = d_out.cpu().numpy()
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(data.tobytes()).hexdigest()
    # the un-indexed single-chain plan of the same stream gives the same bytes (checked on the first 4 MiB to keep it short)
    m = 4 << 20
    s1 = H.encode(H.RAW, 64, bits, data[:m])
    r, got1 = gpu_ctx.decode_host(H.RAW, 64, bits, s1, m)
    assert r == m and np.array_equal(got1, data[:m])


def test_slice_upload_is_enough_for_a_slice(gpu_ctx, zipf, nonstat):
    """A rank that holds only its WINDOW of the stream (hsrans_plan_stream_ranges body, a buffer as long as the window, not as
    the stream) decodes its chains correctly through hsrans_decode_device_window; a raw stream's shared histogram is not in
    the window at all (the plan carries its copy)."""
    import torch
    from hypersonic_rans_amd import sharded
    for container, src in ((H.RAW, zipf), (H.MT, nonstat[:1_500_000]), (H.BLOCK, nonstat[:1_500_000])):
        d = src
        stream, plan = H.encode(container, 64, 11, d, index_interval=32, block_size=0 if container == H.RAW else 65536)
        runs = sharded.shard_chains(plan, 3)
        out = torch.zeros(d.size, dtype=torch.uint8, device="cuda")
        uploaded = 0
        for first, count in runs:
            (hb, he), (bb, be) = H.plan_stream_ranges(plan, first, count)
            lo = bb & ~15
            window = torch.full((be - lo + 16,), 0xEE, dtype=torch.uint8, device="cuda")
            window[bb - lo: be - lo] = torch.from_numpy(stream[bb:be]).cuda()
            uploaded += be - bb
            dplan = gpu_ctx.make_device_plan(H.plan_slice(plan, first, count))
            gpu_ctx.decode_device_window(dplan, window, lo, be - lo, out)
            assert gpu_ctx.status(dplan) == 0
        assert torch.equal(out.cpu(), torch.from_numpy(d)), container
        assert uploaded < 1.1 * stream.size


def test_output_window_and_window_lower_edge(gpu_ctx, nonstat):
    """hsrans_decode_device_ranges: a rank holds only ITS range of the output (and its window of the stream); a window that starts
    above the lowest byte the plan's chains read, or an output range the chains do not fit in, is refused (HSRANS_E_FORMAT = 3)."""
    import torch
    from hypersonic_rans_amd import sharded
    d = nonstat[:1_200_000]
    stream, plan = H.encode(H.MT, 64, 11, d, index_interval=32, block_size=65536)
    layout = sharded.ShardLayout(plan, 3, parts=2)
    for rank in range(3):
        b, e = layout.ranges[rank]
        lo, hi = layout.windows[rank]
        window = torch.full((hi - lo + 16,), 0xEE, dtype=torch.uint8, device="cuda")
        window[:hi - lo] = torch.from_numpy(stream[lo:hi]).cuda()
        # a canary behind (and in front of) the rank's range: nothing a kernel rounds up may leave the window (the kernels'
        # own output bound is now the window's end, not the end of the whole output)
        guard = 4096
        buf = torch.full((guard + e - b + guard,), 0xCC, dtype=torch.uint8, device="cuda")
        out = buf[guard:guard + e - b]
        for f, c in layout.sub_runs[rank]:
            dplan = gpu_ctx.make_device_plan(H.plan_slice(plan, f, c))
            gpu_ctx.decode_device_ranges(dplan, window, lo, hi - lo, out, b, e - b)
            assert gpu_ctx.status(dplan) == 0
        assert torch.equal(out.cpu(), torch.from_numpy(d[b:e])), rank
        assert bool((buf[:guard] == 0xCC).all()) and bool((buf[guard + e - b:] == 0xCC).all()), f"rank {rank} wrote outside its output window"
    # refusals: rank 1's plan with a window that begins 16 bytes too late / an output range that begins too late or is too short
    f, c = layout.runs[1]
    b, e = layout.ranges[1]
    lo, hi = layout.windows[1]
    dplan = gpu_ctx.make_device_plan(H.plan_slice(plan, f, c))
    window = torch.zeros(hi - lo + 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(e - b + 64, dtype=torch.uint8, device="cuda")
    (_hb, _he), (bb, _be) = H.plan_stream_ranges(plan, f, c)
    too_late = (bb & ~15) + 16
    L = gpu_ctx.L
    s = torch.cuda.current_stream().cuda_stream
    import ctypes
    rc = L.hsrans_decode_device_ranges(gpu_ctx.handle, dplan.handle, window.data_ptr(), too_late, hi - too_late, out.data_ptr(), b, e - b, ctypes.c_void_p(s))
    assert rc == 3
    rc = L.hsrans_decode_device_window(gpu_ctx.handle, dplan.handle, window.data_ptr(), too_late, hi - too_late, torch.zeros(d.size, dtype=torch.uint8, device="cuda").data_ptr(),
                                       d.size, ctypes.c_void_p(s))
    assert rc == 3
    rc = L.hsrans_decode_device_ranges(gpu_ctx.handle, dplan.handle, window.data_ptr(), lo, hi - lo, out.data_ptr(), b + 64, e - b, ctypes.c_void_p(s))
    assert rc == 3
    rc = L.hsrans_decode_device_ranges(gpu_ctx.handle, dplan.handle, window.data_ptr(), lo, hi - lo, out.data_ptr(), b, e - b - 64, ctypes.c_void_p(s))
    assert rc == 3
    torch.cuda.synchronize()


def test_pipelined_host_decode(gpu_ctx, nonstat, zipf):
    """Upload / decode / download overlapped over slices of the plan (pinned host buffers): same bytes as one decode."""
    import torch
    from hypersonic_rans_amd import pipeline
    for container, d, kw in ((H.MT, nonstat, dict(block_size=65536, index_interval=32)), (H.RAW, zipf, dict(index_interval=64)),
                             (H.MT, nonstat[:500_000], dict(block_size=65536))):
        out = H.encode(container, 64, 11, d, **kw)
        stream, plan = out if isinstance(out, tuple) else (out, H.plan_build(container, 64, 11, out))
        host_stream = torch.from_numpy(stream).pin_memory()
        host_out = torch.full((d.size,), 0xCC, dtype=torch.uint8).pin_memory()
        for k in (1, 3, 8):
            dec = pipeline.PipelinedHostDecoder(gpu_ctx, plan, n_slices=k)
            host_out.fill_(0xCC)
            dec.decode(host_stream, host_out)
            assert np.array_equal(host_out.numpy(), d), (container, k)
            dec.decode(host_stream, host_out)  # reusable
            assert np.array_equal(host_out.numpy(), d), (container, k)
            # pageable output (and stream): the staged path — device-side output buffer, copied down slice by slice
            pageable = torch.full((d.size,), 0xCC, dtype=torch.uint8)
            dec.decode(torch.from_numpy(stream.copy()), pageable)
            assert np.array_equal(pageable.numpy(), d), (container, k, "pageable")
        # the one-call host entry with a page-locked output: the kernel stores straight into it
        host_out.fill_(0xCC)
        r = gpu_ctx.L.hsrans_decode_host(gpu_ctx.handle, container, 64, 11, host_stream.data_ptr(), host_stream.numel(), host_out.data_ptr(), host_out.numel(),
                                         H.api._p(plan), plan.size)
        assert r == d.size and np.array_equal(host_out.numpy(), d)
        host_out.fill_(0xCC)
        pipeline.decode_from_host_unpipelined(gpu_ctx, plan, host_stream, host_out)
        assert np.array_equal(host_out.numpy(), d)


def test_a_process_second_pipeline_is_as_fast_as_its_first(gpu_ctx):
    """The pipelines' three streams belong to the context.  Streams made per pipe put every pipe after a process's first on ONE
    hardware queue: its legs ran one after the other (24-26 instead of 33+ k MiB/s in the harness), bit-exact all the same — only a
    rate can see it.  Loose bound: a serialised pipeline is 30 % slower, not 15."""
    import time
    import torch
    from hypersonic_rans_amd import pipeline

    n = 64 << 20
    g = torch.Generator(device="cuda").manual_seed(3)
    rates = []
    for k in range(3):
        d_in = torch.rand(n, device="cuda", generator=g).pow_(6).mul_(205).to(torch.uint8)
        d_enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
        m, dplan = gpu_ctx.encode_device(H.MT, 64, 11, d_in, d_enc, block_size=1 << 18, index_interval=32, want_plan=True)
        plan = gpu_ctx.read_device_plan(dplan, capacity=1 << 28)
        host_stream = torch.empty(m, dtype=torch.uint8).pin_memory()
        host_stream.copy_(d_enc[:m])
        host_out = torch.empty(n, dtype=torch.uint8).pin_memory()
        ref = d_in.cpu()
        dec = pipeline.PipelinedHostDecoder(gpu_ctx, plan)
        dec.decode(host_stream, host_out)
        assert torch.equal(host_out, ref)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            dec.decode(host_stream, host_out)
            best = min(best, time.perf_counter() - t0)
        rates.append(n / best)
        del dec, dplan
    assert min(rates[1:]) > 0.85 * rates[0], [round(r / 1e9, 1) for r in rates]


@pytest.mark.parametrize("one_launch", (True, False))
@pytest.mark.parametrize("world,root,parts,root_share", ((2, None, 3, 0.0), (8, 0, 4, 0.66), (4, 2, 2, 0.0), (8, None, 1, 0.0)))
def test_every_ranks_gpu_side_of_a_sharded_decode_on_one_gpu(gpu_ctx, nonstat, monkeypatch, world, root, parts, root_share, one_launch):
    """No multi-GPU node here, so the N > 1 path is checked in two halves: the exchange logic on gloo (tests/test_sharded_gloo.py,
    test_bench_contract.py) and — this test — the GPU side of EVERY rank on the one GPU: each rank's slice plans, stream window,
    output window (a rank that is not the root of a gather holds only its own range), sub-runs and weighted shares, launched
    through hsrans_decode_device_ranges exactly as ShardedDecoder.step does; the exchange is replaced by copies of the ranks'
    sub-ranges into the root's buffer, in the order pipelined_gather posts them."""
    import torch
    from hypersonic_rans_amd import sharded
    # round 6: a rank's sub-runs are ONE launch with a completion word per sub-run (mt_rANS32x64_16w_decode.cpp:182-224 hands every block to
    # the pool in one pass); HSRANS_SHARD_ONE_LAUNCH=0 keeps round 5's launch per sub-run — both sides run here
    monkeypatch.setenv("HSRANS_SHARD_ONE_LAUNCH", "1" if one_launch else "0")
    d = nonstat[:3_000_000]
    stream, plan = H.encode(H.MT, 64, 11, d, index_interval=32, block_size=65536)
    weights = sharded.root_weights(world, root, root_share) if (root is not None and root_share) else None
    decs = [sharded.ShardedDecoder(gpu_ctx, plan, parts=parts, weights=weights, root=root, world=world, rank=r) for r in range(world)]
    for dec in decs:
        assert bool(dec.c.info["one_launch"]) == (one_launch and parts > 1 and dec.count > 0), (dec.rank, dec.c.info)
    outs = []
    for dec in decs:
        d_window = dec.upload_window(stream, "cuda")
        out = dec.alloc_out("cuda")
        if root is not None and dec.rank != root:
            assert out.numel() == max(dec.out_len, 4) and dec.out_len == dec.ranges[dec.rank][1] - dec.ranges[dec.rank][0] < d.size
        dec.step(d_window, out, gather=False)  # the rank's sub-runs, no exchange
        dec.check()
        outs.append(out)
    torch.cuda.synchronize()
    receivers = range(world) if root is None else (root,)
    for recv in receivers:
        full = outs[recv].clone()
        for k in range(parts):
            for src in range(world):
                if src == recv:
                    continue
                b, e = decs[src].layout.sub_ranges[src][k]
                if e > b:
                    full[b - decs[recv].out_base:e - decs[recv].out_base] = outs[src][b - decs[src].out_base:e - decs[src].out_base]
        assert torch.equal(full[:d.size].cpu(), torch.from_numpy(d)), (world, root, recv)
    if root_share:
        share = (decs[root].ranges[root][1] - decs[root].ranges[root][0]) / d.size
        assert abs(share - root_share) < 0.05


@pytest.mark.parametrize("one_launch", (True, False))
@pytest.mark.parametrize("bits,block,interval,size", ((12, 65536, 32, 6_000_000), (14, 1 << 17, 16, 6_000_000), (11, 1 << 18, 8, 24_000_000), (14, 1 << 18, 8, 24_000_000)))
def test_a_ranks_sub_runs_announce_their_completion(gpu_ctx, monkeypatch, one_launch, bits, block, interval, size):
    """hsrans_sharded_wait_part — what the exchange's stream does inside hsrans_decode_sharded, reachable on one GPU: a SECOND stream waits
    for sub-run k of the decode queued on the first (its completion word, published by the one launch that decodes all sub-runs:
    hipStreamWaitValue32; or the event behind sub-run k's own launch) and copies the sub-run's range away at once.  Only the second
    stream is synchronised: every copied range must already hold the decoded bytes (the output is poisoned before every step), i.e. a
    completion word never fires before its sub-run's stores are visible device-wide.  Checked against the source bytes (the streams
    round-trip through the oracle in the CPU suite).  The three cases reach the three kernels that count into sub-runs: the grouped
    launch with the 8-byte table, with the rank table (14 bits), and the spread launch (few large blocks, many chains)."""
    import torch
    from hypersonic_rans_amd import sharded
    monkeypatch.setenv("HSRANS_SHARD_ONE_LAUNCH", "1" if one_launch else "0")
    d = synth.nonstationary(size, seed=5) if size < 10_000_000 else synth.enwik8_shaped(size, seed=5)
    stream, plan = H.encode(H.MT, 64, bits, d, index_interval=interval, block_size=block)
    src = torch.from_numpy(d).cuda()
    world, parts = 2, 4
    side = torch.cuda.Stream()
    for rank in range(world):
        dec = sharded.ShardedDecoder(gpu_ctx, plan, parts=parts, world=world, rank=rank)
        assert bool(dec.c.info["one_launch"]) == one_launch
        d_window = dec.upload_window(stream, "cuda")
        out = dec.alloc_out("cuda")
        if one_launch:
            assert bool(dec.c.whole_plan() is not None)
        for rep in range(4):
            out.fill_(0xA5)
            copy = torch.full_like(out, 0x5A)
            torch.cuda.synchronize()
            dec.step(d_window, out, gather=False)
            for k in reversed(range(parts)) if rep & 1 else range(parts):  # (the waits need not come in the sub-runs' order)
                b, e = dec.layout.sub_ranges[rank][k]
                if e > b:
                    dec.c.wait_part(k, side)
                    with torch.cuda.stream(side):
                        copy[b:e].copy_(out[b:e], non_blocking=True)
            side.synchronize()
            b, e = dec.ranges[rank]
            assert torch.equal(copy[b:e], src[b:e]), (rank, rep)
            torch.cuda.synchronize()
            dec.check()
        if one_launch and size > 10_000_000:
            assert dec.launch_info()["spread"] == 2  # (k_decode_dealt counting its workgroups into the sub-runs)


def test_sharded_decode_single_rank_over_rccl(gpu_ctx, zipf):
    """decode_sharded on the `nccl` backend (= RCCL) with a world of one rank: the communicator is created on the GPU and the
    status all-reduce runs through RCCL (the N>1 exchange logic runs on CPU under gloo in tests/test_sharded_gloo.py)."""
    import os

    import torch
    import torch.distributed as dist

    from hypersonic_rans_amd import sharded

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        seen = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(seen)  # RCCL executes
        assert int(seen.item()) == 1 and dist.get_backend() == "nccl"
        for container in (H.RAW, H.MT):
            s, plan = H.encode(container, 64, 11, zipf, index_interval=32)
            pad = (-s.size) % 16
            d_in = torch.from_numpy(np.concatenate([s, np.zeros(pad, np.uint8)])).cuda()
            out = sharded.decode_sharded(gpu_ctx, d_in, s.size, plan, gather=True)
            assert np.array_equal(out.cpu().numpy(), zipf)
            out = sharded.decode_sharded_from_host(gpu_ctx, s, plan, gather=True)  # stream in host memory, window upload on a side stream
            assert np.array_equal(out.cpu().numpy(), zipf)
            dec = sharded.ShardedDecoder(gpu_ctx, plan)
            out = torch.zeros(zipf.size, dtype=torch.uint8, device="cuda")
            for _ in range(3):  # prepared once, decoded repeatedly
                dec.decode(d_in, out)
            assert dec.global_status() == 0 and np.array_equal(out.cpu().numpy(), zipf)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("states", (32, 64))
def test_device_side_mt_walk_matches_host_planner(gpu_ctx, ref, zipf, nonstat, states):
    """K2: the mt_ header chain followed on the GPU gives the same chains as the host planner, on streams written by our
    encoder and by the real reference (incl. a quirk length), and decodes bit-exactly."""
    import torch

    cases = [(H.encode(H.MT, states, 11, nonstat[:2_000_003]), nonstat[:2_000_003], 11)]
    for bits, n in ((11, 65560), (14, 1 << 20), (11, 3_000_000)):
        src = zipf if n <= zipf.size else nonstat
        cases.append((ref.encode(MT, states, bits, src[:n]), None, bits))
    for s, data, bits in cases:
        n = int(s[:8].view(np.uint64)[0])
        host_plan = H.plan_build(H.MT, states, bits, s)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        dp = gpu_ctx.make_device_plan_from_stream(H.MT, states, bits, d_in, s.size, n)
        dev_plan = gpu_ctx.read_device_plan(dp, capacity=host_plan.size + 4096)
        _, cf_h, pc_h = H.api.plan_tables(host_plan)
        hd, cf_d, pc_d = H.api.plan_tables(dev_plan)
        assert hd["n_chains"] == H.plan_chain_count(host_plan) and np.array_equal(cf_h, cf_d)
        assert pc_h.tobytes() == pc_d.tobytes()
        so = 64 + ((hd["n_chains"] + 1) * 4 + 15) // 16 * 16 + 48 * hd["n_pieces"]
        assert np.array_equal(host_plan[so:so + 4 * states * hd["n_chains"]], dev_plan[so:so + 4 * states * hd["n_chains"]])
        d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
        assert gpu_ctx.status(dp) == 0
        r, want = gpu_ctx.decode_host(H.MT, states, bits, s, n)
        assert r == n and np.array_equal(d_out.cpu().numpy(), want)
        if data is not None:
            assert np.array_equal(want, data)
    bad = cases[0][0].copy()
    bad[16 + 16 + 4 * states] ^= 1  # first block's first count
    d_bad = torch.from_numpy(np.concatenate([bad, np.zeros((-bad.size) % 16, np.uint8)])).cuda()
    with pytest.raises(H.HsransError):
        gpu_ctx.make_device_plan_from_stream(H.MT, states, 11, d_bad, bad.size, 2_000_003)


@pytest.mark.parametrize("states", (32, 64))
def test_index_build_for_reference_mt_streams(gpu_ctx, ref, nonstat, zipf, states):
    """An mt_ stream written by the real reference gets in-block checkpoints from one GPU pass; the indexed plan decodes
    to the same bytes through the grouped launch."""
    for bits, src, n in ((11, nonstat, 3_000_000), (14, zipf, 1 << 20)):
        d = src[:n]
        s = ref.encode(MT, states, bits, d)
        plan = gpu_ctx.index_build(H.MT, states, bits, s, 32)
        assert H.plan_chain_count(plan) > 4 * H.plan_chain_count(H.plan_build(H.MT, states, bits, s))
        r, got = gpu_ctx.decode_host(H.MT, states, bits, s, n, plan=plan)
        assert r == n and np.array_equal(got, d)
        # and it is the plan our encoder writes for the same stream (same block policy -> same bytes -> same checkpoints)
        s2, plan2 = H.encode(H.MT, states, bits, d, index_interval=32)
        assert np.array_equal(s, s2) and np.array_equal(plan, plan2)


@pytest.mark.parametrize("states", (32, 64))
def test_first_decode_of_an_unindexed_stream_leaves_the_index_behind(gpu_ctx, ref, nonstat, zipf, states):
    """hsrans_decode_device_indexing: a reference-emitted mt_ stream that only exists in device memory is planned on the device
    (K2), decoded once with one chain per block — that pass records the checkpoints — and decoded again with the plan it left
    behind, which is byte for byte the one hsrans_index_build makes from a host copy of the stream.  Raw streams likewise."""
    import torch

    for container, bits, src, n, interval in ((H.MT, 11, nonstat, 3_000_000, 32), (H.MT, 14, zipf, 1 << 20, 64), (H.MT, 11, zipf, 65560, 4),
                                              (H.RAW, 11, zipf, 400_037, 32)):
        d = src[:n]
        s = ref.encode(MT if container == H.MT else RAW, states, bits, d)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        if container == H.MT:
            base = gpu_ctx.make_device_plan_from_stream(H.MT, states, bits, d_in, s.size, n)
        else:
            base = gpu_ctx.make_device_plan(H.plan_build(H.RAW, states, bits, s))
        first = torch.full((n + 64,), 0xCC, dtype=torch.uint8, device="cuda")
        indexed = gpu_ctx.decode_device_indexing(base, d_in, first[:n], interval, stream_length=s.size)
        r, want = gpu_ctx.decode_host(container, states, bits, s, n)  # (65560 is a length whose last block the reference itself mis-decodes)
        assert r == n and (n == 65560 or np.array_equal(want, d))
        assert np.array_equal(first[:n].cpu().numpy(), want) and bool((first[n:] == 0xCC).all())
        want_plan = gpu_ctx.index_build(container, states, bits, s, interval)
        assert np.array_equal(gpu_ctx.read_device_plan(indexed, capacity=want_plan.size + 4096), want_plan)
        assert H.plan_chain_count(want_plan) > H.plan_chain_count(gpu_ctx.read_device_plan(base, capacity=want_plan.size + 4096))
        again = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(indexed, d_in, again, stream_length=s.size)
        assert gpu_ctx.status(indexed) == 0 and np.array_equal(again.cpu().numpy(), want)
    # arguments: an interval that is not a multiple of 4, a plan that has checkpoints already
    with pytest.raises(H.HsransError):
        gpu_ctx.decode_device_indexing(base, d_in, first[:n], 6, stream_length=s.size)
    with pytest.raises(H.HsransError):
        gpu_ctx.decode_device_indexing(indexed, d_in, first[:n], 32, stream_length=s.size)
    # a corrupted block: reported as a device error, no plan
    s_mt = ref.encode(MT, states, 11, nonstat[:3_000_000])
    host_plan = H.plan_build(H.MT, states, 11, s_mt)
    bad = s_mt.copy()
    _, _, pc = H.api.plan_tables(host_plan)
    bad[int(pc[1]["hist_off"]) + 2] ^= 0x40  # a count of the second block's histogram: its sum is no longer 2^bits
    d_bad = torch.from_numpy(np.concatenate([bad, np.zeros((-bad.size) % 16, np.uint8)])).cuda()
    base = gpu_ctx.make_device_plan(host_plan)
    out = torch.zeros(3_000_000, dtype=torch.uint8, device="cuda")
    with pytest.raises(H.HsransError):
        gpu_ctx.decode_device_indexing(base, d_bad, out, 32, stream_length=bad.size)
    d_good = torch.from_numpy(np.concatenate([s_mt, np.zeros((-s_mt.size) % 16, np.uint8)])).cuda()
    gpu_ctx.decode_device(base, d_good, out, stream_length=s_mt.size)
    assert gpu_ctx.status(base) == 0 and np.array_equal(out.cpu().numpy(), nonstat[:3_000_000])  # the status word was cleared


@pytest.mark.parametrize("states", (32, 64))
def test_index_build_for_block_streams(gpu_ctx, oracle, ref, nonstat, zipf, states):
    """block_ streams are one chain with inline headers: the single wavefront that walks them also reports the headers it meets,
    so one pass turns a stream (ours or the real reference's) into a plan with a chain per block and per checkpoint."""
    for bits in (11, 13):
        for src, n, block in ((zipf, 300_000, 65536), (nonstat, 2_000_000, 65536), (nonstat, 1_000_003, 32768)):
            d = src[:n]
            s, plan_enc = H.encode(H.BLOCK, states, bits, d, index_interval=32, block_size=block)
            plan = gpu_ctx.index_build(H.BLOCK, states, bits, s, 32)
            assert np.array_equal(plan, plan_enc), (states, bits, n, block)
            r, got = gpu_ctx.decode_host(H.BLOCK, states, bits, s, n, plan=plan)
            assert r == n and np.array_equal(got, d)
        for src, n in ((nonstat, 3_000_000), (zipf, 1 << 20), (zipf, 524_300)):  # the real reference's adaptive blocks (incl. a quirk length)
            d = src[:n]
            s = ref.encode(BLOCK, states, bits, d)
            r0, want = oracle.decode(BLOCK, states, bits, s, n)
            plan = gpu_ctx.index_build(H.BLOCK, states, bits, s, 64)
            assert H.plan_chain_count(plan) > 1
            r, got = gpu_ctx.decode_host(H.BLOCK, states, bits, s, n, plan=plan)
            assert r == r0 and np.array_equal(got, want), (states, bits, n)


def test_corrupted_streams_fail_cleanly(gpu_ctx, zipf):
    """Memory safety: random corruption of headers / histograms / words must end in `return 0` or in (wrong) bytes of the
    right length — never in a fault or a hang (every index in the kernel is masked, every loop bounded by the plan)."""
    rng = np.random.default_rng(12345)
    d = zipf[:200_000]
    for container in (H.RAW, H.BLOCK, H.MT):
        for states in (32, 64):
            s = H.encode(container, states, 11, d)
            for trial in range(12):
                bad = s.copy()
                hot = 16 + 4 * states + 1024  # headers, counts, states live here
                for _ in range(1 + trial % 4):
                    pos = int(rng.integers(16, hot)) if trial % 2 == 0 else int(rng.integers(16, bad.size))
                    bad[pos] ^= 1 << int(rng.integers(0, 8))
                r, got = gpu_ctx.decode_host(container, states, 11, bad, d.size)
                assert r in (0, d.size)
            # truncated input
            for cut in (17, 600, s.size // 2, s.size - 2):
                r, _ = gpu_ctx.decode_host(container, states, 11, s[:cut], d.size)
                assert r == 0
    # the context is still healthy afterwards
    s = H.encode(H.RAW, 64, 11, d)
    r, got = gpu_ctx.decode_host(H.RAW, 64, 11, s, d.size)
    assert r == d.size and np.array_equal(got, d)


# ---- GPU encoder: byte-identical to the host encoder with the same block layout ----------------
def _gpu_encode(ctx, states, bits, data, block):
    import torch
    d_in = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    cap = H.capacity(H.MT, states, data.size)
    d_out = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block)
    return d_out[:n].cpu().numpy(), d_out


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_gpu_encoder_matches_host_encoder(gpu_ctx, oracle, zipf, nonstat, states, bits):
    runs = np.concatenate([np.full(70_000, 7, np.uint8), zipf[:100_000], np.full(200_000, 200, np.uint8), zipf[:33]])
    for src, n, block in ((zipf, 1, 64), (zipf, 63, 64), (zipf, 64, 64), (zipf, 65, 64), (zipf, 100_000, 4096), (zipf, 4096 + 17, 4096), (zipf, 65536, 65536),
                          (zipf, 65536 + 31, 65536), (zipf, 65536 + 64, 65536), (zipf, 300_001, 32768), (zipf, 8192 + 5, 8192), (zipf, 12288 + 64 + 3, 65536), (zipf, 3 * 4096 + 130, 1 << 20), (nonstat, 3_000_000, 65536),
                          (nonstat, 1_000_003, 1 << 18), (runs, runs.size, 4096), (runs, runs.size, 65536), (nonstat, 3_000_000, 1 << 21), (zipf, 500_000, 192 * 64)):
        d = src[:n]
        want = H.encode(H.MT, states, bits, d, block_size=block, independent_blocks=True)
        got, _ = _gpu_encode(gpu_ctx, states, bits, d, block)
        assert got.size == want.size, (states, bits, n, block, got.size, want.size)
        assert np.array_equal(got, want), (states, bits, n, block, int(np.argmax(got != want)))
        if n >= states - 1:  # shorter inputs are undefined behaviour in the reference decoders
            r, back = oracle.decode(MT, states, bits, got, n)
            assert r == n and np.array_equal(back, d), (states, bits, n, block)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("interval", (0, 4, 32, 256))
def test_gpu_encoder_emits_the_host_encoders_plan(gpu_ctx, zipf, nonstat, states, interval):
    """Stream AND sidecar plan from the device == hsrans_encode_ex on the host, byte for byte; decode with the device plan."""
    import torch
    runs = np.concatenate([zipf[:70_000], np.full(140_000, 9, np.uint8), zipf[:50_001]])
    for src, n, block, bits in ((zipf, 64, 64, 11), (zipf, 65536, 65536, 11), (zipf, 65536 + 63, 65536, 12), (zipf, 300_001, 32768, 11), (zipf, 8192 * 3 + 4 * 64, 8192, 14),
                                (nonstat, 2_000_000, 1 << 18, 11), (runs, runs.size, 65536, 15), (zipf, 1 << 20, 1 << 20, 10)):
        d = src[:n]
        want = H.encode(H.MT, states, bits, d, block_size=block, index_interval=interval, independent_blocks=True) if interval else \
            (H.encode(H.MT, states, bits, d, block_size=block, independent_blocks=True), None)
        want_stream, want_plan = want
        d_in = torch.from_numpy(np.ascontiguousarray(d)).cuda()
        d_out = torch.full((H.capacity(H.MT, states, n),), 0x5A, dtype=torch.uint8, device="cuda")
        m, dplan = gpu_ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block, index_interval=interval, want_plan=True)
        assert m == want_stream.size and np.array_equal(d_out[:m].cpu().numpy(), want_stream), (states, interval, n, block)
        got_plan = gpu_ctx.read_device_plan(dplan)
        if want_plan is None:
            want_plan = H.plan_build(H.MT, states, bits, want_stream)
        assert got_plan.size == want_plan.size and np.array_equal(got_plan, want_plan), (states, interval, n, block, int(np.argmax(got_plan[:want_plan.size] != want_plan)))
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_out, back, stream_length=m)
        assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d_in), (states, interval, n, block)


def test_gpu_encode_decode_never_leaves_hbm_1gib(gpu_ctx):
    """BASELINE config 4 shape: 2^30 bytes, 256 KiB blocks, checkpoints every 32 groups; encoder and decoder on the device."""
    import torch
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(7)
    # Zipf-ish bytes made on the device (the host generator would dominate the test's run time)
    u = torch.rand(n, device="cuda", generator=g)
    d_in = (u.pow_(6).mul_(205)).to(torch.uint8)
    del u
    d_out = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    m, dplan = gpu_ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=1 << 18, index_interval=32, want_plan=True)
    assert 0.3 * n < m < 0.9 * n
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.decode_device(dplan, d_out, back, stream_length=m)
    assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d_in)
    # the stream alone (no sidecar) decodes too: device walk of the block headers
    dplan2 = gpu_ctx.make_device_plan_from_stream(H.MT, 64, 11, d_out, m, n)
    back.zero_()
    gpu_ctx.decode_device(dplan2, d_out, back, stream_length=m)
    assert gpu_ctx.status(dplan2) == 0 and torch.equal(back, d_in)


def test_gpu_encoder_round_trip_on_device_100mb(gpu_ctx):
    import torch
    n = 100_000_000
    d = synth.enwik8_shaped(n, seed=5)
    stream, d_out = _gpu_encode(gpu_ctx, 64, 11, d, 1 << 16)
    assert 0.5 * n < stream.size < 0.8 * n
    dplan = gpu_ctx.make_device_plan_from_stream(H.MT, 64, 11, d_out, stream.size, n)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.decode_device(dplan, d_out, back, stream_length=stream.size)
    assert gpu_ctx.status(dplan) == 0
    assert torch.equal(back.cpu(), torch.from_numpy(d))


def _extractions_needed(block, bits):
    """how many entries of the reference's heap sort the normalisation of `block` needs (the GPU encoder replays exactly that many
    extractions: DESIGN 5b) — the host-side restatement of adjust_counts' arithmetic, used to pick inputs, not to check results"""
    target = 1 << bits
    raw = np.bincount(block, minlength=256).astype(np.int64)
    factor = np.float32(target) / np.float32(block.size)
    c = ((raw.astype(np.float32) * factor).astype(np.float32) + np.float32(0.5)).astype(np.uint16).astype(np.int64)
    c[(c == 0) & (raw != 0)] = 1
    s = int(c.sum())
    if s == target:
        return 0
    m = int((c >= 2).sum())
    if s > target:
        e, j, mj = s - target, 1, m
        while e > mj:
            e -= mj
            j += 1
            mj = int((c >= j + 1).sum())
        return mj - e
    e = target - s
    return e - (e - 1) // m * m


@pytest.mark.parametrize("states", (32, 64))
def test_gpu_encoder_every_depth_of_the_heap_sort_replay(gpu_ctx, oracle, states):
    """The GPU encoder replays the reference's heap sort (hist.cpp:16-215) only as far as the normalisation needs it, with one piece of
    straight-line code while the heap has >= 127 entries, another down to 63 and a third below: blocks that need few, ~130,
    ~160 and > 192 extractions, byte-identical to the host encoder (whose normalisation is pinned to the reference's make_hist)."""
    seen = set()
    for name, d, bits, block in (("text", synth.enwik8_shaped(1 << 18, seed=5), 11, 1 << 16), ("text15", synth.enwik8_shaped(1 << 18, seed=6), 15, 1 << 16),
                                 ("uniform", synth.uniform_bytes(1 << 18, seed=7), 15, 1 << 16), ("uniform12", synth.uniform_bytes(1 << 17, seed=8), 12, 1 << 15),
                                 ("zipf", synth.zipf_bytes(1 << 18, 1.05, seed=10), 14, 1 << 15)):
        for b in range(0, d.size, block):
            t = _extractions_needed(d[b:b + block], bits)
            seen |= {"none"} if t == 0 else {"high"} | ({"mid"} if t > 129 else set()) | ({"general"} if t > 192 else set())  # (the first 129 extractions of any block are the high form's)
        want = H.encode(H.MT, states, bits, d, block_size=block, independent_blocks=True)
        got, _ = _gpu_encode(gpu_ctx, states, bits, d, block)
        assert got.size == want.size and np.array_equal(got, want), (name, states, bits, block, int(np.argmax(got[:want.size] != want[:got.size])))
        r, back = oracle.decode(MT, states, bits, got, d.size)
        assert r == d.size and np.array_equal(back, d), (name, states, bits)
    assert seen == {"none", "high", "mid", "general"}, seen  # (the inputs reach every form of the extraction)


def test_gpu_encoder_placement_by_either_kernel(gpu_ctx, zipf):
    """Up to 4,096 blocks every workgroup of the gather kernel adds up the image sizes in front of its block itself; beyond, the scan
    kernel does: the same stream as the host encoder's on both sides of the limit; a too-small output buffer is refused and left untouched."""
    import torch
    src = synth.zipf_bytes(5000 * 4096 + 17, 1.1, seed=3)
    for n, block in ((4096 * 4096, 4096), (4097 * 4096, 4096), (5000 * 4096 + 17, 4096), (300_000, 4096)):
        d = src[:n]
        want = H.encode(H.MT, 64, 11, d, block_size=block, independent_blocks=True)
        got, _ = _gpu_encode(gpu_ctx, 64, 11, d, block)
        assert got.size == want.size and np.array_equal(got, want), (n, block)
        d_in = torch.from_numpy(d.copy()).cuda()
        small = torch.full((want.size - 2,), 0xA5, dtype=torch.uint8, device="cuda")
        with pytest.raises(H.HsransError):  # (the capacity contract of the host encoders: refused before anything is launched)
            gpu_ctx.encode_device(H.MT, 64, 11, d_in, small, block_size=block)
        assert bool((small == 0xA5).all()), (n, block)


def test_gpu_encoder_rejects_bad_arguments(gpu_ctx, zipf):
    import torch
    d_in = torch.from_numpy(zipf[:4096].copy()).cuda()
    d_out = torch.empty(H.capacity(H.MT, 64, 4096), dtype=torch.uint8, device="cuda")
    for kw in (dict(container=H.BLOCK), dict(block_size=100), dict(block_size=0), dict(bits=9), dict(states=16)):
        a = dict(container=H.MT, states=64, bits=11, block_size=1024)
        a.update(kw)
        with pytest.raises(H.HsransError):
            gpu_ctx.encode_device(a["container"], a["states"], a["bits"], d_in, d_out, block_size=a["block_size"])
    with pytest.raises(H.HsransError):
        gpu_ctx.encode_device(H.MT, 64, 11, d_in, d_out[:1000], block_size=1024)  # capacity contract


@pytest.mark.parametrize("container", ("mt", "raw"))
def test_plain_host_decode_keeps_the_index_of_its_first_call(gpu_ctx, ref, nonstat, container):
    """hsrans_decode_host without a plan (the reference's decodeFunc shape): the first call on an mt_/raw stream records an index, the
    next call on the same bytes uses it, and other bytes at the same address — even one flipped bit — are noticed (device-side
    fingerprint of the whole stream) and decoded from scratch."""
    C, RC = (H.MT, MT) if container == "mt" else (H.RAW, RAW)
    n = 3_000_000
    d = nonstat[:n]
    s = np.ascontiguousarray(ref.encode(RC, 64, 11, d))
    ctx = H.Context(0)  # (its own context: the cache belongs to it)
    assert ctx.host_index_chains() == 0
    r, got = ctx.decode_host(C, 64, 11, s, n)
    assert r == n and np.array_equal(got, d)
    chains = ctx.host_index_chains()
    assert chains > 100
    for _ in range(3):
        r, got = ctx.decode_host(C, 64, 11, s, n)
        assert r == n and np.array_equal(got, d) and ctx.host_index_chains() == chains
    # one bit of one word flipped IN PLACE: a decoder that trusted the address would emit the old states' bytes
    s[s.size // 2] ^= 0x10
    want_r, want = hsrans_cpu_decode(C, 64, 11, s, n)
    r, got = ctx.decode_host(C, 64, 11, s, n)
    assert r == want_r and np.array_equal(got[:r], want[:r]) and not np.array_equal(got, d)
    s[s.size // 2] ^= 0x10
    r, got = ctx.decode_host(C, 64, 11, s, n)
    assert r == n and np.array_equal(got, d)
    # another stream, another codec, short streams: never served from the cache
    d2 = np.ascontiguousarray(nonstat[:n][::-1])
    s2 = np.ascontiguousarray(ref.encode(RC, 32, 12, d2))
    r, got = ctx.decode_host(C, 32, 12, s2, n)
    assert r == n and np.array_equal(got, d2)
    small = np.ascontiguousarray(ref.encode(RC, 64, 11, d[:300_000]))
    r, got = ctx.decode_host(C, 64, 11, small, 300_000)
    assert r == 300_000 and np.array_equal(got, d[:300_000])
    r, got = ctx.decode_host(C, 64, 11, s, n)
    assert r == n and np.array_equal(got, d)


@pytest.mark.parametrize("states,bits", ((64, 11), (32, 12), (64, 14)))
def test_plain_host_decode_of_a_raw_stream_without_a_host_core(gpu_ctx, oracle, zipf, monkeypatch, states, bits):
    """A raw stream of more than 1 MiB through the plain decodeFunc entry with NO host-side decode pass at all:
      * HSRANS_HOST_INDEX_CACHE_OFF=1: every call decodes the stream as what it is — one dependent chain, one wavefront (k_decode_single
        for the 8-byte-table widths: above 1 MiB the default path never reaches that kernel, because the first call leaves an index);
      * HSRANS_HIP_STRICT=1: the first call's checkpoints are recorded by the decoding wavefront itself instead of by the host SIMD
        decoder's pass; the second call launches the index it left.
    Both against the oracle (rANS32x64_16w.cpp:168-283 / rANS32x32_16w.cpp:161-269)."""
    n = (1 << 21) + 12_345
    d = np.resize(zipf, n)
    s = H.encode(H.RAW, states, bits, d)
    r0, want = oracle.decode(RAW, states, bits, s, n)
    assert r0 == n and np.array_equal(want, d)
    monkeypatch.setenv("HSRANS_HOST_INDEX_CACHE_OFF", "1")
    ctx = H.Context(0)
    for _ in range(2):
        r, got = ctx.decode_host(H.RAW, states, bits, s, n)
        assert r == n and np.array_equal(got, want) and ctx.host_index_chains() == 0
    monkeypatch.delenv("HSRANS_HOST_INDEX_CACHE_OFF")
    monkeypatch.setenv("HSRANS_HIP_STRICT", "1")
    ctx = H.Context(0)
    r, got = ctx.decode_host(H.RAW, states, bits, s, n)
    assert r == n and np.array_equal(got, want)
    chains = ctx.host_index_chains()
    assert chains > 1  # the recording wavefront left its checkpoints
    r, got = ctx.decode_host(H.RAW, states, bits, s, n)
    assert r == n and np.array_equal(got, want) and ctx.host_index_chains() == chains


def hsrans_cpu_decode(container, states, bits, stream, n):
    """the library's host decoder (hsrans_decode_cpu, scalar route): a second implementation for streams no encoder wrote"""
    out = np.full(n, 0xCC, np.uint8)
    L = H.load_library()
    r = L.hsrans_decode_cpu(0, 1, container, states, bits, stream.ctypes.data, stream.size, out.ctypes.data, n, None, 0)
    return r, out


# ---- the raw format's encoder on the device (raw half: src/rANS32x64_16w.cpp:34-166) ------------------------
def _gpu_encode_raw(ctx, states, bits, data, **kw):
    import torch
    d_in = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    d_out = torch.full((kw.pop("capacity", None) or H.capacity(H.RAW, states, data.size),), 0xA5, dtype=torch.uint8, device="cuda")
    r = ctx.encode_device_raw(states, bits, d_in, d_out, **kw)
    n = r if isinstance(r, int) else r[0]
    return (d_out[:n].cpu().numpy(), d_out, d_in) + (() if isinstance(r, int) else tuple(r[1:]))


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (10, 11, 12, 13, 14, 15))
def test_gpu_raw_encoder_matches_host_encoder_and_reference(gpu_ctx, oracle, ref, zipf, nonstat, states, bits):
    """the stream one wavefront writes == hsrans_encode (host) == the real reference's rANS32x{32,64}_16w_encode_scalar_N, with the
    histogram made on the device (make_hist's result) and with the caller's; the oracle decodes it"""
    runs = np.concatenate([np.full(70_000, 7, np.uint8), zipf[:100_000], np.full(200_000, 200, np.uint8), zipf[:33]])
    one = np.full(50_000, 65, np.uint8)
    for src, n in ((zipf, 1), (zipf, 31), (zipf, 63), (zipf, 64), (zipf, 65), (zipf, 255), (zipf, 256), (zipf, 4096 + 17), (zipf, 8192), (zipf, 12288 + 64 + 3), (zipf, 100_000),
                   (zipf, 300_001), (nonstat, 1_000_003), (runs, runs.size), (one, one.size)):
        d = src[:n]
        want = H.encode(H.RAW, states, bits, d)
        got = _gpu_encode_raw(gpu_ctx, states, bits, d)[0]
        assert got.size == want.size and np.array_equal(got, want), (states, bits, n, got.size, want.size)
        assert np.array_equal(got, ref.encode(RAW, states, bits, d)), (states, bits, n)
        hist = H.make_hist(zipf[:500_000] if src is zipf else d, bits)  # zipf: a histogram that is NOT the data's own
        if src is zipf and n < 256:
            hist = H.make_hist(d, bits)
        want_h = H.encode(H.RAW, states, bits, d, hist=hist, out_capacity=2 * n + 4096)
        got_h = _gpu_encode_raw(gpu_ctx, states, bits, d, hist=hist, capacity=2 * n + 4096)[0]
        assert got_h.size == want_h.size and np.array_equal(got_h, want_h), (states, bits, n, "given histogram")
        if n >= states - 1:  # shorter inputs are undefined behaviour in the reference decoders
            r, back = oracle.decode(RAW, states, bits, got, n)
            assert r == n and np.array_equal(back, d), (states, bits, n)


@pytest.mark.parametrize("states", (32, 64))
def test_gpu_raw_encoder_emits_the_host_encoders_index(gpu_ctx, zipf, nonstat, states):
    """uniform intervals and listed checkpoints (incl. the one-chain-per-wavefront boundaries): stream and plan == hsrans_encode_ex's;
    the device plan it returns decodes the stream it wrote, nothing having left HBM"""
    import torch
    for src, n, bits, kw in ((zipf, 300_001, 11, dict(index_interval=4)), (zipf, 300_001, 14, dict(index_interval=32)), (zipf, 64 * 8, 11, dict(index_interval=4)),
                             (zipf, 64 * 8 + 5, 11, dict(index_interval=8)), (zipf, 100, 11, dict(index_interval=4)), (nonstat, 3_000_000, 11, dict(index_interval=256)),
                             (nonstat, 3_000_000, 11, dict(index_groups="wave")), (nonstat, 3_000_000, 15, dict(index_groups="wave")),
                             (zipf, 300_001, 12, dict(index_groups=[4, 8, 400, 404, 4000, 1 << 20])), (zipf, 64 * 100, 11, dict(index_groups=[96, 100, 104]))):
        d = src[:n]
        kw = dict(kw)
        if isinstance(kw.get("index_groups"), str):
            kw["index_groups"] = H.index_boundaries(states, bits, n, gpu_ctx)
            assert len(kw["index_groups"]) > 100
        want_stream, want_plan = H.encode(H.RAW, states, bits, d, **kw)
        got, d_out, d_in, plan, dplan = _gpu_encode_raw(gpu_ctx, states, bits, d, want_plan=True, want_device_plan=True, **kw)
        assert np.array_equal(got, want_stream), (states, n, bits)
        assert plan.size == want_plan.size and np.array_equal(plan, want_plan), (states, n, bits, kw.keys())
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_out, back, stream_length=got.size)
        assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d_in), (states, n, bits)
    # hsrans_encode_device(HSRANS_RAW) is the same call
    d = zipf[:200_000]
    d_in = torch.from_numpy(d.copy()).cuda()
    d_out = torch.empty(H.capacity(H.RAW, states, d.size), dtype=torch.uint8, device="cuda")
    m, dplan = gpu_ctx.encode_device(H.RAW, states, 11, d_in, d_out, index_interval=32, want_plan=True)
    want_stream, want_plan = H.encode(H.RAW, states, 11, d, index_interval=32)
    assert m == want_stream.size and np.array_equal(d_out[:m].cpu().numpy(), want_stream) and np.array_equal(gpu_ctx.read_device_plan(dplan), want_plan)


def test_gpu_raw_encode_decode_never_leaves_hbm_1gib(gpu_ctx):
    """BASELINE config 2's shape at 2^30 bytes: raw stream AND its one-chain-per-wavefront index written by the one coding wavefront
    (byte offsets inside the 2 GiB scratch slot are 32-bit: this is the size that exercises them), decoded by the headline launch;
    input, stream, index and output stay on the device.  The stream's header fields and the plan's chain count are checked too."""
    import torch
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(11)
    u = torch.rand(n, device="cuda", generator=g)
    d_in = (u.pow_(6).mul_(205)).to(torch.uint8)
    del u
    d_out = torch.empty(H.capacity(H.RAW, 64, n), dtype=torch.uint8, device="cuda")
    groups = H.index_boundaries(64, 11, n, gpu_ctx)
    m, dplan = gpu_ctx.encode_device_raw(64, 11, d_in, d_out, index_groups=groups, want_device_plan=True)
    assert 0.3 * n < m < 0.9 * n
    head = d_out[:16].cpu().numpy().view(np.uint64)
    assert int(head[0]) == n and int(head[1]) == m
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.decode_device(dplan, d_out, back, stream_length=m)
    assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d_in)
    assert dplan.launch_info()["chains"] == len(groups) + 1


def test_gpu_raw_encoder_rejects_bad_arguments(gpu_ctx, zipf):
    import torch
    d = zipf[:4096].copy()
    d_in = torch.from_numpy(d).cuda()
    d_out = torch.empty(H.capacity(H.RAW, 64, 4096) + 16, dtype=torch.uint8, device="cuda")
    bad_hist = H.make_hist(d, 11)
    bad_hist.symbolCount[int(d[0])] += 1  # no longer sums to 2^11
    for kw in (dict(bits=9), dict(states=16), dict(hist=bad_hist), dict(index_interval=6, want_plan=True), dict(index_groups=[6], want_plan=True),
               dict(index_groups=[8, 8], want_plan=True), dict(index_groups=[0], want_plan=True)):
        a = dict(states=64, bits=11)
        a.update(kw)
        with pytest.raises(H.HsransError):
            gpu_ctx.encode_device_raw(a.pop("states"), a.pop("bits"), d_in, d_out, **a)
    with pytest.raises(H.HsransError):
        gpu_ctx.encode_device_raw(64, 11, d_in, d_out[:1000])  # capacity contract
    with pytest.raises(H.HsransError):
        gpu_ctx.encode_device_raw(64, 11, d_in, d_out[2:])  # alignment
    # a histogram without a slot for a symbol that occurs: not encodable (the host encoder returns 0 as well)
    other = H.make_hist(np.full(4096, d[0], np.uint8), 11)
    with pytest.raises(H.HsransError):
        H.encode(H.RAW, 64, 11, d, hist=other, out_capacity=3 * d.size)
    big = torch.empty(3 * d.size, dtype=torch.uint8, device="cuda")
    with pytest.raises(H.HsransError):
        gpu_ctx.encode_device_raw(64, 11, d_in, big, hist=other)
    assert gpu_ctx.encode_device_raw(64, 11, d_in, d_out) == H.encode(H.RAW, 64, 11, d).size  # (the context still works)


# ---- seeded random sweep over the whole configuration space ---------------------------------------------------------------
def _random_case(rng, zipf, nonstat):
    kind = rng.integers(0, 4)
    n = int(np.exp(rng.uniform(np.log(63), np.log(400_000))))
    if kind == 0:
        off = int(rng.integers(0, zipf.size - n))
        d = zipf[off:off + n]
    elif kind == 1:
        off = int(rng.integers(0, nonstat.size - n))
        d = nonstat[off:off + n]
    elif kind == 2:
        d = synth.uniform_bytes(n, seed=int(rng.integers(1, 1 << 30)))
    else:
        d = synth.two_symbol(n, seed=int(rng.integers(1, 1 << 30)))
    return np.ascontiguousarray(d)


def test_random_sweep_decode(gpu_ctx, oracle, zipf, nonstat):
    import os
    rng = np.random.default_rng(int(os.environ.get("HSRANS_SWEEP_SEED", "20241008")))
    for case in range(int(os.environ.get("HSRANS_SWEEP_CASES", "160"))):  # a one-off soak run scales these up
        container = int(rng.integers(0, 3))
        states = int(rng.choice((32, 64)))
        bits = int(rng.integers(10, 16))
        d = _random_case(rng, zipf, nonstat)
        n = d.size
        interval = int(rng.choice((0, 4, 8, 32, 100, 1024)))
        block = int(rng.choice((0, 32768, 65536))) if container != RAW else 0  # capacity() is sized for blocks >= 32 KiB
        tag = (case, container, states, bits, n, interval, block)
        if interval:
            s, plan = H.encode(container, states, bits, d, index_interval=interval, block_size=block)
        else:
            s, plan = (H.encode(container, states, bits, d, block_size=block) if block else H.encode(container, states, bits, d)), None
        r0, want = oracle.decode(container, states, bits, s, n)
        if r0 == 0:  # the reference cannot decode its own one-symbol block_/mt_ files (tests/test_oracle_vs_ref.py): neither do we
            assert container != RAW and np.unique(d).size == 1, tag
            assert gpu_ctx.decode_host(container, states, bits, s, n)[0] == 0, tag
            continue
        assert r0 == n and np.array_equal(want, d), tag
        r, got = gpu_ctx.decode_host(container, states, bits, s, n, plan=plan)
        assert r == n and np.array_equal(got, want), tag + (int(np.argmax(got != want)),)
        if container != BLOCK and n >= 4096 and case % 4 == 0:  # checkpoints recovered from the stream alone by a GPU pass
            plan2 = gpu_ctx.index_build(container, states, bits, s, 8)
            r, got = gpu_ctx.decode_host(container, states, bits, s, n, plan=plan2)
            assert r == n and np.array_equal(got, want), tag


def test_random_sweep_gpu_encoder(gpu_ctx, oracle, zipf, nonstat):
    import torch
    import os
    rng = np.random.default_rng(int(os.environ.get("HSRANS_SWEEP_SEED", "7")))
    for case in range(int(os.environ.get("HSRANS_SWEEP_CASES", "160")) * 3 // 8):
        states = int(rng.choice((32, 64)))
        bits = int(rng.integers(10, 16))
        d = _random_case(rng, zipf, nonstat)
        n = d.size
        block = int(rng.choice((32768, 65536, 1 << 17)))  # capacity() is sized for blocks >= 32 KiB
        interval = int(rng.choice((0, 4, 16, 64)))
        tag = (case, states, bits, n, block, interval)
        d_in = torch.from_numpy(d).cuda()
        d_out = torch.empty(H.capacity(H.MT, states, n), dtype=torch.uint8, device="cuda")
        m, dplan = gpu_ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block, index_interval=interval, want_plan=True)
        stream = d_out[:m].cpu().numpy()
        want = H.encode(H.MT, states, bits, d, block_size=block, independent_blocks=True)
        assert np.array_equal(stream, want), tag
        if n >= states - 1 and np.unique(d).size > 1:  # (one-symbol files: the reference rejects its own streams)
            r, back = oracle.decode(MT, states, bits, stream, n)
            assert r == n and np.array_equal(back, d), tag
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_out, back, stream_length=m)
        assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d_in), tag


def test_random_sweep_gpu_raw_encoder(gpu_ctx, oracle, zipf, nonstat):
    """seeded random cases through hsrans_encode_device_raw: stream (and plan, when an index is asked for) byte-identical to
    hsrans_encode_ex for random lengths, widths, state counts, own / given histograms, uniform and listed checkpoints"""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("HSRANS_SWEEP_SEED", "11")))
    for case in range(int(os.environ.get("HSRANS_SWEEP_CASES", "160")) * 3 // 8):
        states = int(rng.choice((32, 64)))
        bits = int(rng.integers(10, 16))
        d = _random_case(rng, zipf, nonstat)
        n = d.size
        kind = int(rng.integers(0, 3))
        kw = {}
        if kind == 1:
            kw["index_interval"] = int(rng.choice((4, 8, 32, 100, 1024)))
        elif kind == 2:
            whole = n // states
            k = int(rng.integers(1, 40))
            kw["index_groups"] = np.unique((rng.integers(1, max(2, whole // 4 + 2), size=k) * 4).astype(np.uint64))  # some behind the last whole group
        hist = H.make_hist(d, bits) if rng.integers(0, 2) else None
        tag = (case, states, bits, n, kind, hist is not None)
        want = H.encode(H.RAW, states, bits, d, hist=hist, **kw)
        got = _gpu_encode_raw(gpu_ctx, states, bits, d, hist=hist, want_plan=bool(kw), want_device_plan=bool(kw), **kw)
        if kw:
            want_stream, want_plan = want
            assert np.array_equal(got[0], want_stream) and np.array_equal(got[3], want_plan), tag
            back = torch.zeros(n, dtype=torch.uint8, device="cuda")
            gpu_ctx.decode_device(got[4], got[1], back, stream_length=got[0].size)
            assert gpu_ctx.status(got[4]) == 0 and torch.equal(back, got[2]), tag
        else:
            assert np.array_equal(got[0], want), tag
        if n >= states - 1:
            r, back = oracle.decode(RAW, states, bits, got[0], n)
            assert r == n and np.array_equal(back, d), tag


def test_random_sweep_few_large_blocks(gpu_ctx, oracle, zipf, nonstat):
    """seeded random mt_ / block_ streams in FEW large blocks with many checkpoints (the shapes k_decode_spread takes, and their
    neighbours that fall back to the grouped launch): decoded bytes against the oracle, several launches of one plan"""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("HSRANS_SWEEP_SEED", "13")))
    base = np.concatenate([nonstat, zipf[:2_000_000], np.full(700_000, 3, np.uint8), nonstat[::-1], zipf[2_000_000:4_000_000]])
    took = 0
    for case in range(int(os.environ.get("HSRANS_SWEEP_CASES", "160")) // 8):
        container = int(rng.choice((MT, BLOCK)))
        bits = int(rng.choice((10, 11, 11, 12, 13)))
        n = int(rng.integers(5_000_000, base.size))
        off = int(rng.integers(0, base.size - n + 1))
        d = np.ascontiguousarray(base[off:off + n])
        block = int(rng.choice((1 << 17, 1 << 18, 3 << 17, 1 << 19, 1 << 20)))
        interval = int(rng.choice((4, 8, 8, 16)))
        tag = (case, container, bits, n, block, interval)
        s, plan = H.encode(container, 64, bits, d, block_size=block, index_interval=interval)
        r0, want = oracle.decode(container, 64, bits, s, n)
        assert r0 == n and np.array_equal(want, d), tag
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        d_want = torch.from_numpy(d).cuda()
        dplan = gpu_ctx.make_device_plan(plan)
        for launch in range(2):
            d_out.zero_()
            gpu_ctx.decode_device(dplan, d_in, d_out, stream_length=s.size)
            assert gpu_ctx.status(dplan) == 0 and torch.equal(d_out, d_want), tag + (launch,)
        took += dplan.launch_info()["spread"]
    assert took >= 4  # (the sweep does reach the launch it is about)


@pytest.mark.parametrize("states", (32, 64))
@pytest.mark.parametrize("bits", (14, 15))
def test_wide_histogram_rank_table_on_awkward_histograms(gpu_ctx, oracle, bits, states):
    """14 / 15-bit persistent launches decode through the rank table (a byte per slot = the symbol's rank by frequency, then 256
    entries ordered by rank): all 256 symbols present with equal counts (ties in the ranking), two symbols (254 zero counts),
    non-stationary and text-shaped data; uniform-interval plans (the generic loop) and one chain per wave (the two-chain kernel's
    hand-scheduled loop, with chains of different lengths behind it); 32 states: the pair loops."""
    for name, d in (("uniform", synth.uniform_bytes(1_000_003, seed=4)), ("two", synth.two_symbol(500_000, seed=2)),
                    ("nonstat", synth.nonstationary(2_000_000)), ("zipf", synth.enwik8_shaped(300_000, seed=1))):
        plans = [H.encode(H.RAW, states, bits, d, index_interval=interval) for interval in (4, 32)]
        plans.append(H.encode(H.RAW, states, bits, d, index_groups=H.index_boundaries(states, bits, d.size, gpu_ctx)))
        for k, (s, plan) in enumerate(plans):
            r0, want = oracle.decode(RAW, states, bits, s, d.size)
            r, got = gpu_ctx.decode_host(H.RAW, states, bits, s, d.size, plan=plan)
            assert r == r0 == d.size and np.array_equal(got, want), (name, k)


def test_private_pair_mode_in_a_subprocess(tmp_path):
    """32-state mt_ plans without a sidecar: two blocks per wavefront, one per wave half, each with its own table
    (run_private_pair).  By default only used when there are more blocks than wave slots, so it is forced here
    (HSRANS_PRIVATE_PAIR=2 is read when the library initialises: separate process)."""
    import os
    import subprocess
    import sys
    script = tmp_path / "pair.py"
    script.write_text("""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import MT, Oracle
ctx = H.Context(0)
oracle = Oracle()
zipf = synth.enwik8_shaped(1 << 20, seed=11)
nonstat = synth.nonstationary(3_000_000)
for bits in (10, 11, 12, 13):
    for src, n, block in ((zipf, 65536, 0), (zipf, 300_001, 32768), (nonstat, 3_000_000, 0), (nonstat, 1_000_003, 65536), (zipf, 131072 + 65, 65536)):
        d = src[:n]
        s = H.encode(H.MT, 32, bits, d, block_size=block) if block else H.encode(H.MT, 32, bits, d)
        r0, want = oracle.decode(MT, 32, bits, s, n)
        r, got = ctx.decode_host(H.MT, 32, bits, s, n)
        assert r == r0 == n and np.array_equal(got, want), (bits, n, block)
print("pair ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, HSRANS_PRIVATE_PAIR="2")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "pair ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_golden_vectors_from_the_real_reference(gpu_ctx):
    """The committed fixtures (tests/golden: streams written and decoded by the REAL reference, incl. the quirk lengths whose
    reference round trip is wrong) through the HIP path: same return value, same bytes.  Needs neither the oracle nor
    the reference at run time.  The large manifest entries are regenerated by the product's own encoder (byte-identical to
    the reference's, checked by SHA-256) and decoded on the GPU."""
    import hashlib
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    small = np.load(os.path.join(here, "golden", "small_vectors.npz"))
    manifest = json.load(open(os.path.join(here, "golden", "manifest.json")))
    names = {"raw": H.RAW, "block": H.BLOCK, "mt": H.MT}
    keys = [k[:-7] for k in small.files if k.endswith("_stream") and not k.startswith("quirk_")]
    assert len(keys) >= 100
    for k in keys:
        cont, s, b, _tag = k.split("_", 3)
        stream, want = small[k + "_stream"], small[k + "_in"]
        if names[cont] == H.RAW and want.size < int(s[1:]) - 1:
            continue  # the reference itself walks off its buffers there
        r, got = gpu_ctx.decode_host(names[cont], int(s[1:]), int(b[1:]), stream, want.size)
        assert r == want.size and np.array_equal(got, want), k
    for q in manifest["quirks"]:
        k = q["key"]
        _, cont, s, b, n = k.split("_")
        stream, want = small[k + "_stream"], small[k + "_decoded"]
        r, got = gpu_ctx.decode_host(names[cont], int(s[1:]), int(b[1:]), stream, int(n[1:]))
        assert r == q["returned"], k
        if r:
            assert np.array_equal(got[:r], want[:r]), k
    inputs = {"zipf1M_seed1": synth.enwik8_shaped(1 << 20, seed=1), "uniform1M_seed1": synth.uniform_bytes(1 << 20, seed=1),
              "nonstat1M_seed99": synth.nonstationary(1 << 20, seed=99), "two3000": synth.two_symbol(3000, seed=5)}
    checked = 0
    for e in manifest["large"]:
        data = inputs[e["input"]]
        stream = H.encode(names[e["container"]], e["states"], e["bits"], data)
        assert stream.size == e["stream_len"] and hashlib.sha256(stream.tobytes()).hexdigest() == e["stream_sha256"], e
        r, got = gpu_ctx.decode_host(names[e["container"]], e["states"], e["bits"], stream, data.size)
        assert r == data.size and hashlib.sha256(got.tobytes()).hexdigest() == e["decoded_sha256"], e
        checked += 1
    assert checked >= 36


def test_offsets_beyond_4_gib(gpu_ctx):
    """Maximum sizes: 7 GiB of input, so that output offsets AND stream offsets cross 2^32 — encoded, planned (by the encoder and
    by the device-side header walk) and decoded without leaving HBM."""
    import torch
    n = (7 << 30) + 12345
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    step = 1 << 28
    for o in range(0, n, step):
        m = min(step, n - o)
        d[o:o + m] = torch.rand(m, device="cuda", generator=g).pow_(6).mul_(205).to(torch.uint8)
    out = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    m, dplan = gpu_ctx.encode_device(H.MT, 64, 11, d, out, block_size=1 << 18, index_interval=32, want_plan=True)
    assert m > 1 << 32
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.decode_device(dplan, out, back, stream_length=m)
    assert gpu_ctx.status(dplan) == 0 and torch.equal(back, d)
    dplan2 = gpu_ctx.make_device_plan_from_stream(H.MT, 64, 11, out, m, n)
    back.zero_()
    gpu_ctx.decode_device(dplan2, out, back, stream_length=m)
    assert gpu_ctx.status(dplan2) == 0 and torch.equal(back, d)
