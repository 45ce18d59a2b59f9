"""BASELINE.json configs 2-5 at their stated shape, on one GPU (the N>1 legs of configs 4/5 are `bench.py --workload sharded`
under the driver's multi-GPU run; their host logic runs under gloo in tests/test_sharded_gloo.py).

Full-size checks compare SHA-256 of the GPU output with the scalar CPU oracle's output for the same stream (not with the input:
the oracle is the contract), plus the size-independent property decode(encode(x)) == x."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import BLOCK, MT, RAW

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _tiled(n, seed=20241008):
    tile = 1 << 24
    base = synth.enwik8_shaped(min(tile, n), seed=seed)
    out = np.empty(n, np.uint8)
    for k, s in enumerate(range(0, n, tile)):
        c = min(tile, n - s)
        out[s:s + c] = (base if k % 7 == 0 else synth._permutation(1000 + k % 7)[base])[:c]
    return out


@pytest.fixture(scope="module")
def data100():
    return synth.enwik8_shaped(100_000_000, seed=20241008)


def _device_decode(ctx, stream, plan, n):
    import torch
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dplan = ctx.make_device_plan(plan)
    ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
    torch.cuda.synchronize()
    assert ctx.status(dplan) == 0
    return d_out.cpu().numpy(), dplan.launch_info()


@pytest.mark.parametrize("states,bits,index", ((64, 11, "wave"), (64, 11, 32), (32, 11, "wave"), (32, 10, 32), (64, 10, "wave"), (64, 13, "wave"), (32, 13, 32),
                                               (64, 12, "wave"), (64, 14, "wave"), (64, 15, "wave"), (32, 15, "wave")))
def test_config2_and_3_raw_100mb_against_the_oracle(gpu_ctx, oracle, data100, states, bits, index):
    """BASELINE configs 2/3: raw, 100 MB, 10..15 bits, both state counts, both index kinds; SHA-256 against the scalar oracle."""
    n = data100.size
    if index == "wave":
        stream, plan = H.encode(RAW, states, bits, data100, index_groups=H.index_boundaries(states, bits, n, gpu_ctx))
    else:
        stream, plan = H.encode(RAW, states, bits, data100, index_interval=index)
    r, want = oracle.decode(RAW, states, bits, stream, n)
    assert r == n
    got, info = _device_decode(gpu_ctx, stream, plan, n)
    assert _sha(got) == _sha(want) == _sha(data100), (states, bits, index, info)
    assert info["shared_table"] == 1 and info["grid"] >= 256


@pytest.mark.parametrize("container,states,bits,block,interval", ((MT, 64, 11, 0, 0), (MT, 64, 11, 1 << 18, 256), (BLOCK, 64, 11, 0, 32), (BLOCK, 32, 12, 1 << 16, 64),
                                                                  (MT, 32, 13, 0, 0)))
def test_containers_100mb_against_the_oracle(gpu_ctx, oracle, data100, container, states, bits, block, interval):
    n = data100.size
    if interval:
        stream, plan = H.encode(container, states, bits, data100, block_size=block, index_interval=interval)
    else:
        stream = H.encode(container, states, bits, data100, block_size=block) if block else H.encode(container, states, bits, data100)
        plan = H.plan_build(container, states, bits, stream)
    r, want = oracle.decode(container, states, bits, stream, n)
    assert r == n
    got, info = _device_decode(gpu_ctx, stream, plan, n)
    assert _sha(got) == _sha(want), (container, states, bits, block, interval, info)


def test_config4_block_1gib_in_256k_blocks_on_one_gpu(gpu_ctx, oracle):
    """BASELINE config 4 at its stated size: block_rANS32x64 16w 11-bit, 2^30 B in 256 KiB blocks + an index every 32 groups
    (what makes a block_ stream chunk-parallel), decoded by one launch; SHA-256 against the scalar oracle."""
    n = 1 << 30
    data = _tiled(n)
    stream, plan = H.encode(BLOCK, 64, 11, data, block_size=1 << 18, index_interval=32)
    assert H.plan_chain_count(plan) >= 4096 * 127
    r, want = oracle.decode(BLOCK, 64, 11, stream, n)
    assert r == n
    want_sha = _sha(want)
    assert want_sha == _sha(data)
    del want
    got, info = _device_decode(gpu_ctx, stream, plan, n)
    assert _sha(got) == want_sha, info


def test_dynamic_block_order_with_many_small_blocks(gpu_ctx, oracle):
    """More blocks than resident workgroups: the grouped launch hands the blocks behind the first round out through its ticket
    counter (k_decode_grouped, launch info `dynamic_groups`).  Non-stationary data so that single-symbol (fill) blocks are in the
    list; several launches of one plan back to back (the counter is never reset: ticket mod n_groups); the same through a device
    plan written by the GPU encoder; bit-exact against the oracle."""
    import torch
    n = 40 << 20
    data = synth.nonstationary(n, seed=77)
    stream, plan = H.encode(MT, 64, 11, data, block_size=1 << 14, index_interval=16)  # 2,560 blocks of 16 chains
    r, want = oracle.decode(MT, 64, 11, stream, n)
    assert r == n and np.array_equal(want, data)
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dplan = gpu_ctx.make_device_plan(plan)
    for launch in range(5):
        d_out.zero_()
        gpu_ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
        assert gpu_ctx.status(dplan) == 0
        assert torch.equal(d_out.cpu(), torch.from_numpy(data)), launch
    info = dplan.launch_info()
    assert info["dynamic_groups"] == 1 and info["waves_per_block"] == 8 and info["grid"] < 2560, info
    # the encoder's device-built plan (its group list is written on the device) takes the same path
    d_src = torch.from_numpy(data).cuda()
    d_enc = torch.empty(H.capacity(MT, 64, n), dtype=torch.uint8, device="cuda")
    m, dplan2 = gpu_ctx.encode_device(MT, 64, 11, d_src, d_enc, block_size=1 << 14, index_interval=16, want_plan=True)
    for launch in range(3):
        d_out.zero_()
        gpu_ctx.decode_device(dplan2, d_enc, d_out, stream_length=m)
        assert gpu_ctx.status(dplan2) == 0 and torch.equal(d_out, d_src), launch
    assert dplan2.launch_info()["dynamic_groups"] == 1


def test_few_large_blocks_are_dealt_out_evenly(gpu_ctx, oracle):
    """Fewer blocks than workgroup slots (BASELINE config 4's layout at 100 MB-class sizes): k_decode_spread deals ALL chains out
    evenly, a workgroup building the two tables its share touches (launch info `spread`).  Shares that start / end inside blocks,
    single-symbol (fill) blocks in the list, a short last block; plans from the host encoder, from the GPU encoder's plan kernel and
    from the device-side index assembly of a first decode; several launches; bit-exact against the oracle.  HSRANS_SPREAD=0 (the
    one-block-per-workgroup launch) decodes the same bytes."""
    import torch
    n = (24 << 20) + 70_001
    data = np.concatenate([synth.nonstationary(12 << 20, seed=79), np.full(1 << 19, 7, np.uint8), synth.enwik8_shaped((12 << 20) - (1 << 19) + 70_001, seed=3)])
    assert data.size == n
    for bits, block, interval in ((11, 1 << 18, 8), (10, 1 << 19, 16)):
        stream, plan = H.encode(MT, 64, bits, data, block_size=block, index_interval=interval, independent_blocks=True)
        r, want = oracle.decode(MT, 64, bits, stream, n)
        assert r == n and np.array_equal(want, data)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
        d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        d_src = torch.from_numpy(data).cuda()
        dplan = gpu_ctx.make_device_plan(plan)
        for launch in range(3):
            d_out.zero_()
            gpu_ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
            assert gpu_ctx.status(dplan) == 0 and torch.equal(d_out, d_src), (bits, launch)
        info = dplan.launch_info()  # (of the last launch)
        assert info["spread"] == 1 and info["waves_per_block"] == 16 and info["dynamic_groups"] == 0, info
        # the GPU encoder's device-built plan
        d_enc = torch.empty(H.capacity(MT, 64, n), dtype=torch.uint8, device="cuda")
        m, dplan2 = gpu_ctx.encode_device(MT, 64, bits, d_src, d_enc, block_size=block, index_interval=interval, want_plan=True)
        d_out.zero_()
        gpu_ctx.decode_device(dplan2, d_enc, d_out, stream_length=m)
        assert gpu_ctx.status(dplan2) == 0 and torch.equal(d_out, d_src)
        assert m == stream.size and dplan2.launch_info()["spread"] == 1
        # the index a first decode leaves behind (assembled on the device)
        base = gpu_ctx.make_device_plan_from_stream(MT, 64, bits, d_in, stream.size, n)
        d_out.zero_()
        dplan3 = gpu_ctx.decode_device_indexing(base, d_in, d_out, interval, stream_length=stream.size)
        assert torch.equal(d_out, d_src)
        d_out.zero_()
        gpu_ctx.decode_device(dplan3, d_in, d_out, stream_length=stream.size)
        assert gpu_ctx.status(dplan3) == 0 and torch.equal(d_out, d_src) and dplan3.launch_info()["spread"] == 1
    # blocks shorter than a workgroup's share, 32 states, wide histograms: the grouped launch as before
    for states, bits, block, interval in ((64, 11, 1 << 14, 8), (32, 11, 1 << 18, 8), (64, 13, 1 << 18, 8)):
        stream, plan = H.encode(MT, states, bits, data, block_size=block, index_interval=interval, independent_blocks=True)
        dplan = gpu_ctx.make_device_plan(plan)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
        d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
        assert gpu_ctx.status(dplan) == 0 and torch.equal(d_out, d_src) and dplan.launch_info()["spread"] == 0
    code = (
        "import numpy as np, torch\n"
        "import hypersonic_rans_amd as H\n"
        "from hypersonic_rans_amd import synth\n"
        "n = 24 << 20\n"
        "data = synth.nonstationary(n, seed=80)\n"
        "stream, plan = H.encode(H.MT, 64, 11, data, block_size=1 << 18, index_interval=8, independent_blocks=True)\n"
        "ctx = H.Context(0)\n"
        "d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()\n"
        "d_out = torch.zeros(n, dtype=torch.uint8, device='cuda')\n"
        "dp = ctx.make_device_plan(plan)\n"
        "ctx.decode_device(dp, d_in, d_out, stream_length=stream.size)\n"
        "assert ctx.status(dp) == 0 and np.array_equal(d_out.cpu().numpy(), data)\n"
        "print('ok', dp.launch_info()['spread'])\n")
    # (round 6: a plan without single-symbol blocks takes the host-dealt launch, `spread` 2 — tests/test_gpu_dealt.py; HSRANS_DEALT=0 keeps k_decode_spread,
    # HSRANS_SPREAD=0 the one-block-per-workgroup launch)
    for env, want in (({"HSRANS_SPREAD": "0"}, ("ok 0",)), ({"HSRANS_DEALT": "0"}, ("ok 1",)), ({}, ("ok 1", "ok 2"))):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0 and any(w in r.stdout for w in want), (env, r.stdout[-500:], r.stderr[-2000:])


def test_dynamic_block_order_variants_in_a_subprocess():
    """The A/B switches of the grouped launch (static order, 16-wave workgroups, no raised priority) all decode the same bytes: each is an environment variable read when a plan is made / launched, so each runs in
    its own process."""
    code = (
        "import numpy as np, torch, hashlib\n"
        "import hypersonic_rans_amd as H\n"
        "from hypersonic_rans_amd import synth\n"
        "n = 24 << 20\n"
        "data = synth.nonstationary(n, seed=78)\n"
        "stream, plan = H.encode(H.MT, 64, 11, data, block_size=1 << 13, index_interval=8)\n"
        "ctx = H.Context(0)\n"
        "d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()\n"
        "d_out = torch.zeros(n, dtype=torch.uint8, device='cuda')\n"
        "dp = ctx.make_device_plan(plan)\n"
        "for _ in range(3):\n"
        "    ctx.decode_device(dp, d_in, d_out, stream_length=stream.size)\n"
        "assert ctx.status(dp) == 0 and np.array_equal(d_out.cpu().numpy(), data)\n"
        "print('ok', dp.launch_info()['dynamic_groups'], dp.launch_info()['waves_per_block'])\n")
    for env, want in (({"HSRANS_GROUP_STATIC": "1"}, "ok 0 8"), ({"HSRANS_WAVES_PER_WG": "16"}, "ok 1 16"), ({"HSRANS_GROUP_PRIO": "0"}, "ok 1 8")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0 and want in r.stdout, (env, r.stdout[-500:], r.stderr[-2000:])


def test_config5_eight_gib_of_mt_streams_through_the_pipelined_host_path(gpu_ctx):
    """BASELINE config 5 shape on one GPU: 8 x 2^30 B mt_ streams in pinned host memory, upload / decode / download overlapped
    over slices of the plan (hsrans_hpipe through the C ABI); every stream's output compared with its source.  The streams are
    written by the GPU encoder (8 GiB through the scalar host encoder would take minutes) — its streams are checked against the
    oracle and the real reference elsewhere (test_gpu_parity.py, test_oracle_vs_ref.py)."""
    import torch
    from hypersonic_rans_amd import pipeline
    n = 1 << 30
    base = _tiled(n)
    host_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_enc = torch.empty(H.capacity(MT, 64, n), dtype=torch.uint8, device="cuda")
    total_s = 0.0
    for k in range(8):
        data = base if k == 0 else synth._permutation(2000 + k)[base]
        d_in.copy_(torch.from_numpy(data))
        m, dplan = gpu_ctx.encode_device(MT, 64, 11, d_in, d_enc, block_size=1 << 18, index_interval=256, want_plan=True)
        plan = gpu_ctx.read_device_plan(dplan, capacity=64 << 20)
        host_stream = torch.empty(m, dtype=torch.uint8).pin_memory()
        host_stream.copy_(d_enc[:m])
        torch.cuda.synchronize()
        dec = pipeline.PipelinedHostDecoder(gpu_ctx, plan)  # default slicing (hsrans_hpipe_create n_slices = 0: by size)
        host_out.fill_(0xCC)
        t0 = time.perf_counter()
        dec.decode(host_stream, host_out)
        total_s += time.perf_counter() - t0
        assert np.array_equal(host_out.numpy(), data), k
        dec.close()
        del dec, dplan, host_stream
    print(f"\n8 x 2^30 B, pipelined host decode: {8 * n / total_s / 1e9:.1f} GB/s decoded (PCIe-inclusive)")


def test_config3_spilled_table_and_single_chain_variants_are_bit_exact():
    """The comparison builds of BASELINE config 3 (decode table left in global memory: HSRANS_TABLE_SPILL=1; no two-chain kernel:
    HSRANS_DUAL=0; un-indexed streams on the general kernel: HSRANS_SINGLE_FAST=0) must be as exact as the defaults.  The
    library reads these knobs once per process, hence the child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
d = synth.enwik8_shaped(3_000_001, seed=9)
ctx = H.Context(0)
for S in (64, 32):
    for bits in (11, 13, 15):
        for kw in (dict(index_groups=H.index_boundaries(S, bits, d.size, ctx)), dict(index_interval=32)):
            s, plan = H.encode(H.RAW, S, bits, d, **kw)
            got = ctx.decode(H.RAW, S, bits, s, plan=plan)
            assert np.array_equal(got, d), (S, bits, list(kw))
        got = ctx.decode(H.RAW, S, bits, H.encode(H.RAW, S, bits, d[:200_003]))  # no index: one chain
        assert np.array_equal(got, d[:200_003]), (S, bits, 'single chain')
print('ok')
""" % (root, root)
    for env in ({"HSRANS_TABLE_SPILL": "1"}, {"HSRANS_DUAL": "0"}, {"HSRANS_SINGLE_FAST": "0"}, {"HSRANS_DUAL": "2"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
