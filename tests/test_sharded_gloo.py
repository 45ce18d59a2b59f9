"""world_size-2 check of the multi-GPU path's host side on CPU (gloo): chain sharding + the gather of decoded ranges.
The per-rank decode itself needs a GPU, so here each rank fills its range with the oracle's plan interpreter."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hypersonic_rans_amd as H
from hypersonic_rans_amd import sharded, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, container, root, q):
    from oracle_lib import Oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.nonstationary(700_003, seed=3)
        stream, plan = H.encode(container, 64, 11, data, index_interval=32)
        runs = sharded.shard_chains(plan, world)
        ranges = [sharded.local_range(plan, f, c) for f, c in runs]
        first, count = runs[rank]
        local = torch.zeros(data.size, dtype=torch.uint8)
        upload = 0
        if count:
            # the rank only "uploads" the stream bytes its chains can read (hsrans_plan_stream_ranges); everything else is junk
            masked = np.full_like(stream, 0xEE)
            for lo, hi in H.plan_stream_ranges(plan, first, count):
                masked[lo:hi] = stream[lo:hi]
                upload += hi - lo
            r, part = Oracle().exec_plan(H.plan_slice(plan, first, count), masked, data.size)
            assert r == data.size
            b, e = ranges[rank]
            local[b:e] = torch.from_numpy(part[b:e].copy())
        full = sharded.gather_ranges(local, ranges, root=root)  # in place: point-to-point straight into `local`
        assert full.data_ptr() == local.data_ptr()
        complete = bool(np.array_equal(full.numpy(), data))
        if root is not None and rank != root:  # a root gather leaves the other ranks with their own range only
            b, e = ranges[rank]
            complete = bool(np.array_equal(full.numpy()[b:e], data[b:e])) and not bool(np.array_equal(full.numpy(), data))
        q.put((rank, complete, ranges, upload, stream.size))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("container,root", ((H.RAW, None), (H.MT, None), (H.MT, 0)))
def test_two_ranks_gather(container, root):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, container, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results)
    # the two uploads together are the stream once — plus the shared histogram of a raw stream on both ranks, or the part of
    # an mt_ block between its histogram and the checkpoint the second rank starts from — not twice
    assert sum(r[3] for r in results) < 1.1 * results[0][4]
    ranges = results[0][2]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 700_003


def _pipelined_worker(rank, world, port, container, root, parts, root_share, q):
    """The pipelined exchange (sharded.pipelined_gather) with the sub-runs decoded by the ORACLE's plan interpreter: sub-run k's
    ranges are posted before sub-run k+1 is decoded; a rank that is not the root holds only its own range of the output."""
    from oracle_lib import Oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.nonstationary(900_001, seed=5)
        stream, plan = H.encode(container, 64, 11, data, index_interval=32, block_size=0 if container == H.RAW else 65536)
        weights = sharded.root_weights(world, root, root_share) if (root is not None and root_share) else None
        layout = sharded.ShardLayout(plan, world, parts, weights)
        b0, e0 = layout.ranges[rank]
        local = root is not None and rank != root
        out_base = b0 if local else 0
        out = torch.zeros((e0 - b0) if local else data.size, dtype=torch.uint8)
        lo, hi = layout.windows[rank]
        masked = np.full_like(stream, 0xEE)
        masked[lo:hi] = stream[lo:hi]
        if container == H.RAW:  # the shared histogram travels with the plan on the GPU; the oracle's interpreter reads it from the stream
            (hb, he), _ = H.plan_stream_ranges(plan, *layout.runs[rank]) if layout.runs[rank][1] else ((0, 0), None)
            masked[hb:he] = stream[hb:he]
        order = []

        def decode_part(k):
            order.append(k)
            f, c = layout.sub_runs[rank][k]
            if c == 0:
                return
            r, part = Oracle().exec_plan(H.plan_slice(plan, f, c), masked, data.size)
            assert r == data.size
            b, e = layout.sub_ranges[rank][k]
            out[b - out_base:e - out_base] = torch.from_numpy(part[b:e].copy())

        sharded.pipelined_gather(out, layout, decode_part, None, root, out_base)
        if local:
            ok = bool(np.array_equal(out.numpy(), data[b0:e0]))
        else:
            ok = bool(np.array_equal(out.numpy(), data))
        q.put((rank, ok, layout.ranges, order, [list(x) for x in layout.sub_ranges[rank]], int(out.numel())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,container,root,parts,root_share", (
    (2, H.MT, None, 3, 0.0),      # everyone gets everything, three sub-runs
    (3, H.MT, 0, 4, 0.6),         # weighted gather to rank 0: it decodes 60 % and sends nothing
    (2, H.RAW, 1, 2, 0.0),        # raw stream + index, gather to rank 1
    (4, H.BLOCK, None, 2, 0.0),
))
def test_pipelined_gather_is_bit_exact(world, container, root, parts, root_share):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, container, root, parts, root_share, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results)
    ranges = results[0][2]
    n = 900_001
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    for rank, _ok, _ranges, order, subs, held in results:
        assert order == list(range(parts))
        # the sub-runs tile the rank's range
        nonempty = [s for s in subs if s[1] > s[0]]
        assert nonempty[0][0] == ranges[rank][0] and nonempty[-1][1] == ranges[rank][1]
        assert all(nonempty[i][1] == nonempty[i + 1][0] for i in range(len(nonempty) - 1))
        if root is not None and rank != root:
            assert held == ranges[rank][1] - ranges[rank][0]  # not the whole output
    if root_share:
        share = (ranges[root][1] - ranges[root][0]) / n
        assert abs(share - root_share) < 0.05


def test_shard_weights_and_balance_formula():
    data = synth.nonstationary(400_000, seed=9)
    stream, plan = H.encode(H.MT, 64, 11, data, index_interval=32, block_size=65536)
    runs = sharded.shard_chains(plan, 4, weights=[3, 1, 1, 1])
    sizes = [sharded.local_range(plan, f, c) for f, c in runs]
    assert sizes[0][0] == 0 and sizes[-1][1] == data.size
    assert abs((sizes[0][1] - sizes[0][0]) / data.size - 0.5) < 0.03
    # a = D / (D + B): decode 2.1 TB/s against 1.07 TB/s inbound -> two thirds on the root; never below the equal share
    assert abs(sharded.balanced_root_share(8, 2.1e12, 1.07e12) - 0.6625) < 1e-3
    assert sharded.balanced_root_share(8, 1.0, 1e9) == 1 / 8
    assert sharded.balanced_root_share(1, 1.0, 1.0) == 1.0
    assert sharded.root_weights(4, 2, 0.7)[2] == 0.7 and abs(sum(sharded.root_weights(4, 2, 0.7)) - 1) < 1e-12


# ---- ShardLayout is pure host arithmetic: property test over containers, world sizes, sub-run counts and weights -------------
from hypothesis import given, settings, strategies as st


@settings(max_examples=40, deadline=None)
@given(container=st.sampled_from((H.RAW, H.MT, H.BLOCK)), n=st.integers(70_000, 400_000), world=st.integers(1, 9), parts=st.integers(1, 5),
       interval=st.sampled_from((4, 16, 64)), heavy=st.integers(0, 8), seed=st.integers(0, 1000))
def test_shard_layout_tiles_output_and_stream(container, n, world, parts, interval, heavy, seed):
    """Whatever the plan and the split: the ranks' ranges tile the output in order, a rank's sub-runs tile its run and its range,
    the window of a rank covers what its chains read (hsrans_plan_stream_ranges of its run and of every sub-run), and weights move
    bytes towards the heavy rank."""
    data = synth.nonstationary(n, seed=seed)
    stream, plan = H.encode(container, 64, 11, data, index_interval=interval, block_size=0 if container == H.RAW else 32768)
    weights = None
    if heavy and world > 1:
        weights = [1.0] * world
        weights[heavy % world] = 3.0
    lay = sharded.ShardLayout(plan, world, parts, weights)
    n_chains = H.plan_chain_count(plan)
    assert sum(c for _f, c in lay.runs) == n_chains and lay.runs[0][0] == 0
    pos = 0
    for r in range(world):
        f, c = lay.runs[r]
        assert f == (lay.runs[r - 1][0] + lay.runs[r - 1][1] if r else 0)
        if c == 0:
            continue
        b, e = lay.ranges[r]
        assert b == pos and e > b
        pos = e
        subs = [(sf, sc) for sf, sc in lay.sub_runs[r] if sc]
        assert subs[0][0] == f and sum(sc for _sf, sc in subs) == c
        sub_pos = b
        lo, hi = lay.windows[r]
        assert lo % 16 == 0
        for (sf, sc), (sb, se) in zip(lay.sub_runs[r], lay.sub_ranges[r]):
            if sc == 0:
                continue
            assert sb == sub_pos and se > sb
            sub_pos = se
            (_hb, _he), (bb, be) = H.plan_stream_ranges(plan, sf, sc)
            assert lo <= bb and be <= hi, "a sub-run reads outside its rank's window"
        assert sub_pos == e
    assert pos == n
    if weights is not None and n_chains >= 8 * world:
        hv = heavy % world
        sizes = [e - b for b, e in lay.ranges]
        assert sizes[hv] >= max(s for i, s in enumerate(sizes) if i != hv)
