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
