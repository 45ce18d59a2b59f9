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


def _worker(rank, world, port, container, q):
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
        if count:
            r, part = Oracle().exec_plan(H.plan_slice(plan, first, count), stream, data.size)
            assert r == data.size
            b, e = ranges[rank]
            local[b:e] = torch.from_numpy(part[b:e].copy())
        full = sharded.gather_ranges(local, ranges)
        q.put((rank, bool(np.array_equal(full.numpy(), data)), ranges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("container", (H.RAW, H.MT))
def test_two_ranks_gather(container):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, container, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    ranges = results[0][2]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 700_003
