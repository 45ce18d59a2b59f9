"""One rank of tests/test_sharded_rccl.py (started by `python -m torch.distributed.run`, one process per GPU, backend nccl = RCCL).

Every rank builds the same stream, takes its shard through hypersonic_rans_amd.sharded.ShardedDecoder — since round 5 a thin wrapper
over the C ABI's hsrans_comm_create / hsrans_sharded_create / hsrans_decode_sharded (csrc/hsrans_comm.cpp: the communicator, the
layout, the grouped ncclSend / ncclRecv on RCCL's own stream all live in the library; torch.distributed only starts the ranks and
carries rank 0's ncclUniqueId) — and compares what it ends up holding with the CPU oracle's bytes.  Legs: gather none / all / root, each with parts 1 and 4 (4 = the exchange of sub-run k
is posted behind its decode and travels while sub-run k + 1 decodes: sharded.pipelined_gather's stream-ordering assumption,
which gloo cannot check).  Fan-out this replaces: /root/reference/src/mt_rANS32x64_16w_decode.cpp:182-224.
Exit code != 0 on any mismatch; rank 0 prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import hypersonic_rans_amd as H
from hypersonic_rans_amd import sharded, synth
from oracle_lib import BLOCK, MT, RAW, Oracle


def main():
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    n_visible = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % n_visible)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctx = H.Context(dev.index)
    oracle = Oracle()
    report, bad = [], 0
    for container, ocont, name in ((H.MT, MT, "mt_"), (H.RAW, RAW, "raw"), (H.BLOCK, BLOCK, "block_")):
        data = synth.nonstationary(6_000_011, seed=11)
        stream, plan = H.encode(container, 64, 11, data, index_interval=32, block_size=0 if container == H.RAW else 65536)
        r, want = oracle.decode(ocont, 64, 11, stream, data.size)
        assert r == data.size and np.array_equal(want, data), "oracle failed to round-trip"
        d_want = torch.from_numpy(want).to(dev)
        for gather, root in (("none", None), ("all", None), ("root", 0), ("root", world - 1)):
            for parts in (1, 4):
                weights = sharded.root_weights(world, root, 0.5) if root is not None else None  # an uneven split too
                dec = sharded.ShardedDecoder(ctx, plan, parts=parts, weights=weights, root=root)
                assert dec.uses_c_abi and dec.comm is not None and dec.comm.world == world and H.load_library().hsrans_comm_rccl_version() > 0
                d_window = dec.upload_window(stream, dev)
                out = dec.alloc_out(dev)
                for rep in range(3):  # repeated: the exchange must also be right when the buffers already hold the right bytes of the previous step ...
                    if rep == 1:
                        out.fill_(0xA5)  # ... and when they hold junk
                    dec.step(d_window, out, gather=gather != "none")
                    torch.cuda.synchronize()
                    assert dec.global_status() == 0
                    if gather != "none" and (root is None or rank == root):
                        ok = torch.equal(out[:data.size], d_want)
                    else:
                        b, e = dec.ranges[rank]
                        ok = torch.equal(out[b - dec.out_base:e - dec.out_base], d_want[b:e])
                    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    bad += int(t.item())
                    report.append({"container": name, "gather": gather, "root": root, "parts": parts, "rep": rep, "ok": int(t.item()) == 0})
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "checks": len(report), "failed": [r for r in report if not r["ok"]], "backend": dist.get_backend()}), flush=True)
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
