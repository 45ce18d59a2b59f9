#!/usr/bin/env python3
"""What every rank of a sharded decode would do, on the ONE GPU this pool has — a PROJECTION, not a multi-GPU
measurement: BASELINE config 4's stream (2^30 bytes, mt_, 256 KiB blocks + a checkpoint every G groups; GPU-encoded) is sharded with
hsrans_shard_layout for world = 2 / 4 / 8 exactly as `bench.py --gpus N` shards it, and rank r's GPU side — its device plans, its
window of the stream, its range of the output, hsrans_decode_sharded(HSRANS_SHARD_DECODE_ONLY) — runs alone on the device, rank
after rank.  Per row: every rank's time per step (rotated over COPIES (stream, output) pairs: a rank's 1.69 GB / world of traffic
per step times the copies stays beyond the 256 MB Infinity Cache), the slowest rank, and

    projected decode-only speedup = T(one rank decodes the whole stream) / max_r T(rank r)

which is what the 1 -> N curve would be if the exchange over xGMI were free (it is pipelined behind the decode in `parts`
sub-runs; its cost is NOT in this number).  `--parts 1 4` compares a rank's run as one sub-run / four; since round 6 the sub-runs
of a rank are ONE launch with a completion word per sub-run (HSRANS_SHARD_ONE_LAUNCH=0: one launch per sub-run, as in round 5).
Reference counterpart: every block of a stream handed to the pool in one pass, joined once
(/root/reference/src/mt_rANS32x64_16w_decode.cpp:182-224, :262).

    python tools/shard_projection.py > profiles/r06_shard_projection.jsonl"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import sharded, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--block", type=int, default=1 << 18)
ap.add_argument("--interval", type=int, nargs="+", default=[256, 64])
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--parts", type=int, nargs="+", default=[1, 4])
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--copies", type=int, default=4)
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--regions", type=int, default=5)
ap.add_argument("--label", default="")
args = ap.parse_args()

ctx = H.Context(0)
ctx.calibrate()
for copies in (2, 3, 5):  # class lengths fitted for runs of ~190 / 290 / 480 groups too (100 MB, 128 MiB, 256 MiB per launch): the dealt launch picks the nearest
    ctx.calibrate_runs(copies=copies)
n = args.size
d_in = torch.from_numpy(synth.enwik8_shaped(n, seed=20241008)).cuda()
COPIES = args.copies


def timed(fn, launches, regions, settle_s=0.03):
    """median over `regions` of (HIP events around `launches` calls) behind `settle_s` of sustained launches; microseconds per call"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(regions):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / launches)
    return sorted(ts)[len(ts) // 2] * 1e3


for interval in args.interval:
    enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    m, dplan = ctx.encode_device(H.MT, 64, args.bits, d_in, enc, block_size=args.block, index_interval=interval, want_plan=True)
    plan = ctx.read_device_plan(dplan, capacity=1 << 30)
    del dplan
    pad = (-m) % 16
    streams = [torch.cat([enc[:m], torch.zeros(pad + 16, dtype=torch.uint8, device="cuda")]) for _ in range(COPIES)]
    del enc
    outs = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(COPIES)]
    b_alg = n + m

    def run_world(world, parts):
        decs = [sharded.ShardedDecoder(ctx, plan, parts=parts, world=world, rank=r) for r in range(world)]
        per_rank, infos = [1e30] * world, [None] * world
        # two passes over the ranks, a rank's figure = the lower of its two medians: the first rank timed after `world` decoders were
        # created read 10 us above the others in one regeneration (65.8 against 54.4-55.4) and like them when timed again
        for _ in range(2):
            for r, dec in enumerate(decs):
                k = [0]

                def step():
                    i = k[0] % COPIES
                    k[0] += 1
                    dec.decode(streams[i], outs[i], gather=False)

                per_rank[r] = min(per_rank[r], timed(step, args.launches, args.regions))
                infos[r] = dec.launch_info()
        # bit-exact: every rank once more into a cleared copy, then the whole output against the input
        outs[0].zero_()
        for dec in decs:
            dec.decode(streams[0], outs[0], gather=False)
            dec.check()
        torch.cuda.synchronize()
        ok = bool(torch.equal(outs[0], d_in))
        one_launch = bool(getattr(decs[0].c, "info", {}).get("one_launch", 0))
        for dec in decs:
            dec.c.close()
        return per_rank, infos, ok, one_launch

    one_us, one_info, ok1, _ = run_world(1, 1)
    base = {"codec": f"mt_ rANS32x64 16w {args.bits}", "size": n, "compressed": m, "block": args.block, "interval": interval, "copies": COPIES,
            "projection": True, "label": args.label, "one_launch_env": os.environ.get("HSRANS_SHARD_ONE_LAUNCH", "")}
    print(json.dumps({**base, "world": 1, "parts": 1, "per_rank_us": [round(one_us[0], 1)], "max_us": round(one_us[0], 1),
                      "frac_of_8TBs": round(b_alg / (one_us[0] * 1e-6) / 8e12, 3), "launch": one_info[0], "bit_exact": ok1}), flush=True)
    for world in args.worlds:
        for parts in args.parts:
            per_rank, infos, ok, one_launch = run_world(world, parts)
            mx = max(per_rank)
            print(json.dumps({**base, "world": world, "parts": parts, "sub_runs_in_one_launch": one_launch, "per_rank_us": [round(t, 1) for t in per_rank],
                              "max_us": round(mx, 1), "one_rank_us": round(one_us[0], 1), "projected_decode_only_speedup": round(one_us[0] / mx, 2),
                              "projected_efficiency": round(one_us[0] / mx / world, 3),
                              "slowest_rank_frac_of_8TBs": round(b_alg / world / (mx * 1e-6) / 8e12, 3),
                              "launch_rank0": {k: infos[0][k] for k in ("grid", "block", "lds_bytes", "chains", "spread", "dynamic_groups") if infos[0] and k in infos[0]},
                              "bit_exact": ok}), flush=True)
    del streams, outs
    torch.cuda.empty_cache()
