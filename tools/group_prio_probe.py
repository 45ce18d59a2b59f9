#!/usr/bin/env python3
"""Grouped launches (block_/mt_ plans with checkpoints, one block per workgroup and round): which share of its run should a wave of
each scheduling class decode at raised instruction priority (s_setprio)?  A SIMD serves its oldest wave first; in a grouped launch
every workgroup gets a whole block whatever its age, so the older workgroup of a CU is done long before the younger one, which then
has the CU to itself at half the occupancy.  HSRANS_GROUP_PRIO_CLASS = ten per-mille values (KParams::group_prio_class: eight classes
for evenly split groups, two grid halves for class-weighted ones) is read at every launch, so one process tries them all on the same
buffers, alternating, rotated over COPIES (stream, output) pairs.

    python tools/group_prio_probe.py --size 134217728 --block 262144 --interval 256 64 32"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 27)
ap.add_argument("--block", type=int, default=1 << 18)
ap.add_argument("--interval", type=int, nargs="+", default=[256, 64, 32])
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--copies", type=int, default=4)
ap.add_argument("--configs", nargs="*", default=None)
args = ap.parse_args()

CONFIGS = args.configs or [
    "default",
    "0,0,0,0,0,0,0,0,0,0",
    "0,0,350,350,0,0,350,350,0,0",
    "0,0,350,350,300,300,600,600,0,300",
    "0,100,200,300,400,500,600,700,0,400",
    "0,0,200,300,500,600,750,900,0,500",
    "0,0,350,350,500,500,800,800,0,500",
    "0,0,350,350,700,700,1000,1000,0,700",
    "0,0,350,350,1000,1000,1000,1000,0,1000",
    "0,0,0,0,1000,1000,1000,1000,0,1000",
    "0,200,400,600,700,800,900,1000,0,800",
]

ctx = H.Context(0)
ctx.calibrate()
n = args.size
d_in = torch.from_numpy(synth.enwik8_shaped(n, seed=20241008)).cuda()


def timed(fn, launches=30, regions=5, settle_s=0.03):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(10):
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
    streams = [enc[:m].clone() for _ in range(args.copies)]
    outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(args.copies)]
    k = [0]

    def rotated():
        i = k[0] % args.copies
        k[0] += 1
        ctx.decode_device(dplan, streams[i], outs[i], stream_length=m)

    for rnd in range(2):  # every configuration twice, alternating: a drifting box shows up as a difference between the rounds
        for cfg in CONFIGS:
            if cfg == "default":
                os.environ.pop("HSRANS_GROUP_PRIO_CLASS", None)
            else:
                os.environ["HSRANS_GROUP_PRIO_CLASS"] = cfg
            us = timed(rotated)
            info = dplan.launch_info()
            ok = ctx.status(dplan) == 0 and all(bool(torch.equal(o, d_in)) for o in outs)
            print(json.dumps({"size": n, "block": args.block, "interval": interval, "bits": args.bits, "round": rnd, "prio_class": cfg, "rotated_us": round(us, 2),
                              "frac_of_8TBs": round((n + m) / (us * 1e-6) / 8e12, 3), "grid": info["grid"], "block_threads": info["block"], "spread": info["spread"],
                              "dynamic": info["dynamic_groups"], "bit_exact": ok}), flush=True)
    os.environ.pop("HSRANS_GROUP_PRIO_CLASS", None)
    del streams, outs, dplan, enc
    torch.cuda.empty_cache()
