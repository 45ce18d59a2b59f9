#!/usr/bin/env python3
"""Decode throughput of every container / state count / histogram width on one GPU (device-resident buffers, kernel-only,
bit-exact check against the input).  Writes one JSON line per configuration; used for DESIGN.md's table (BASELINE configs 2-4)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", type=int, default=100_000_000)
ap.add_argument("--only-raw", action="store_true", help="raw rows only (e.g. for the HSRANS_TABLE_SPILL=1 / HSRANS_DUAL=0 comparison runs)")
ap.add_argument("--bits", default="", help="comma-separated histogram widths: raw 64-state rows of these widths only")
ap.add_argument("--states", default="64", help="with --bits: comma-separated state counts")
ap.add_argument("--tag", default="", help="copied into every row (which environment the library ran under)")
args = ap.parse_args()
n = args.size
ctx = H.Context(0)
data = synth.enwik8_shaped(n)
d_ref = torch.from_numpy(data).cuda()
names = {H.RAW: "raw", H.BLOCK: "block_", H.MT: "mt_"}


def measure(container, S, bits, interval, block_size=0, reps=20):
    """interval: checkpoints every `interval` groups, 0 = none, "wave" = one chain per resident wavefront (hsrans_index_boundaries)"""
    t0 = time.perf_counter()
    if interval == "wave":
        s, plan = H.encode(container, S, bits, data, index_groups=H.index_boundaries(S, bits, n, ctx))
    elif interval:
        s, plan = H.encode(container, S, bits, data, index_interval=interval, block_size=block_size)
    else:
        s = H.encode(container, S, bits, data, block_size=block_size) if block_size else H.encode(container, S, bits, data)
        plan = H.plan_build(container, S, bits, s)
    t_enc = time.perf_counter() - t0
    d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dp = ctx.make_device_plan(plan)
    ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
    torch.cuda.synchronize()
    ok = ctx.status(dp) == 0 and bool(torch.equal(d_out, d_ref))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    info = dp.launch_info()
    row = {"container": names[container], "states": S, "bits": bits, "interval": interval, "block_size": block_size, "chains": H.plan_chain_count(plan),
           "ratio": round(s.size / n, 4), "ms": round(ms, 4), "MiB_s": round(n / 2**20 / (ms * 1e-3)), "hbm_frac": round((s.size + n) / (ms * 1e-3) / 8e12, 4),
           "bit_exact": ok, "grid": info["grid"], "block": info["block"], "lds": info["lds_bytes"], "shared_table": info["shared_table"],
           "table_mode": info["table_mode"], "chains_per_wave": info["chains_per_wave"], "plan_bytes": int(plan.size), "encode_s": round(t_enc, 2), "tag": args.tag}
    print(json.dumps(row), flush=True)


if args.bits:
    for bits in (int(b) for b in args.bits.split(",")):
        for S in (int(x) for x in args.states.split(",")):
            measure(H.RAW, S, bits, "wave")
            measure(H.RAW, S, bits, 32)
    sys.exit(0)
for bits in (10, 11, 12, 13, 14, 15):
    measure(H.RAW, 64, bits, "wave")
    measure(H.RAW, 64, bits, 32)
for bits in (11, 14):
    measure(H.RAW, 32, bits, "wave")
    measure(H.RAW, 32, bits, 32)
if args.only_raw:
    sys.exit(0)
for bits in (11, 12, 14, 15):
    measure(H.MT, 64, bits, 0, block_size=1 << 16)       # as the reference's mt_ encoder typically emits: 64 KiB blocks, no sidecar
measure(H.MT, 64, 11, 0, block_size=1 << 18)             # BASELINE config 4 block size (256 KiB)
measure(H.MT, 64, 11, 32, block_size=1 << 18)            # the same with checkpoints inside the blocks
measure(H.MT, 32, 11, 0, block_size=1 << 16)
measure(H.BLOCK, 64, 11, 32, block_size=1 << 18)         # block_ is one chain by format; with a plan it parallelises
if n <= 16_000_000:
    measure(H.BLOCK, 64, 11, 0, block_size=1 << 18, reps=2)  # device walk of the inline headers: one wavefront
    measure(H.RAW, 64, 11, 0, reps=2)                        # raw without plan: one wavefront
