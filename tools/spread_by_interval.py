#!/usr/bin/env python3
"""mt_ decode of the 100 MB enwik8-shaped input against block size and index interval: the stream and its
sidecar plan come from the GPU encoder (hsrans_encode_device, checkpoints every G groups), the decode is hsrans_decode_device with
that plan.  Per row: the launch the library chose, the time per decode ROTATED over four copies of the stream and the output (no
replay out of the caches) and REPLAYED on one copy, both as the median of five regions of 40 launches behind 25 ms of sustained
launches (the first launches after an idle gap run slower: DESIGN 3), and the fraction of 8 TB/s on the algorithmic bytes
(decoded + compressed).  Run on the GPU box: python tools/spread_by_interval.py > profiles/rNN_spread_by_interval.txt"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ctx = H.Context(0)
ctx.calibrate()
for copies in (2, 3, 5):  # class lengths fitted for runs of ~190 / 290 / 480 groups too (100 MB, 128 MiB, 256 MiB per launch): the dealt launch picks the nearest
    ctx.calibrate_runs(copies=copies)
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
COPIES = 4


def timed(fn, launches=40, regions=5, settle_s=0.025):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(20):
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
    return sorted(ts)[len(ts) // 2] * 1e3  # us


for states, bits in ((64, 11), (64, 14)):
    for block in (1 << 18, 1 << 16):
        for interval in (8, 16, 32, 64, 128):
            if interval * states > block:
                continue
            enc = torch.empty(H.capacity(H.MT, states, d.size), dtype=torch.uint8, device="cuda")
            n, dplan = ctx.encode_device(H.MT, states, bits, d_in, enc, block_size=block, index_interval=interval, want_plan=True)
            streams = [enc[:n].clone() for _ in range(COPIES)]
            outs = [torch.empty(d.size, dtype=torch.uint8, device="cuda") for _ in range(COPIES)]
            k = [0]

            def rotated():
                i = k[0] % COPIES
                k[0] += 1
                ctx.decode_device(dplan, streams[i], outs[i], stream_length=n)

            def replayed():
                ctx.decode_device(dplan, streams[0], outs[0], stream_length=n)

            rot = timed(rotated)
            rep = timed(replayed)
            ok = ctx.status(dplan) == 0 and all(bool(torch.equal(o, d_in)) for o in outs)
            info = dplan.launch_info()
            print(json.dumps({"codec": f"mt_ rANS32x{states} 16w {bits}", "block": block, "interval": interval, "chains": info.get("chains"),
                              "launch": "spread" if info.get("spread") else "grouped" if info.get("dynamic_groups") is not None and info.get("shared_table") else "other",
                              "rotated_us": round(rot, 2), "replayed_us": round(rep, 2), "frac_of_8TBs_rotated": round((d.size + n) / (rot * 1e-6) / 8e12, 3),
                              "plan_MB": round(dplan.plan_bytes() / 1e6, 1) if hasattr(dplan, "plan_bytes") else None, "bit_exact": ok}), flush=True)
            del streams, outs, dplan, enc
            torch.cuda.empty_cache()
