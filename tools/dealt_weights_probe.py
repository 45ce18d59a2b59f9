#!/usr/bin/env python3
"""Class lengths for the host-dealt launch at 13 / 14 bits (k_decode_dealt_rank): HSRANS_DEALT_WEIGHTS is read at every dealing, so one
process times several sets on the same buffers (100 MB mt_, 256 KiB blocks, G = 16, four copies rotated), every set twice.
    python tools/dealt_weights_probe.py > profiles/rNN_dealt_weights_14bit.txt"""
import os, sys, json, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0); ctx.calibrate()
n = 100_000_000
d_in = torch.from_numpy(synth.enwik8_shaped(n, seed=1)).cuda()
def timed(fn, launches=40, regions=5, settle_s=0.025):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(regions):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / launches)
    return sorted(ts)[len(ts) // 2] * 1e3
for bits in (14, 13):
    enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    m, dplan = ctx.encode_device(H.MT, 64, bits, d_in, enc, block_size=1 << 18, index_interval=16, want_plan=True)
    streams = [enc[:m].clone() for _ in range(4)]; outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
    k = [0]
    def rot():
        i = k[0] % 4; k[0] += 1
        ctx.decode_device(dplan, streams[i], outs[i], stream_length=m)
    for rnd in range(2):
        for w in ("", "1192,1159,1120,1072,976,907,829,745", "1100,1075,1050,1020,975,950,920,900", "1300,1250,1180,1100,980,860,720,610", "1250,1200,1150,1080,1000,900,780,650", "1000,1000,1000,1000,1000,1000,1000,1000"):
            if w: os.environ["HSRANS_DEALT_WEIGHTS"] = w
            else: os.environ.pop("HSRANS_DEALT_WEIGHTS", None)
            us = timed(rot)
            ok = ctx.status(dplan) == 0 and all(bool(torch.equal(o, d_in)) for o in outs)
            print(bits, rnd, w or "calibrated", round(us, 2), dplan.launch_info()["spread"], ok, flush=True)
    os.environ.pop("HSRANS_DEALT_WEIGHTS", None)
