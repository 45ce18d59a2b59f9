"""GPU encoder rate: hsrans_encode_device on the 100 MB enwik8-shaped input, HBM-resident.
Prints one JSON line per configuration.  Run on the GPU box: python tools/encode_rate.py [--size N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
ctx = H.Context(0)
d = synth.enwik8_shaped(args.size, seed=1)
d_in = torch.from_numpy(d).cuda()
for states, bits, block in ((64, 11, 1 << 16), (64, 11, 1 << 15), (64, 11, 1 << 18), (32, 11, 1 << 16), (64, 15, 1 << 16)):
    d_out = torch.empty(H.capacity(H.MT, states, d.size), dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block)  # synchronises its stream
        ts.append(time.perf_counter() - t0)
    best, mean = min(ts), sum(ts) / len(ts)
    print(json.dumps({"codec": f"mt_ rANS32x{states} 16w {bits}", "block": block, "size": args.size, "stream": n, "ratio": round(n / args.size, 4),
                      "ms_best": round(best * 1e3, 3), "ms_mean": round(mean * 1e3, 3), "GB_s_best": round(args.size / best / 1e9, 1)}))
