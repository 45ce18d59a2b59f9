#!/usr/bin/env python3
"""Diagnostic: why do the bench's four (stream, output) pairs replay at different speeds (39 / 45 / 45 / 43 us in BENCH_r03)?

Separates DATA (the stream's bytes: a byte permutation of one base block) from PLACEMENT (where a buffer lies):
  1. the 4 x 4 matrix "stream i decoded into output buffer j", each replayed (warm);
  2. stream i copied to other places (a fresh allocation; offsets inside one big buffer) and replayed into output 0;
  3. the rotated (cold) time of the bench's own order, and of the same rotation with the outputs swapped around.

    python tools/pair_probe.py [--size N] [--steps K]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--calibrate", action="store_true")
a = ap.parse_args()
n, S, bits, P = a.size, 64, a.bits, a.pairs
ctx = H.Context(0)
if a.calibrate:
    print(json.dumps({"calibration": ctx.calibrate(bits=bits)}), flush=True)
groups = H.index_boundaries(S, bits, n, ctx)
base = synth.enwik8_shaped(n, seed=20241008)
streams, plans, dplans, d_in, d_out, lens = [], [], [], [], [], []
for k in range(P):
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
    streams.append(s), plans.append(p), lens.append(s.size)
    dplans.append(ctx.make_device_plan(p))
    d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
    d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
    print(json.dumps({"pair": k, "compressed": int(s.size), "d_in": hex(d_in[-1].data_ptr()), "d_out": hex(d_out[-1].data_ptr())}), flush=True)


def timed(launch, steps=a.steps, warm=8):
    for i in range(warm):
        launch(i)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for i in range(steps):
        launch(warm + i)
    eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb) / steps * 1e3  # us


def dec(i, din, dout):
    ctx.decode_device(dplans[i], din, dout, stream_length=lens[i])


# 1. stream i -> output j, replayed
m = [[timed(lambda _, i=i, j=j: dec(i, d_in[i], d_out[j])) for j in range(P)] for i in range(P)]
print(json.dumps({"warm_us[stream i][output j]": [[round(x, 2) for x in r] for r in m]}), flush=True)
# twice more for the diagonal: how stable is one cell?
print(json.dumps({"warm_us diagonal again": [[round(timed(lambda _, i=i: dec(i, d_in[i], d_out[i])), 2) for i in range(P)] for _ in range(2)]}), flush=True)

# 2. the same stream bytes somewhere else
big = torch.zeros(max(lens) + (4 << 20), dtype=torch.uint8, device="cuda")
for i in range(P):
    row = {}
    fresh = d_in[i].clone()
    row["fresh allocation " + hex(fresh.data_ptr())] = round(timed(lambda _, i=i: dec(i, fresh, d_out[0])), 2)
    for off in (0, 16, 256, 4096, 65536, 1 << 20, (1 << 21) + 4096):
        view = big[off:off + d_in[i].numel()]
        view.copy_(d_in[i])
        row[f"big+{off}"] = round(timed(lambda _, i=i, v=view: dec(i, v, d_out[0])), 2)
    print(json.dumps({"stream": i, "placements_us": row}), flush=True)
# and the output somewhere else
bigo = torch.zeros(n + (4 << 20), dtype=torch.uint8, device="cuda")
row = {}
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 21) + 4096):
    row[f"out big+{off}"] = [round(timed(lambda _, i=i, v=bigo[off:off + n]: dec(i, d_in[i], v)), 2) for i in range(P)]
print(json.dumps({"output placements_us (per stream)": row}), flush=True)

# 3. rotated
rot = lambda perm: timed(lambda t: dec(t % P, d_in[t % P], d_out[perm[t % P]]), steps=max(a.steps, 40))
print(json.dumps({"rotated_us": {"i->i": [round(rot(list(range(P))), 2) for _ in range(3)], "i->i+1": round(rot([(i + 1) % P for i in range(P)]), 2),
                                 "i->0 (one output)": round(rot([0] * P), 2)}}), flush=True)
print("done")
