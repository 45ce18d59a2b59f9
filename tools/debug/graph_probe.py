#!/usr/bin/env python3
"""K rotated launches of the headline decode captured into one hipGraph against the same K plain launches: wall clock per step.
    python tools/debug/graph_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

n, S, bits, P, K = 100_000_000, 64, 11, 4, 20
ctx = H.Context(0)
ctx.calibrate(bits=bits)
base = synth.enwik8_shaped(n, seed=20241008)
groups = H.index_boundaries(S, bits, n, ctx)
dplans, d_in, d_out, lens = [], [], [], []
for k in range(P):
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
    lens.append(s.size)
    dplans.append(ctx.make_device_plan(p))
    d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
    d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
    ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
    torch.cuda.synchronize()
    assert np.array_equal(d_out[k].cpu().numpy(), data)


def steps(stream=None):
    for t in range(K):
        k = t % P
        ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k], stream=stream)


side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    steps(side)
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        steps(side)
torch.cuda.synchronize()


def timed(fn, reps):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append((a.elapsed_time(b) / K * 1e3, (time.perf_counter() - t0) / K * 1e6))
    return out


for _ in range(30):
    steps()
res = {}
for name, fn in (("plain", steps), ("graph", g.replay), ("plain again", steps), ("graph again", g.replay)):
    for _ in range(10):
        fn()
    r = timed(fn, 15)
    res[name] = {"events_us_per_step_median": round(float(np.median([x[0] for x in r])), 2), "wall_us_per_step_median": round(float(np.median([x[1] for x in r])), 2)}
print(json.dumps(res))
