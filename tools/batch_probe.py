#!/usr/bin/env python3
"""K streams in one launch against the same streams one launch each: sustained, rotated, in ONE process (alternating windows), with
the per-class finish times of the batch launch (HSRANS_BATCH_STAMPS) and an optional fit of the batch's class weights.

    python tools/batch_probe.py [--pairs 4] [--size 100000000] [--fit 3] [--index wave|G] [--out FILE]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HSRANS_BATCH_STAMPS"] = "1"
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--states", type=int, default=64)
ap.add_argument("--index", default="wave")
ap.add_argument("--window", type=int, default=48, help="batch launches per window (serial windows run pairs x as many)")
ap.add_argument("--windows", type=int, default=6)
ap.add_argument("--fit", type=int, default=0, help="iterations of fitting the batch's class weights to the finish times")
ap.add_argument("--no-calibrate", action="store_true")
ap.add_argument("--batch-index", action="store_true", help="index every stream for its share of the batch launch (hsrans_index_boundaries_batch) and fit the lengths at the batch's run length (hsrans_ctx_calibrate_runs)")
ap.add_argument("--out", default="")
a = ap.parse_args()
n, S, bits, P = a.size, a.states, a.bits, a.pairs
cache = f"/tmp/zipf_{n}_20241008.bin"
if os.path.exists(cache):
    base = np.fromfile(cache, np.uint8)
else:
    base = synth.enwik8_shaped(n, seed=20241008)
    base.tofile(cache)
ctx = H.Context(0)
cal = None if (a.no_calibrate or S != 64 or bits > 12) else ctx.calibrate(bits=bits)
if a.batch_index and cal is not None:
    run = n / S / 8192.0
    cal["runs"] = [ctx.calibrate_runs(bits=bits, copies=c) for c in sorted({min(16, max(2, round(run / 96))), min(16, max(2, round(P * run / 96)))})]
    print(json.dumps(cal), flush=True)
groups = H.index_boundaries(S, bits, n, ctx) if a.index == "wave" else None
dplans, d_in, d_out, lens, datas, streams = [], [], [], [], [], []
for k in range(P):
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    if groups is not None:
        s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
    else:
        s, p = H.encode(H.RAW, S, bits, data, index_interval=int(a.index))
    lens.append(s.size)
    datas.append(data)
    dplans.append(ctx.make_device_plan(p))
    streams.append(s)
    d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
    d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
alg = float(np.mean(lens)) + n


def check(tag):
    torch.cuda.synchronize()
    for k in range(P):
        assert np.array_equal(d_out[k].cpu().numpy(), datas[k]), f"{tag}: pair {k} not bit-exact"
        d_out[k].zero_()


def serial_window(count):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for t in range(count):
        k = t % P
        ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
    eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb) / count * 1e3


def batch_window(batch, count):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for t in range(count):
        ctx.decode_device_batch(batch, d_in, d_out, stream_lengths=lens)
    eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb) / count * 1e3 / P


def class_finish(batch):
    """mean / max finish time (us after the launch's first wave) of the 8 wave classes of the batch's LAST launch"""
    f = batch.read_finish().astype(np.int64)
    W = f.size - 1
    t0 = f[W]
    rel = (f[:W] - t0) / 100.0  # 100 MHz
    wg, wave = np.arange(W) // 16, np.arange(W) % 16
    cls = (wg >= (W // 16 + 1) // 2) * 4 + wave // 4
    return [float(rel[cls == c].mean()) for c in range(8)], [float(rel[cls == c].max()) for c in range(8)], float(rel.max())


for k in range(P):
    ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
check("serial")
report = {"size": n, "pairs": P, "states": S, "bits": bits, "index": a.index, "batch_index": a.batch_index, "calibration": cal, "rounds": []}
weights = None
for it in range(a.fit + 1):
    if weights is not None:
        os.environ["HSRANS_BATCH_WEIGHTS"] = ",".join(str(int(w)) for w in weights)
    if a.batch_index:
        bplans = [ctx.make_device_plan(ctx.index_build_at(H.RAW, S, bits, streams[k], H.index_boundaries_batch(S, bits, [n] * P, k, ctx))) for k in range(P)]
        batch = ctx.make_batch(bplans)
    else:
        batch = ctx.make_batch(dplans)
    info = batch.info()
    ctx.decode_device_batch(batch, d_in, d_out, stream_lengths=lens)
    check("batch")
    # settle, then alternate
    for _ in range(3):
        serial_window(a.window * P)
        batch_window(batch, a.window)
    ser, bat = [], []
    for w in range(a.windows):
        ser.append(serial_window(a.window * P))
        bat.append(batch_window(batch, a.window))
    mean_c, max_c, last = class_finish(batch)
    rnd = {"weights": info["class_weights"], "imbalance": info["imbalance"], "serial_us_per_stream": [round(x, 2) for x in ser], "batch_us_per_stream": [round(x, 2) for x in bat],
           "serial_frac": alg / (np.median(ser) * 1e-6) / 8e12, "batch_frac": alg / (np.median(bat) * 1e-6) / 8e12,
           "class_finish_mean_us": [round(x, 2) for x in mean_c], "class_finish_max_us": [round(x, 2) for x in max_c], "last_wave_us": round(last, 2)}
    report["rounds"].append(rnd)
    print(json.dumps(rnd), flush=True)
    w0 = np.array(info["class_weights"], float)
    fin = np.array(mean_c)
    w1 = w0 * (fin.mean() / fin) ** 0.8
    weights = np.round(w1 * 8000 / w1.sum())
    batch.close()
check("end")
if a.out:
    with open(a.out, "a") as f:
        f.write(json.dumps(report) + "\n")
