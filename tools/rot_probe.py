#!/usr/bin/env python3
"""A/B helper: sustained time per 100 MB decode (rotated over P pairs, and one pair replayed) for the library / environment this
process was started with (HSRANS_LIB, HSRANS_DIRECT_WEIGHTS, ... are read when the library loads: one process per variant; for A/Bs
prefer tools/ab_probe.py: between processes one box moves by +-3 us).
Every pair is validated bit-exact first.  The synthetic data is cached in /tmp between processes (the generator makes 7 MB/s).

    [HSRANS_LIB=...] python tools/rot_probe.py [--tag NAME] [--windows 4] [--window 200] [--stamps]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="")
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--window", type=int, default=200)
ap.add_argument("--windows", type=int, default=4)
ap.add_argument("--calibrate", action="store_true")
ap.add_argument("--no-check", action="store_true", help="diagnostic builds that do not produce the output (HSRANS_DIAG_NO_STORES)")
a = ap.parse_args()
n, S, bits, P = a.size, 64, a.bits, a.pairs
stamps = bool(os.environ.get("HSRANS_DEBUG_STAMPS"))
cache = f"/tmp/zipf_{n}_20241008.bin"
if os.path.exists(cache):
    base = np.fromfile(cache, np.uint8)
else:
    base = synth.enwik8_shaped(n, seed=20241008)
    base.tofile(cache)
ctx = H.Context(0)
cal = ctx.calibrate(bits=bits) if a.calibrate else None
groups = H.index_boundaries(S, bits, n, ctx)
dplans, d_in, d_out, lens, plan_bytes = [], [], [], [], 0
for k in range(P):
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
    lens.append(s.size)
    plan_bytes = p.size
    dplans.append(ctx.make_device_plan(p))
    d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
    d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
    ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
    torch.cuda.synchronize()
    assert ctx.status(dplans[k]) == 0
    assert a.no_check or np.array_equal(d_out[k].cpu().numpy(), data), "not bit-exact"
    d_out[k].zero_()
L = H.load_library()


def run(pick, windows):
    out = []
    for w in range(windows):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for t in range(a.window):
            k = pick(w * a.window + t)
            ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
        eb.record()
        torch.cuda.synchronize()
        out.append(round(ea.elapsed_time(eb) / a.window * 1e3, 2))
    return out


def stamp_summary(dp):
    L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
    L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros(16384 * 8, np.uint64)
    L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 8).astype(np.int64)
    st = st[st[:, 3] > 0]
    rel = (st - st[:, 0].min()) / 100.0
    q = lambda c, p: round(float(np.percentile(rel[:, c], p)), 2)
    ghz = st[:, 5] / ((st[:, 3] - st[:, 0]) * 10.0)
    wait_us = (st[:, 7] & 0xFFFFFFFF) / (ghz * 1e3)   # shader clocks waiting at chunk crossings -> us
    store_us = ((st[:, 7] >> 32) & 0xFFFFFFFF) / (ghz * 1e3)
    done = rel[:, 3]
    slow, fast = done > np.percentile(done, 90), done <= np.percentile(done, 50)
    extra = {"wait_us_mean_all": round(float(wait_us.mean()), 2), "wait_us_mean_slowest10pct": round(float(wait_us[slow].mean()), 2), "wait_us_mean_fastest50pct": round(float(wait_us[fast].mean()), 2),
             "store_us_mean_all": round(float(store_us.mean()), 2), "store_us_mean_slowest10pct": round(float(store_us[slow].mean()), 2), "store_us_mean_fastest50pct": round(float(store_us[fast].mean()), 2),
             "lifetime_us_slowest10pct": round(float((rel[slow, 3] - rel[slow, 2]).mean()), 2), "lifetime_us_fastest50pct": round(float((rel[fast, 3] - rel[fast, 2]).mean()), 2),
             "corr_wait_done": round(float(np.corrcoef(wait_us, done)[0, 1]), 3)}
    return {**extra, "ready_p50": q(2, 50), "static_done_p50": q(4, 50), "done_p10": q(3, 10), "done_p50": q(3, 50), "done_p90": q(3, 90), "done_p99": q(3, 99), "done_max": q(3, 100),
            "done_mean": round(float(rel[:, 3].mean()), 2), "GHz_p50": round(float(np.median(st[:, 5] / ((st[:, 3] - st[:, 0]) * 10.0))), 3)}


run(lambda t: t % P, 2)  # settle
rot = run(lambda t: t % P, a.windows)
res = {"tag": a.tag, "lib": os.path.basename(os.environ.get("HSRANS_LIB", "default")), "rotated_us": rot, "rotated_us_median": float(np.median(rot)),
       "chains": H.plan_chain_count(p), "plan_bytes": int(plan_bytes), "launch": dplans[0].launch_info()}
if stamps:
    res["stamps_rotated"] = stamp_summary(dplans[(a.windows * a.window - 1) % P])
warm = run(lambda t: 0, 2)
res["warm_us"] = warm
if stamps:
    res["stamps_warm"] = stamp_summary(dplans[0])
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("HSRANS_") and k != "HSRANS_LIB"}
if cal:
    res["calibration"] = cal["class_weights"] if isinstance(cal, dict) and "class_weights" in cal else cal
for k in range(P):  # still bit-exact after everything
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    assert a.no_check or (ctx.status(dplans[k]) == 0 and np.array_equal(d_out[k].cpu().numpy(), data)), "not bit-exact after the timed loops"
print(json.dumps(res), flush=True)
