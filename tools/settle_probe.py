#!/usr/bin/env python3
"""Diagnostic: how does the decode time move with what the GPU did just before?  (tools/pair_probe.py saw one replayed pair go
from 44 us to 32.4 us over a few thousand back-to-back launches.)

Windows of `--window` launches, no idle time between them; per window: us per launch (HIP events), and — with the diagnostic
library (HSRANS_DEBUG_STAMPS=1) — the shader clock the last launch's waves saw (s_memtime / s_memrealtime).  Phases:
replay one pair; rotate P pairs; replay again; idle; replay again; rotate again.  Also samples the GPU's hwmon files if readable.

    python tools/settle_probe.py [--window 200] [--windows 15]
"""
import argparse
import ctypes
import glob
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
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--window", type=int, default=200)
ap.add_argument("--windows", type=int, default=15)
ap.add_argument("--pairs", type=int, default=4)
a = ap.parse_args()
n, S, bits, P = a.size, 64, a.bits, a.pairs
stamps = bool(os.environ.get("HSRANS_DEBUG_STAMPS"))
ctx = H.Context(0)
groups = H.index_boundaries(S, bits, n, ctx)
base = synth.enwik8_shaped(n, seed=20241008)
dplans, d_in, d_out, lens = [], [], [], []
for k in range(P):
    data = base if k == 0 else synth._permutation(1000 + k)[base]
    s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
    lens.append(s.size)
    dplans.append(ctx.make_device_plan(p))
    d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
    d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
L = H.load_library()
if stamps:
    L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
    L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]


def hwmon():
    out = {}
    for pat, key in (("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_W"), ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "power_in_W"),
                     ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "sclk_MHz"), ("/sys/class/drm/card*/device/hwmon/hwmon*/freq2_input", "mclk_MHz"),
                     ("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "temp_C")):
        for f in glob.glob(pat)[:1]:
            try:
                v = float(open(f).read().strip())
                out[key] = round(v / 1e6, 1) if "power" in key or "clk" in key else round(v / 1e3, 1)
            except (OSError, ValueError):
                pass
    return out


def clock_of(dp):
    if not stamps:
        return None
    buf = np.zeros(16384 * 8, np.uint64)
    L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 8).astype(np.int64)
    st = st[st[:, 3] > 0]
    ghz = st[:, 5] / ((st[:, 3] - st[:, 0]) * 10.0)
    rel = (st - st[:, 0].min()) / 100.0
    return {"GHz_p50": round(float(np.median(ghz)), 3), "ready_p50": round(float(np.median(rel[:, 2])), 2), "done_p50": round(float(np.median(rel[:, 3])), 2),
            "done_p90": round(float(np.percentile(rel[:, 3], 90)), 2), "done_max": round(float(rel[:, 3].max()), 2)}


def phase(name, pick, windows=a.windows):
    rows = []
    for w in range(windows):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for t in range(a.window):
            k = pick(w * a.window + t)
            ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
        eb.record()
        torch.cuda.synchronize()
        row = {"us": round(ea.elapsed_time(eb) / a.window * 1e3, 2)}
        row.update(hwmon())
        c = clock_of(dplans[pick(w * a.window + a.window - 1)])
        if c:
            row.update(c)
        rows.append(row)
    print(json.dumps({"phase": name, "windows": rows}), flush=True)


torch.cuda.synchronize()
time.sleep(0.3)
phase("replay pair 0 (after 0.3 s idle)", lambda t: 0)
phase("rotate", lambda t: t % P)
phase("replay pair 1", lambda t: 1)
time.sleep(0.3)
phase("replay pair 1 (after 0.3 s idle)", lambda t: 1, windows=6)
phase("rotate again", lambda t: t % P, windows=6)
phase("rotate 2 pairs", lambda t: t % 2, windows=6)
phase("rotate 3 pairs", lambda t: t % 3, windows=6)
