#!/usr/bin/env python3
"""Tunes the per-class run lengths (HSRANS_SLOT_WEIGHTS / HSRANS_SLOT_WEIGHTS4) of the one-chain-per-wave launch so that all
eight wave classes finish together: runs tools/stamps.py in a child process per iteration (the library reads the weights once
per process), reads the mean finish time per class and moves every weight towards length * (mean finish / class finish).

    python tools/tune_weights.py [--bits 11] [--iters 6] [--start w0,...,w7]
"""
import argparse, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--start", default="")
ap.add_argument("--var", default="HSRANS_DIRECT_WEIGHTS")
ap.add_argument("--states", type=int, default=64)
ap.add_argument("--damp", type=float, default=0.8)
ap.add_argument("--index", default="wave", help="stamps.py --index: wave (one chain per wave) or a checkpoint interval in groups (uniform plans: HSRANS_SLOT_WEIGHTS)")
ap.add_argument("--cold", type=int, default=0, help="tune with this many stream/output pairs rotated (stamps.py --cold)")
a = ap.parse_args()
w = [float(x) for x in a.start.split(",")] if a.start else [1241, 1204, 1160, 1100, 974, 886, 774, 660]
for it in range(a.iters):
    env = dict(os.environ)
    env[a.var] = ",".join(str(int(round(x))) for x in w)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamps.py"), "--index", a.index, "--bits", str(a.bits), "--states", str(a.states)] + (["--cold", str(a.cold)] if a.cold > 1 else []), env=env, capture_output=True, text=True).stdout
    # the library must have read the variable this run varies (a misspelt name tuned nothing): the launch reports the lengths it was shaped with
    used = re.search(r"'class_weights': \[([\d, ]+)\]", out)
    assert used and [int(x) for x in used.group(1).split(",")] == [int(round(x)) for x in w], f"{a.var} was not picked up by the launch: {used.group(1) if used else out[-400:]}"
    m = re.search(r"^static done by wave.*?: ([\d. ]+)\| second half: ([\d. ]+)$", out, re.M)
    done = re.search(r"^done\s+min.*max\s+([\d.]+) us", out, re.M)
    t = [float(x) for x in (m.group(1) + " " + m.group(2)).split()]
    mean = sum(t) / len(t)
    print(json.dumps({"iter": it, "weights": [int(round(x)) for x in w], "static_done_by_class_us": t, "spread_us": max(t) - min(t), "last_wave_done_us": float(done.group(1))}), flush=True)
    w = [wi * (mean / ti) ** a.damp for wi, ti in zip(w, t)]
    s = sum(w)
    w = [wi * 8000 / s for wi in w]
print("next:", ",".join(str(int(round(x))) for x in w))
