#!/usr/bin/env python3
"""A/B of library builds / tuning environments IN ONE PROCESS on the SAME device buffers (between processes the rotated 100 MB
decode moves by +-3 us on one box, with no change at all: where the driver puts the pages decides; tools/rot_probe.py saw 37.5 and
42.9 us for one and the same library).  Every variant is its own dlopen of a library file (its tuning environment is read when
it loads) with its own context and device plans; the stream bytes do not depend on the index, so the streams and the output
buffers are shared.  Windows of `--window` launches alternate between the variants, `--rounds` times.

    python tools/ab_probe.py --variant base --variant other:lib/variants/libhsrans_hip_other.so:HSRANS_DIRECT_WEIGHTS=1400,1330,1240,1130,960,810,640,490
    python tools/ab_probe.py --container mt --block 262144 --index 32 --variant spread --variant grouped::HSRANS_SPREAD=0
"""
import argparse
import ctypes
import json
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--variant", action="append", required=True, help="tag[:library path (relative to hypersonic_rans_amd/, or absolute)[:ENV=VAL,ENV=VAL...]]")
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--states", type=int, default=64)
ap.add_argument("--container", choices=("raw", "mt", "block"), default="raw")
ap.add_argument("--index", default="wave", help="raw: 'wave' (one chain per resident wavefront) or a checkpoint interval in groups; mt/block: the interval")
ap.add_argument("--block", type=int, default=1 << 18, help="mt/block: block size in bytes")
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--window", type=int, default=200)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--no-check", default="", help="comma-separated tags of diagnostic variants whose output is not the decoded data")
ap.add_argument("--calibrate", action="store_true")
a = ap.parse_args()
n, S, bits, P = a.size, a.states, a.bits, a.pairs
container = {"raw": H.RAW, "mt": H.MT, "block": H.BLOCK}[a.container]
tile = min(n, 100_000_000)
cache = f"/tmp/zipf_{tile}_20241008.bin"
if os.path.exists(cache):
    base = np.fromfile(cache, np.uint8)
else:
    base = synth.enwik8_shaped(tile, seed=20241008)
    base.tofile(cache)
if n > tile:  # bigger workloads: tiles of the base block under byte permutations (as bench.py's _tiled)
    big = np.empty(n, np.uint8)
    for t, o in enumerate(range(0, n, tile)):
        c = min(tile, n - o)
        big[o:o + c] = (base if t % 7 == 0 else synth._permutation(1000 + t % 7)[base])[:c]
    base = big
datas = [base if k == 0 else synth._permutation(2000 + k)[base] for k in range(P)]
no_check = set(a.no_check.split(",")) if a.no_check else set()
tmpdir = tempfile.mkdtemp(prefix="hsrans_ab_")

variants = []
d_in, d_out, lens = [], [], []


def env_changes_index(env):
    return any(k.startswith(("HSRANS_DIRECT", "HSRANS_DUAL", "HSRANS_SLOT")) for k in env)


for spec in a.variant:
    parts = spec.split(":", 2)
    tag = parts[0]
    lib = parts[1] if len(parts) > 1 and parts[1] else "lib/libhsrans_hip.so"
    lib = lib if os.path.isabs(lib) else os.path.join(ROOT, "hypersonic_rans_amd", lib)
    env = dict(kv.split("=", 1) for kv in re.split(r",(?=[A-Z][A-Z0-9_]*=)", parts[2])) if len(parts) > 2 and parts[2] else {}  # (values may hold commas: weight lists)
    # a private copy of the file: dlopen of one path twice would give the SAME library instance (one set of tuning globals)
    private = os.path.join(tmpdir, f"{tag}.so")
    shutil.copy(lib, private)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["HSRANS_LIB"] = private
    api._LIB = None
    ctx = H.Context(0)  # loads `private` and reads the tuning environment
    cal = ctx.calibrate(bits=bits) if a.calibrate else None
    groups = H.index_boundaries(S, bits, n, ctx) if (a.container == "raw" and a.index == "wave") else None
    dplans = []
    for k in range(P):
        if len(d_in) > k and hasattr(a, "_streams") and not env_changes_index(env):
            s, p = a._streams[k]  # the host encoders take seconds per GiB: encode once per pair when the index cannot differ
        elif groups is not None:
            s, p = H.encode(container, S, bits, datas[k], index_groups=groups)
        elif a.container == "raw":
            s, p = H.encode(container, S, bits, datas[k], index_interval=int(a.index))
        else:
            s, p = H.encode(container, S, bits, datas[k], block_size=a.block, index_interval=int(a.index) if a.index != "wave" else 256)
        if not hasattr(a, "_streams"):
            a._streams = {}
        a._streams.setdefault(k, (s, p))
        if len(d_in) <= k:
            lens.append(s.size)
            d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
            d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
        else:
            assert s.size == lens[k]
        dplans.append(ctx.make_device_plan(p))
        d_out[k].zero_()
        ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])
        torch.cuda.synchronize()
        assert ctx.status(dplans[k]) == 0
        assert tag in no_check or np.array_equal(d_out[k].cpu().numpy(), datas[k]), f"{tag}: not bit-exact"
    variants.append({"tag": tag, "lib": os.path.relpath(lib, ROOT), "env": env, "ctx": ctx, "dplans": dplans, "chains": H.plan_chain_count(p), "plan_bytes": int(p.size),
                     "launch": dplans[0].launch_info(), "rot": [], "warm": [], "calibration": cal["class_weights"] if cal else None})
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def window(v, pick):
    saved = {k: os.environ.get(k) for k in v["env"]}  # (some knobs are read at every launch: the variant's environment is in force while it runs)
    os.environ.update(v["env"])
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for t in range(a.window):
        k = pick(t)
        v["ctx"].decode_device(v["dplans"][k], d_in[k], d_out[k], stream_length=lens[k])
    eb.record()
    torch.cuda.synchronize()
    for k, val in saved.items():
        if val is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = val
    return round(ea.elapsed_time(eb) / a.window * 1e3, 2)


for v in variants:  # settle
    window(v, lambda t: t % P)
for r in range(a.rounds):
    for v in variants:
        v["rot"].append(window(v, lambda t: t % P))
    for v in variants:
        v["warm"].append(window(v, lambda t: r % P))
for v in variants:  # whatever the launches raced about (tail stealing), the bytes must still be right: every pair once more, from a cleared output
    if v["tag"] in no_check:
        continue
    for k in range(P):
        d_out[k].zero_()
        for _ in range(3):
            v["ctx"].decode_device(v["dplans"][k], d_in[k], d_out[k], stream_length=lens[k])
        torch.cuda.synchronize()
        assert v["ctx"].status(v["dplans"][k]) == 0
        assert np.array_equal(d_out[k].cpu().numpy(), datas[k]), f"{v['tag']}: not bit-exact after the timed loops (pair {k})"
for v in variants:
    print(json.dumps({"tag": v["tag"], "rotated_us_median": float(np.median(v["rot"])), "warm_us_median": float(np.median(v["warm"])), "rotated_us": v["rot"], "warm_us": v["warm"],
                      "chains": v["chains"], "plan_bytes": v["plan_bytes"], "lib": v["lib"], "env": v["env"], "launch": v["launch"], "calibration": v["calibration"]}), flush=True)
shutil.rmtree(tmpdir, ignore_errors=True)
