"""The library's host SIMD decoder (hsrans_decode_cpu, what `*_decode_auto_N` uses for single-chain streams) against the REAL reference's
fastest decoders (oracle/_ref: AVX2 xmmShfl2 varC/varA, AVX-512 ymmShfl2 varC/varA) on one core, same stream, same process.
"Never regresses"  means every row has ratio >= 1.   python tools/host_decoder_vs_reference.py [--size N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, synth
from oracle_lib import RAW, Ref

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=32_000_000)
ap.add_argument("--budget", type=float, default=2.0)
ap.add_argument("--cases", default="32:11,32:14,64:11,64:14")
a = ap.parse_args()
data = synth.enwik8_shaped(a.size, seed=20241008)
ref = Ref()
L = api.load_library()
for case in a.cases.split(","):
    S, bits = (int(v) for v in case.split(":"))
    stream = H.encode(H.RAW, S, bits, data)
    row = {"states": S, "bits": bits, "bytes": a.size, "cpu_level": api.CPU_LEVELS[api.cpu_level()]}
    for variant, name in ((1, "reference_avx2"), (3, "reference_avx512")):
        if variant == 3 and not ref.has_avx512():
            continue
        try:
            best, r, out, runs = ref.timed_decode(RAW, S, bits, stream, a.size, variant=variant, budget_s=a.budget)
        except Exception:  # noqa: BLE001
            continue
        if r == a.size and np.array_equal(out, data):
            row[name + "_MiB_s"] = round(a.size / 2**20 / best, 1)
    out = np.full(a.size + 64, 0xCC, np.uint8)
    for level in range(api.cpu_level() + 1):
        best, t_total, runs = None, 0.0, 0
        while runs < 3 or (t_total < a.budget and runs < 40):
            t0 = time.perf_counter()
            r = L.hsrans_decode_cpu(level, 1, H.RAW, S, bits, api._p(stream), stream.size, api._p(out), a.size, None, 0)
            dt = time.perf_counter() - t0
            assert r == a.size
            best = dt if best is None else min(best, dt)
            t_total += dt
            runs += 1
        assert np.array_equal(out[:a.size], data)
        row["own_" + api.CPU_LEVELS[level] + "_MiB_s"] = round(a.size / 2**20 / best, 1)
    refs = [v for k, v in row.items() if k.startswith("reference_")]
    own = row["own_" + api.CPU_LEVELS[api.cpu_level()] + "_MiB_s"]
    row["auto_over_best_reference"] = round(own / max(refs), 3) if refs else None
    print(json.dumps(row), flush=True)
