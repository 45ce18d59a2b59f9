#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer drop-in entry (hsrans_decode_host: H2D + plan + launch + D2H, pageable memory).
Never the benchmark's `value`; quoted in DESIGN.md."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

n = 100_000_000
data = synth.enwik8_shaped(n)
ctx = H.Context(0)
for name, container, kw in (("raw + plan G=32", H.RAW, dict(index_interval=32)), ("raw, no plan", H.RAW, {}), ("mt_ (reference policy)", H.MT, {}),
                            ("mt_ + plan G=32", H.MT, dict(index_interval=32))):
    enc = H.encode(container, 64, 11, data, **kw)
    s, plan = enc if isinstance(enc, tuple) else (enc, None)
    best = 1e9
    for _ in range(1 if name == "raw, no plan" else 4):
        t0 = time.perf_counter()
        r, out = ctx.decode_host(container, 64, 11, s, n, plan=plan)
        best = min(best, time.perf_counter() - t0)
    assert r == n and np.array_equal(out, data)
    print(f"{name:24s} {best * 1e3:8.2f} ms  {n / 2**20 / best:10.0f} MiB/s (PCIe-inclusive, pageable host buffers)")
