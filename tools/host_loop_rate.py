#!/usr/bin/env python3
"""A caller that loops the plain decodeFunc over one file (src/main.cpp:860-889 does exactly this): hsrans_decode_host without a plan,
wall clock per call incl. both PCIe legs, with and without the index the first call leaves behind (HSRANS_HOST_INDEX_CACHE_OFF).
Run on the GPU box: python tools/host_loop_rate.py [size]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = synth.enwik8_shaped(n)
for container, name in ((H.MT, "mt_"), (H.RAW, "raw")):
    stream = np.ascontiguousarray(H.encode(container, 64, 11, data))
    for cache in (True, False):
        if cache:
            os.environ.pop("HSRANS_HOST_INDEX_CACHE_OFF", None)
        else:
            os.environ["HSRANS_HOST_INDEX_CACHE_OFF"] = "1"
        ctx = H.Context(0)
        out = np.empty(n, np.uint8)
        L = ctx.L
        ts = []
        for k in range(6 if (cache or container == H.MT) else 2):
            t0 = time.perf_counter()
            r = L.hsrans_decode_host(ctx.handle, container, 64, 11, stream.ctypes.data, stream.size, out.ctypes.data, n, None, 0)
            ts.append(time.perf_counter() - t0)
            assert r == n and np.array_equal(out, data)
        print(json.dumps({"codec": f"{name} rANS32x64 16w 11", "size": n, "stream": int(stream.size), "index_cache": cache, "calls_ms": [round(t * 1e3, 2) for t in ts],
                          "later_calls_ms_best": round(min(ts[1:]) * 1e3, 2), "index_chains": ctx.host_index_chains(),
                          "note": "pageable host buffers, both PCIe legs and the host-side header walk inside every call"}), flush=True)
        del ctx
