#!/usr/bin/env python3
"""Replays one case of tests/test_gpu_parity.py::test_random_sweep_decode (same RNG stream) and reports where the GPU output
differs from the oracle, through the host entry and through a device plan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth, api
from oracle_lib import Oracle, RAW
from test_gpu_parity import _random_case

want_case = int(sys.argv[1]) if len(sys.argv) > 1 else 4
zipf = synth.enwik8_shaped(1 << 20, seed=11)
nonstat = synth.nonstationary(3_000_000)
rng = np.random.default_rng(20241008)
ctx = H.Context(0)
orc = Oracle()
for case in range(want_case + 1):
    container = int(rng.integers(0, 3)); states = int(rng.choice((32, 64))); bits = int(rng.integers(10, 16))
    d = _random_case(rng, zipf, nonstat); n = d.size
    interval = int(rng.choice((0, 4, 8, 32, 100, 1024)))
    block = int(rng.choice((0, 32768, 65536))) if container != RAW else 0
print("case", case, container, states, bits, n, interval, block, "unique", np.unique(d).size)
s, plan = H.encode(container, states, bits, d, index_interval=interval, block_size=block)
hdr, cf, pc = api.plan_tables(plan)
print(hdr, pc["steps"], pc["tail"])
r0, want = orc.decode(container, states, bits, s, n)
r, got = ctx.decode_host(container, states, bits, s, n, plan=plan)
bad = np.nonzero(got != want)[0]
print("host entry: r", r, "mismatches", bad.size, "first", bad[:5], "last", bad[-5:] if bad.size else None, "groups", np.unique(bad // states)[:20] if bad.size else None)
dp = ctx.make_device_plan(plan)
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
for rep in range(3):
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
    torch.cuda.synchronize()
    got2 = d_out.cpu().numpy()
    bad = np.nonzero(got2 != want)[0]
    print("device plan rep", rep, "status", ctx.status(dp), "mismatches", bad.size, "first", bad[:5], "info", dp.launch_info())
