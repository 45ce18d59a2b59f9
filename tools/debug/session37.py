"""Unindexed mt_ streams (one chain per block, private tables) by histogram width.  Run on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
n = 100_000_000
data = synth.enwik8_shaped(n)
d_ref = torch.from_numpy(data).cuda()
for S in (64, 32):
    for bits in (11, 12, 13, 14, 15):
        s = H.encode(H.MT, S, bits, data, block_size=1 << 16)
        plan = H.plan_build(H.MT, S, bits, s)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dp = ctx.make_device_plan(plan)
        ctx.decode_device(dp, d_in, out, stream_length=s.size)
        ok = bool(torch.equal(out, d_ref)) and ctx.status(dp) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ctx.decode_device(dp, d_in, out, stream_length=s.size)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        info = dp.launch_info()
        print(json.dumps({"states": S, "bits": bits, "ms": round(ms, 4), "frac": round((s.size + n) / (ms * 1e-3) / 8e12, 4), "ok": ok, "mode": info["table_mode"],
                          "grid": info["grid"], "block": info["block"], "lds": info["lds_bytes"]}), flush=True)
