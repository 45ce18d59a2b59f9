"""mt_ / block_ with a sidecar index at the wide histograms (grouped launch): which table, how fast.  Run on the GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
n = 100_000_000
data = synth.enwik8_shaped(n)
d_ref = torch.from_numpy(data).cuda()
for container in (H.MT, H.BLOCK):
    for bits in [int(b) for b in os.environ.get("BITS", "12,13,14,15").split(",")]:
        s, plan = H.encode(container, 64, bits, data, index_interval=32, block_size=1 << 18)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dp = ctx.make_device_plan(plan)
        ctx.decode_device(dp, d_in, out, stream_length=s.size)
        ok = bool(torch.equal(out, d_ref)) and ctx.status(dp) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ctx.decode_device(dp, d_in, out, stream_length=s.size)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        info = dp.launch_info()
        print(json.dumps({"container": container, "bits": bits, "ms": round(ms, 4), "frac": round((s.size + n) / (ms * 1e-3) / 8e12, 4), "ok": ok,
                          "mode": info["table_mode"], "grid": info["grid"], "block": info["block"], "lds": info["lds_bytes"]}), flush=True)
