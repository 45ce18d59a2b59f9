"""How long the device-side mt_ header walk (K2, hsrans_dplan_create_from_device_stream) takes.  Run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H

ctx = H.Context(0)
for n, block in ((100_000_000, 1 << 16), (1 << 30, 1 << 16), (1 << 30, 1 << 18)):
    g = torch.Generator(device="cuda").manual_seed(3)
    d_in = torch.rand(n, device="cuda", generator=g).pow_(6).mul_(205).to(torch.uint8)
    d_out = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=block)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dp = ctx.make_device_plan_from_stream(H.MT, 64, 11, d_out, m, n)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"size": n, "block": block, "blocks": (n + block - 1) // block, "k2_ms_best": round(min(ts) * 1e3, 3)}), flush=True)
    del d_in, d_out, dp
