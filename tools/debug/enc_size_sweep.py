"""mt_ GPU encoder rate against input size (64 KiB blocks, 11 bits): hsrans_encode_device, best and mean of 8 calls. Debug aid."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
base = synth.enwik8_shaped(1 << 27, seed=1)
for size in (1 << 20, 1 << 23, 100_000_000, 1 << 28, 1 << 30):
    d = np.tile(base, (size + base.size - 1) // base.size)[:size]
    d_in = torch.from_numpy(d).cuda()
    d_out = torch.empty(H.capacity(H.MT, 64, size), dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=1 << 16)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=1 << 16); ts.append(time.perf_counter() - t0)
    print(json.dumps({"size": size, "blocks": (size + 65535) >> 16, "stream": n, "ms_best": round(min(ts) * 1e3, 3), "ms_mean": round(sum(ts) / len(ts) * 1e3, 3), "GB_s_best": round(size / min(ts) / 1e9, 1)}), flush=True)
    del d_in, d_out
    torch.cuda.empty_cache()
