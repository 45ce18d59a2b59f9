"""A few mt_ GPU encodes WITH the device-built plan (100 MB, 64 KiB blocks, G = 32): the target of a kernel trace; prints host times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
d_out = torch.empty(H.capacity(H.MT, 64, d.size), dtype=torch.uint8, device="cuda")
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    n, dplan = ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=1 << 16, index_interval=32, want_plan=True)
    ts.append(time.perf_counter() - t0)
    del dplan
torch.cuda.synchronize()
print("host ms per encode with plan:", [round(t * 1e3, 3) for t in ts])
