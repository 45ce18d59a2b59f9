"""Sustained decodes of a 100 MB mt_ stream (64 KiB blocks, device-built plan of the given interval): the target of a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
ctx.calibrate()
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
interval = int(sys.argv[1]) if len(sys.argv) > 1 else 64
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
d_out = torch.empty(H.capacity(H.MT, 64, d.size), dtype=torch.uint8, device="cuda")
n, dplan = ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=block, index_interval=interval, want_plan=True)
back = torch.empty(d.size, dtype=torch.uint8, device="cuda")
for _ in range(600):
    ctx.decode_device(dplan, d_out, back, stream_length=n)
torch.cuda.synchronize()
print(ctx.status(dplan), bool(torch.equal(back, d_in)))
