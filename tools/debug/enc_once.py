"""A few mt_ GPU encodes of the 100 MB enwik8-shaped input (64 KiB blocks), nothing else: the target of a kernel trace (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
d_out = torch.empty(H.capacity(H.MT, 64, d.size), dtype=torch.uint8, device="cuda")
block = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
for _ in range(10):
    ctx.encode_device(H.MT, 64, 11, d_in, d_out, block_size=block)
torch.cuda.synchronize()
