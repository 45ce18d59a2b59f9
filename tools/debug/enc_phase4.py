"""HSRANS_DEBUG_STAMPS=1: phase stamps of the mt_ GPU encoder for 4 and 1526 blocks of 64 KiB (debug aid; results may be wrong under diagnostic builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
for n in (4 << 16, d.size):
    d_out = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        try:
            ctx.encode_device(H.MT, 64, 11, d_in[:n], d_out, block_size=1 << 16)
        except Exception as e:
            print("encode failed:", e, file=sys.stderr)
