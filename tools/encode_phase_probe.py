"""Where the mt_ GPU encoder's time goes, by number of blocks in flight: hsrans_encode_device with HSRANS_DEBUG_STAMPS=1 (the
library prints the mean of the per-block phase stamps) on prefixes of the enwik8-shaped input.  Run on the GPU box:
HSRANS_DEBUG_STAMPS=1 python tools/encode_phase_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ctx = H.Context(0)
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
for block in (1 << 16, 1 << 15, 1 << 18):
    for blocks in (4, 64, 256, 512, 1024, 1536, 2048, 3072, 6144):
        n = min(blocks * block, d.size)
        d_out = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
        print(f"block {block} blocks {n // block}", file=sys.stderr, flush=True)
        for _ in range(3):
            ctx.encode_device(H.MT, 64, 11, d_in[:n], d_out, block_size=block)
        if n == d.size:
            break
