"""Where the GPU mt_ encoder's stream differs from the host encoder's (debug aid). Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
z = synth.zipf_bytes(400_000, 1.1, seed=3)
for states in (64, 32):
    for n, block in ((64 * 4, 4096), (64 * 5, 4096), (64 * 8, 4096), (64 * 12, 4096), (64*13, 4096), (1696, 4096), (4096, 4096), (100_000, 4096), (65536, 65536), (300_001, 32768)):
        d = z[:n]
        want = H.encode(H.MT, states, 11, d, block_size=block, independent_blocks=True)
        d_in = torch.from_numpy(d).cuda()
        d_out = torch.full((H.capacity(H.MT, states, n),), 0xA5, dtype=torch.uint8, device="cuda")
        m = ctx.encode_device(H.MT, states, 11, d_in, d_out, block_size=block)
        got = d_out[:m].cpu().numpy()
        if got.size != want.size:
            print(states, n, block, "SIZE", got.size, want.size)
            continue
        bad = np.nonzero(got != want)[0]
        print(states, n, block, "ok" if bad.size == 0 else f"{bad.size} bytes differ, first {bad[:8]}, last {bad[-4:]}, of {got.size}")
