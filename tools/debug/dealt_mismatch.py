#!/usr/bin/env python3
"""where does a dealt launch differ from the source? (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
os.environ["HSRANS_DEALT_MIN_CHAINS"] = "1"
ctx = H.Context(0)
if os.environ.get("CAL"):
    print(ctx.calibrate()["class_weights"])
    for c in (2, 3, 5):
        print(ctx.calibrate_runs(copies=c)["class_weights"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = synth.enwik8_shaped(n, seed=1)
d_in = torch.from_numpy(d).cuda()
enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
m, dplan = ctx.encode_device(H.MT, 64, 11, d_in, enc, block_size=1 << 18, index_interval=G, want_plan=True)
out = torch.zeros(n, dtype=torch.uint8, device="cuda")
for rep in range(3):
    out.zero_()
    ctx.decode_device(dplan, enc, out, stream_length=m)
    torch.cuda.synchronize()
    bad = (out != d_in).nonzero().flatten().cpu().numpy()
    print(dplan.launch_info()["class_weights"]); print("rep", rep, "status", ctx.status(dplan), "launch", dplan.launch_info()["spread"], "mismatching bytes", bad.size)
    if bad.size:
        # runs of mismatches in units of chains (G groups of 64 bytes)
        ch = np.unique(bad // (64 * G))
        print("  chains with mismatches:", ch[:40], "... total", ch.size)
        print("  first bad byte", bad[0], "= chain", bad[0] // (64 * G), "group in chain", (bad[0] // 64) % G, "; block", bad[0] >> 18)
        gaps = np.diff(ch)
        print("  distinct gaps between bad chains:", np.unique(gaps)[:20])
