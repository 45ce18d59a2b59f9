"""Decode time of a 100 MB mt_ stream of 64 KiB blocks against the index interval of its device-built plan (debug aid)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
ctx.calibrate()
d = synth.enwik8_shaped(100_000_000, seed=1)
d_in = torch.from_numpy(d).cuda()
block = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
for states in (64,):
    for interval in (16, 32, 64, 128, 192, 256, 512, 1024):
        d_out = torch.empty(H.capacity(H.MT, states, d.size), dtype=torch.uint8, device="cuda")
        n, dplan = ctx.encode_device(H.MT, states, 11, d_in, d_out, block_size=block, index_interval=interval, want_plan=True)
        back = torch.empty(d.size, dtype=torch.uint8, device="cuda")
        for _ in range(30):
            ctx.decode_device(dplan, d_out, back, stream_length=n)
        torch.cuda.synchronize()
        ok = ctx.status(dplan) == 0 and bool(torch.equal(back, d_in))
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                ctx.decode_device(dplan, d_out, back, stream_length=n)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 40)
        ms = sorted(ts)[len(ts) // 2]
        print(json.dumps({"block": block, "states": states, "interval": interval, "decode_us": round(ms * 1e3, 2), "frac_of_8TBs": round((d.size + n) / (ms * 1e-3) / 8e12, 3), "ok": ok}), flush=True)
