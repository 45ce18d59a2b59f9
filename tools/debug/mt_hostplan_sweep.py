"""Sustained decode time of a 100 MB mt_ stream (64 KiB blocks) with the HOST encoder's sidecar plan (one group per block, no parts),
for A/B runs of launch-shape knobs (HSRANS_WAVES_PER_WG=...).  Debug aid."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
ctx = H.Context(0)
try:
    ctx.calibrate()
except H.HsransError:
    pass  # (a launch-shape override the calibration launch does not take)
d = synth.enwik8_shaped(100_000_000, seed=1)
block = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
for interval in (32, 64, 128):
    s, plan = H.encode(H.MT, 64, 11, d, block_size=block, index_interval=interval, independent_blocks=True)
    streams = [torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda() for _ in range(4)]
    outs = [torch.empty(d.size, dtype=torch.uint8, device="cuda") for _ in range(4)]
    dp = ctx.make_device_plan(plan)
    k = [0]
    def rot():
        i = k[0] % 4; k[0] += 1
        ctx.decode_device(dp, streams[i], outs[i], stream_length=s.size)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        for _ in range(20): rot()
        torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): rot()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 40 * 1e3)
    ok = all(bool(torch.equal(o.cpu(), torch.from_numpy(d))) for o in outs[:1])
    info = dp.launch_info()
    print(json.dumps({"interval": interval, "rotated_us": round(sorted(ts)[2], 2), "grid": info["grid"], "block": info["block"], "lds": info["lds_bytes"], "waves": info["waves_per_block"], "ok": ok}), flush=True)
