#!/usr/bin/env python3
"""Where a wave of the grouped launch (mt_ / block_ plans with checkpoints: one workgroup per block) spends its time, summed
over its rounds: barrier wait, table build, plan records + first chunks, decode (HSRANS_DEBUG_STAMPS=1; run_grouped).

    python tools/stamps_grouped.py [--size BYTES] [--block BYTES] [--interval G]
"""
import argparse, ctypes, os, sys
os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--block", type=int, default=1 << 18)
ap.add_argument("--interval", type=int, default=256)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--weights", default="", help="4 per-mille chain lengths by wave class (wave/4 in its workgroup): 16 checkpoints per block at weighted positions instead of --interval")
a = ap.parse_args()
base = synth.enwik8_shaped(1 << 26)
data = np.tile(base, (a.size + base.size - 1) // base.size)[: a.size]
ctx = H.Context(0)
if a.weights:
    w = np.repeat(np.array([float(x) for x in a.weights.split(",")]), 4)
    gpb = a.block // 64  # groups per block
    cum = np.cumsum(w) / w.sum()
    inner = (np.round(cum[:-1] * gpb / 4) * 4).astype(np.uint64)  # 15 cuts inside a block, multiples of 4 groups
    nblocks = (a.size + a.block - 1) // a.block
    cuts = (np.arange(nblocks, dtype=np.uint64)[:, None] * np.uint64(gpb) + inner[None, :]).ravel()
    cuts = cuts[cuts < np.uint64((a.size - 63) // 64)]
    s, plan = H.encode(H.MT, 64, a.bits, data, block_size=a.block, index_groups=cuts, independent_blocks=True)
else:
    s, plan = H.encode(H.MT, 64, a.bits, data, block_size=a.block, index_interval=a.interval, independent_blocks=True)
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
d_out = torch.zeros(a.size, dtype=torch.uint8, device="cuda")
dp = ctx.make_device_plan(plan)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.decode_device(dp, d_in, d_out, stream_length=s.size); e1.record(); torch.cuda.synchronize()
assert torch.equal(d_out.cpu(), torch.from_numpy(data))
info = dp.launch_info()
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(16384 * 8, np.uint64)
L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
st = buf.reshape(-1, 8).astype(np.int64)
st = st[st[:, 5] > 0]
t0 = st[:, 0].min()
us = lambda col: st[:, col] / 100.0
life = (st[:, 6] - st[:, 0]) / 100.0
print(f"launch {info}; kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events; {len(st)} waves with work, rounds per wave {st[:, 5].min()}..{st[:, 5].max()}")
print(f"per wave, totals over its rounds (us): lifetime p50 {np.median(life):.1f} max {life.max():.1f} | barrier wait p50 {np.median(us(1)):.1f} | table build p50 {np.median(us(2)):.1f} | "
      f"records + first chunks p50 {np.median(us(3)):.1f} | decode p50 {np.median(us(4)):.1f}")
r = st[:, 5].astype(np.float64)
print(f"per round (us): wait {np.median(us(1) / r):.2f}  build {np.median(us(2) / r):.2f}  records+chunks {np.median(us(3) / r):.2f}  decode {np.median(us(4) / r):.2f}  sum {np.median((us(1) + us(2) + us(3) + us(4)) / r):.2f}")
w = np.arange(len(buf) // 8)[: len(st)] % info["waves_per_block"]
for name, col in (("wait", 1), ("decode", 4)):
    print(f"{name} per round by wave in workgroup:", " ".join(f"{np.median((us(col) / r)[w == k]):.1f}" for k in range(info["waves_per_block"])))
# by scheduling class (grid half x wave quarter): when the waves enter, what each phase costs them, when they are done (us since the launch's first wave)
wpb = info["waves_per_block"]
idx = np.arange(len(buf) // 8)
keep = buf.reshape(-1, 8).astype(np.int64)[:, 5] > 0
wg, wv = (idx // wpb)[keep], (idx % wpb)[keep]
half = (wg >= (info["grid"] + 1) // 2).astype(int)
cls = half * 4 + np.minimum(wv // max(wpb // 4, 1), 3)
print("class: waves | entry p50 | wait | build | records+chunks | decode | done p50 / max   (us; sums over the wave's rounds)")
for k in range(8):
    m = cls == k
    if m.any():
        print(f"  {k}: {m.sum():5d} | {np.median((st[m, 0] - t0) / 100.0):6.1f} | {np.median(us(1)[m]):5.1f} | {np.median(us(2)[m]):5.1f} | {np.median(us(3)[m]):5.1f} | {np.median(us(4)[m]):5.1f} | "
              f"{np.median((st[m, 6] - t0) / 100.0):6.1f} / {((st[m, 6] - t0) / 100.0).max():6.1f}")
