#!/usr/bin/env python3
"""Per-wave timeline of the host-dealt one-round launch (k_decode_dealt; HSRANS_DEBUG_STAMPS=1, diagnostic library): by scheduling
class (grid half x wave quarter) when the waves enter, have their table, have states + first chunks, and are done — which class
ends the launch says which way the class weights are off; the spread inside a class is the dealing's rounding to whole chains.

    python tools/stamps_dealt.py [--size BYTES] [--block BYTES] [--interval G] [--calibrate]"""
import argparse, ctypes, os, sys
os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 27)
ap.add_argument("--block", type=int, default=1 << 18)
ap.add_argument("--interval", type=int, default=32)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--calibrate", action="store_true")
a = ap.parse_args()
ctx = H.Context(0)
if a.calibrate:
    ctx.calibrate()
    for c in (2, 3, 5):
        ctx.calibrate_runs(copies=c)
d_in = torch.from_numpy(synth.enwik8_shaped(a.size, seed=20241008)).cuda()
enc = torch.empty(H.capacity(H.MT, 64, a.size), dtype=torch.uint8, device="cuda")
m, dp = ctx.encode_device(H.MT, 64, a.bits, d_in, enc, block_size=a.block, index_interval=a.interval, want_plan=True)
plan = ctx.read_device_plan(dp, capacity=1 << 30)
dp = ctx.make_device_plan(plan)  # (a host-made device plan: the diagnostic buffers hang off hsrans_dplan_create)
outs = [torch.zeros(a.size, dtype=torch.uint8, device="cuda") for _ in range(3)]
for i in range(12):  # sustained, rotated outputs; the stamps are the last launch's
    ctx.decode_device(dp, enc, outs[i % 3], stream_length=m)
torch.cuda.synchronize()
assert torch.equal(outs[2], d_in)
info = dp.launch_info()
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(16384 * 8, np.uint64)
L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
st = buf.reshape(-1, 8).astype(np.int64)[: info["grid"] * info["waves_per_block"]]
print(f"launch {info['grid']} x {info['block']} threads, spread {info['spread']}, weights {info['class_weights']}, {info['chains']} chains")
if info["spread"] != 2:
    sys.exit("not the dealt launch")
t0 = st[:, 0][st[:, 0] > 0].min()
wpb = info["waves_per_block"]
idx = np.arange(len(st))
cls = (idx // wpb >= (info["grid"] + 1) // 2).astype(int) * 4 + np.minimum((idx % wpb) // (wpb // 4), 3)
us = lambda col: (st[:, col] - t0) / 100.0
print("class: entry p50 | table built p50 | first group p50 | done p10 / p50 / p90 / max   (us since the launch's first wave)")
for k in range(8):
    mk = (cls == k) & (st[:, 3] > 0) & (st[:, 2] > 0)
    if mk.any():
        dn = us(3)[mk]
        print(f"  {k}: {np.median(us(0)[mk]):5.1f} | {np.median(us(1)[mk]):5.1f} | {np.median(us(2)[mk]):5.1f} | {np.percentile(dn, 10):5.1f} / {np.median(dn):5.1f} / {np.percentile(dn, 90):5.1f} / {dn.max():5.1f}")
alld = us(3)[st[:, 3] > 0]
print(f"all waves done: p50 {np.median(alld):.1f}  p99 {np.percentile(alld, 99):.1f}  max {alld.max():.1f}")
# the launch's last waves: who they are (a late start? one workgroup? one class?)
order = np.argsort(-us(3))
print("slowest waves: workgroup.wave class | first group | done | decode")
for i in order[:24]:
    print(f"  {i // wpb:4d}.{i % wpb:<2d} c{cls[i]} | {us(2)[i]:5.1f} | {us(3)[i]:5.1f} | {us(3)[i] - us(2)[i]:5.1f}")
wg_done = us(3).reshape(-1, wpb).max(axis=1)
late = np.argsort(-wg_done)[:32]
print("latest workgroups:", sorted(late.tolist()))
print("decode time (done - first group) by class p50/p99/max:", " ".join(f"c{k}:{np.median((us(3)-us(2))[cls==k]):.1f}/{np.percentile((us(3)-us(2))[cls==k],99):.1f}/{(us(3)-us(2))[cls==k].max():.1f}" for k in range(8)))
