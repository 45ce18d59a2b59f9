#!/usr/bin/env python3
"""Diagnostic: where a persistent decode launch spends its time per wavefront (run with HSRANS_DEBUG_STAMPS=1 on the GPU box)."""
import ctypes
import os
import sys

os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = synth.enwik8_shaped(n)
s, plan = H.encode(H.RAW, 64, 11, data, index_interval=32)
ctx = H.Context(0)
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
dp = ctx.make_device_plan(plan)
for _ in range(5):
    ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
torch.cuda.synchronize()
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8192 * 4, np.uint64)
got = L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
st = buf.reshape(-1, 4).astype(np.int64)
st = st[st[:, 3] > 0]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0  # s_memtime ticks at 100 MHz -> us
print("waves", len(st))
for name, col in (("entry", 0), ("table built", 1), ("stream ready", 2), ("done", 3)):
    v = rel[:, col]
    print(f"{name:13s} min {v.min():8.2f}  p50 {np.median(v):8.2f}  p99 {np.percentile(v, 99):8.2f}  max {v.max():8.2f} us")
d = rel[:, 3] - rel[:, 2]
print(f"decode span   min {d.min():8.2f}  p50 {np.median(d):8.2f}  max {d.max():8.2f} us")

# who finishes late?  group by wave slot inside the workgroup and by workgroup half
w = np.arange(len(buf) // 4)
valid = buf.reshape(-1, 4)[:, 3] > 0
done = (buf.reshape(-1, 4)[:, 3].astype(np.int64) - t0) / 100.0
wave_in_wg = w % 16
blk = w // 16
print("done by wave-in-workgroup:", " ".join(f"{done[valid & (wave_in_wg == k)].mean():.1f}" for k in range(16)))
print("done by SIMD slot (wave%4):", " ".join(f"{done[valid & (wave_in_wg % 4 == k)].mean():.1f}" for k in range(4)))
print("done by wave//4:", " ".join(f"{done[valid & (wave_in_wg // 4 == k)].mean():.1f}" for k in range(4)))
nb = blk.max() + 1
print("done by workgroup half:", f"{done[valid & (blk < nb // 2)].mean():.1f} {done[valid & (blk >= nb // 2)].mean():.1f}")
print("done by workgroup % 8 (XCD):", " ".join(f"{done[valid & (blk % 8 == k)].mean():.1f}" for k in range(8)))
