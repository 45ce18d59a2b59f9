#!/usr/bin/env python3
"""Are some workgroups of the one-chain-per-wave launch ALWAYS slower than their class (a slow CU, a crowded XCD), launch after
launch?  The launch ends with its last wave; per-wave stamps show whole workgroups 5-6 us late (tools/stamps_dealt.py).  If the
same workgroups (grid positions) are late every time, the index could size their chains accordingly, as it does by age class.
Prints, over R launches of the 100 MB raw headline decode (diagnostic library): the correlation between launches of every
workgroup's decode time relative to its class, the same per physical CU (HW_ID), and the slowest grid positions of each launch.

    python tools/wg_stability_probe.py [--launches 8]"""
import argparse, ctypes, os, sys
os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--launches", type=int, default=8)
a = ap.parse_args()
ctx = H.Context(0)
ctx.calibrate()
n = a.size
data = synth.enwik8_shaped(n)
s, plan = H.encode(H.RAW, 64, 11, data, index_groups=H.index_boundaries(64, 11, n, ctx))
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
outs = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
dp = ctx.make_device_plan(plan)
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rel_wg, rel_cu, late = [], [], []
for r in range(a.launches):
    for i in range(6):  # sustained; the stamps are the last launch's
        ctx.decode_device(dp, d_in, outs[(r + i) % 3], stream_length=s.size)
    torch.cuda.synchronize()
    buf = np.zeros(16384 * 8, np.uint64)
    L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
    info = dp.launch_info()
    W = info["grid"] * info["waves_per_block"]
    st = buf.reshape(-1, 8)[:W].astype(np.int64)
    wpb = info["waves_per_block"]
    idx = np.arange(W)
    cls = (idx // wpb >= (info["grid"] + 1) // 2).astype(int) * 4 + (idx % wpb) // (wpb // 4)
    done = (st[:, 3] - st[:, 0].min()) / 100.0
    rel = np.zeros(W)
    for k in range(8):
        m = cls == k
        rel[m] = done[m] - np.median(done[m])
    wg = rel.reshape(-1, wpb).mean(axis=1)
    rel_wg.append(wg)
    hw = st[:, 6]
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 0x1) * 64 + ((hw >> 8) & 0xF)  # XCC, SE, SH, CU
    keys = np.unique(cu_key)
    cu = np.array([rel[cu_key == k].mean() for k in keys])
    rel_cu.append(dict(zip(keys.tolist(), cu.tolist())))
    late.append(np.argsort(-wg)[:12].tolist())
    print(f"launch {r}: last wave {done.max():.1f} us, median {np.median(done):.1f}; workgroup mean lateness: p99 {np.percentile(wg, 99):+.2f} max {wg.max():+.2f} us; {len(keys)} CUs seen; latest grid positions {sorted(late[-1])}")
M = np.array(rel_wg)
c = np.corrcoef(M)
print("correlation of the workgroups' lateness between launches (off-diagonal mean): %.3f" % ((c.sum() - len(M)) / (len(M) * (len(M) - 1))))
common = set(rel_cu[0])
for d in rel_cu[1:]:
    common &= set(d)
C = np.array([[d[k] for k in sorted(common)] for d in rel_cu])
cc = np.corrcoef(C)
print("the same per physical CU (%d CUs in every launch): %.3f" % (len(common), (cc.sum() - len(C)) / (len(C) * (len(C) - 1))))
mean_wg = M.mean(axis=0)
print("mean lateness over the launches, by grid position: p1 %.2f p50 %.2f p99 %.2f max %.2f us; positions above +2 us: %s" %
      (np.percentile(mean_wg, 1), np.median(mean_wg), np.percentile(mean_wg, 99), mean_wg.max(), np.nonzero(mean_wg > 2.0)[0].tolist()))
