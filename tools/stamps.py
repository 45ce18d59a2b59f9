#!/usr/bin/env python3
"""Diagnostic: where a persistent / direct decode launch spends its time per wavefront (sets HSRANS_DEBUG_STAMPS=1).

    python tools/stamps.py [--size N] [--bits B] [--index wave|G]
"""
import argparse
import ctypes
import os
import sys

os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--index", default="32")
ap.add_argument("--states", type=int, default=64)
ap.add_argument("--cold", type=int, default=0, help="rotate this many distinct streams (with their own outputs and plans) so that every launch reads from HBM")
ap.add_argument("--flush", action="store_true", help="write 512 MiB between the dumped launches (everything leaves the Infinity Cache)")
ap.add_argument("--dump-launches", type=int, default=0, help="with HSRANS_STAMPS_DUMP: also record this many further launches (stamps_all[k], set_idx[k])")
a = ap.parse_args()
n = a.size
data = synth.enwik8_shaped(n)
ctx = H.Context(0)
if a.index == "wave":
    s, plan = H.encode(H.RAW, a.states, a.bits, data, index_groups=H.index_boundaries(a.states, a.bits, n, ctx))
else:
    s, plan = H.encode(H.RAW, a.states, a.bits, data, index_interval=int(a.index))
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
dp = ctx.make_device_plan(plan)
if a.cold > 1:
    sets = [(dp, d_in, d_out, s.size)]
    for k in range(1, a.cold):
        dk = synth._permutation(1000 + k)[data]
        if a.index == "wave":
            sk, pk = H.encode(H.RAW, a.states, a.bits, dk, index_groups=H.index_boundaries(a.states, a.bits, n, ctx))
        else:
            sk, pk = H.encode(H.RAW, a.states, a.bits, dk, index_interval=int(a.index))
        sets.append((ctx.make_device_plan(pk), torch.from_numpy(np.concatenate([sk, np.zeros((-sk.size) % 16, np.uint8)])).cuda(),
                     torch.zeros(n, dtype=torch.uint8, device="cuda"), sk.size))
    for i in range(3 * a.cold):
        dpk, ik, ok, lk = sets[i % a.cold]
        ctx.decode_device(dpk, ik, ok, stream_length=lk)
    dp = sets[(3 * a.cold - 1) % a.cold][0]
else:
    for _ in range(5):
        ctx.decode_device(dp, d_in, d_out, stream_length=s.size)
torch.cuda.synchronize()
info = dp.launch_info()
waves = info["waves_per_block"]
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(16384 * 8, np.uint64)
got = L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, buf.size)
allst = buf.reshape(-1, 8).astype(np.int64)
valid = allst[:, 3] > 0
st = allst[valid]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0  # s_memtime ticks at 100 MHz -> us
print(f"index {a.index}  bits {a.bits}  chains {H.plan_chain_count(plan)}  waves {len(st)}  launch {info}")
for name, col in (("entry", 0), ("table built", 1), ("stream ready", 2), ("static done", 4), ("done", 3)):
    v = rel[:, col]
    print(f"{name:13s} min {v.min():8.2f}  p10 {np.percentile(v, 10):8.2f}  p50 {np.median(v):8.2f}  p90 {np.percentile(v, 90):8.2f}  p99 {np.percentile(v, 99):8.2f}  max {v.max():8.2f} us")
if st[:, 5].max() > 0:
    ghz = st[:, 5] / ((st[:, 3] - st[:, 0]) * 10.0)
    print(f"shader clock over the wave's lifetime: min {ghz.min():.2f}  p50 {np.median(ghz):.2f}  max {ghz.max():.2f} GHz")
d = rel[:, 4] - rel[:, 2]
print(f"static span   min {d.min():8.2f}  p50 {np.median(d):8.2f}  max {d.max():8.2f} us")

# who finishes late?  group by wave slot inside the workgroup and by workgroup half
w = np.arange(len(allst))
static_done = (allst[:, 4] - t0) / 100.0
done = (allst[:, 3] - t0) / 100.0
wave_in_wg = w % waves
blk = w // waves
pc = max(1, waves // 4)
nb = blk[valid].max() + 1
for label, arr in (("static done", static_done), ("done", done)):
    print(f"{label} by wave//{pc} (age class), first half of the grid:", " ".join(f"{arr[valid & (wave_in_wg // pc == k) & (blk < (nb + 1) // 2)].mean():.1f}" for k in range(4)),
          "| second half:", " ".join(f"{arr[valid & (wave_in_wg // pc == k) & (blk >= (nb + 1) // 2)].mean():.1f}" for k in range(4)) if nb > 1 else "")
print("done by workgroup % 8 (XCD):", " ".join(f"{done[valid & (blk % 8 == k)].mean():.1f}" for k in range(8)))
entry = (allst[:, 0] - t0) / 100.0
nbk = 8
print("by workgroup index (8 buckets over the grid): entry", " ".join(f"{entry[valid & (blk * nbk // nb == k)].mean():.1f}" for k in range(nbk)),
      "| done", " ".join(f"{done[valid & (blk * nbk // nb == k)].mean():.1f}" for k in range(nbk)),
      "| max done", " ".join(f"{done[valid & (blk * nbk // nb == k)].max():.1f}" for k in range(nbk)))
cu_slot = blk % (nb // 2) if nb > 1 else blk
late = np.argsort(-np.where(valid, done, 0))[:12]
print("latest waves (workgroup, wave, entry, ready, done):", [(int(blk[i]), int(wave_in_wg[i]), round(float(entry[i]), 1), round(float((allst[i, 2] - t0) / 100.0), 1), round(float(done[i]), 1)) for i in late])
wg_done = np.array([done[valid & (blk == b)].max() for b in range(nb)])
wg_entry = np.array([entry[valid & (blk == b)].min() for b in range(nb)])
print("per-workgroup last wave: p10 %.1f p50 %.1f p90 %.1f max %.1f; corr(entry, done) %.2f" % (np.percentile(wg_done, 10), np.median(wg_done), np.percentile(wg_done, 90), wg_done.max(), np.corrcoef(wg_entry, wg_done)[0, 1]))
if allst[valid, 6].max() > 0:  # HW_ID | XCC_ID << 32 (k_decode_direct): where the waves really ran
    hw = allst[:, 6] & 0xFFFFFFFF
    xcc = (allst[:, 6] >> 32) & 0xF
    cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
    ids = np.unique(cu[valid])
    per_cu = np.array([done[valid & (cu == i)].max() for i in ids])
    print(f"hardware CUs used: {len(ids)}; last wave per CU: p10 {np.percentile(per_cu, 10):.1f} p50 {np.median(per_cu):.1f} p90 {np.percentile(per_cu, 90):.1f} max {per_cu.max():.1f} us;",
          "done by XCC:", " ".join(f"{done[valid & (xcc == k)].mean():.1f}" for k in range(8)))
if os.environ.get("HSRANS_STAMPS_DUMP"):
    from hypersonic_rans_amd import api as _api
    hdr, cf, pieces = _api.plan_tables(plan if a.cold <= 1 else pk)
    more, which = [], []
    for i in range(a.dump_launches):
        k = i % a.cold if a.cold > 1 else 0
        dpk, ik, ok, lk = sets[k] if a.cold > 1 else (dp, d_in, d_out, s.size)
        if a.flush:
            if i == 0:
                flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
            flush_buf.fill_(i & 0xFF)
        ctx.decode_device(dpk, ik, ok, stream_length=lk)
        torch.cuda.synchronize()
        b2 = np.zeros(16384 * 8, np.uint64)
        L.hsrans_debug_read_stamps(dpk.handle, b2.ctypes.data, b2.size)
        more.append(b2.reshape(-1, 8).astype(np.int64))
        which.append(k)
    np.savez(os.environ["HSRANS_STAMPS_DUMP"], stamps=allst, words_off=pieces["words_off"], out_off=pieces["out_off"], steps=pieces["steps"],
             in_ptr=np.array([(sets[(3 * a.cold - 1) % a.cold][1] if a.cold > 1 else d_in).data_ptr()]), out_ptr=np.array([(sets[(3 * a.cold - 1) % a.cold][2] if a.cold > 1 else d_out).data_ptr()]),
             stamps_all=np.array(more), set_idx=np.array(which),
             in_ptrs=np.array([t[1].data_ptr() for t in sets] if a.cold > 1 else [d_in.data_ptr()]), out_ptrs=np.array([t[2].data_ptr() for t in sets] if a.cold > 1 else [d_out.data_ptr()]))
