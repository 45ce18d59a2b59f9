#!/usr/bin/env python3
"""Which part of the working set has to be cold for the headline decode to slow down?  Times the launch with the same buffers
replayed (warm: Infinity-Cache resident) and with P distinct streams / outputs / both rotated (> 256 MiB in flight), and prints
the per-wave stamps of a cold launch (HSRANS_DEBUG_STAMPS=1).

    python tools/cold_cache.py [--pairs 4] [--steps 40]
"""
import argparse, ctypes, json, os, sys
os.environ["HSRANS_DEBUG_STAMPS"] = "1" if "--stamps" in sys.argv else os.environ.get("HSRANS_DEBUG_STAMPS", "")
if not os.environ["HSRANS_DEBUG_STAMPS"]:
    del os.environ["HSRANS_DEBUG_STAMPS"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--slab", action="store_true", help="carve every stream and output from ONE 2 GiB allocation made first (2 MiB-aligned pieces)")
ap.add_argument("--quick", action="store_true", help="only the warm / streams-rotated / both-rotated lines")
a = ap.parse_args()
n, S, bits, P = a.size, 64, a.bits, a.pairs
slab, slab_off = None, 0
if a.slab:
    slab = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
    slab_off = (-slab.data_ptr()) % (2 << 20)


def dev_bytes(nbytes, src=None):
    """a device buffer of nbytes (zeroed, or a copy of the numpy array `src`)"""
    global slab_off
    if slab is None:
        return torch.zeros(nbytes, dtype=torch.uint8, device="cuda") if src is None else torch.from_numpy(src).cuda()
    t = slab[slab_off:slab_off + nbytes]
    slab_off += (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1)
    if src is None:
        t.zero_()
    else:
        t.copy_(torch.from_numpy(src))
    return t


ctx = H.Context(0)
base = synth.enwik8_shaped(n)
g = H.index_boundaries(S, bits, n, ctx)
ins, outs, plans, lens = [], [], [], []
for k in range(P):
    d = base if k == 0 else synth._permutation(1000 + k)[base]
    s, plan = H.encode(H.RAW, S, bits, d, index_groups=g)
    ins.append(dev_bytes(s.size + (-s.size) % 16, np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])))
    outs.append(dev_bytes(n))
    if k == 0:
        first = d
    plans.append(ctx.make_device_plan(plan))
    lens.append(s.size)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def run(label, pick_in, pick_out, flush_between=False):
    for i in range(4):
        ctx.decode_device(plans[pick_in(i)], ins[pick_in(i)], outs[pick_out(i)], stream_length=lens[pick_in(i)])
    torch.cuda.synchronize()
    if flush_between:
        ts = []
        for i in range(a.steps):
            flush.fill_(i & 0xFF)  # 512 MiB written: everything older leaves the Infinity Cache
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.decode_device(plans[pick_in(i)], ins[pick_in(i)], outs[pick_out(i)], stream_length=lens[pick_in(i)])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = float(np.median(ts))
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.steps):
            ctx.decode_device(plans[pick_in(i)], ins[pick_in(i)], outs[pick_out(i)], stream_length=lens[pick_in(i)])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.steps * 1e3
    print(json.dumps({"case": label, "kernel_us": round(us, 2), "frac_hbm": round((lens[0] + n) / (us * 1e-6) / 8e12, 4)}), flush=True)


run("warm: one stream, one output replayed", lambda i: 0, lambda i: 0)
if not np.array_equal(outs[0].cpu().numpy(), first):
    print(json.dumps({"error": "decoded bytes differ"}), flush=True)
    sys.exit(1)
run("streams rotated, one output", lambda i: i % P, lambda i: 0)
run("one stream, outputs rotated", lambda i: 0, lambda i: i % P)
run("both rotated", lambda i: i % P, lambda i: i % P)
for k in range(1, P):  # does the replayed number depend on WHICH pair (where its buffers landed physically)?
    run("warm: pair %d replayed" % k, lambda i, k=k: k, lambda i, k=k: k)
if a.quick:
    sys.exit(0)
# the SAME stream in P different buffers, one plan: only the stream bytes are cold (plan, states, table stay warm)
copies = [ins[0]] + [ins[0].clone() for _ in range(P - 1)]


def run_copies(label):
    for i in range(4):
        ctx.decode_device(plans[0], copies[i % P], outs[0], stream_length=lens[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.steps):
        ctx.decode_device(plans[0], copies[i % P], outs[0], stream_length=lens[0])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.steps * 1e3
    print(json.dumps({"case": label, "kernel_us": round(us, 2), "frac_hbm": round((lens[0] + n) / (us * 1e-6) / 8e12, 4)}), flush=True)


run_copies("one plan, the same stream in %d buffers rotated (only the stream bytes cold)" % P)
# a pure read of the same bytes for scale: how long does streaming one cold 64 MB buffer take at all?
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
acc = torch.zeros(1, dtype=torch.int64, device="cuda")
for i in range(4):
    acc += copies[i % P].view(torch.int64)[: lens[0] // 8].sum()
e0.record()
for i in range(a.steps):
    acc += copies[i % P].view(torch.int64)[: lens[0] // 8].sum()
e1.record()
torch.cuda.synchronize()
print(json.dumps({"case": "torch .sum() over the rotated stream copies (read-only, cold)", "us": round(e0.elapsed_time(e1) / a.steps * 1e3, 2)}), flush=True)
run("one pair, 512 MiB written between launches (single-launch events)", lambda i: 0, lambda i: 0, flush_between=True)
run("warm, single-launch events (for comparison with the line above)", lambda i: 0, lambda i: 0, flush_between=False)
if a.stamps:
    L = H.load_library()
    L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
    L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    for label, cold in (("warm", False), ("cold", True)):
        for i in range(3 * P):
            k = i % P if cold else 0
            ctx.decode_device(plans[k], ins[k], outs[k], stream_length=lens[k])
        torch.cuda.synchronize()
        buf = np.zeros(16384 * 8, np.uint64)
        L.hsrans_debug_read_stamps(plans[(3 * P - 1) % P if cold else 0].handle, buf.ctypes.data, buf.size)
        st = buf.reshape(-1, 8).astype(np.int64)
        st = st[st[:, 3] > 0]
        rel = (st - st[:, 0].min()) / 100.0
        print(label, "stamps:", " | ".join(f"{name} p50 {np.median(rel[:, c]):.1f} p99 {np.percentile(rel[:, c], 99):.1f}" for name, c in (("entry", 0), ("table", 1), ("ready", 2), ("done", 3))))
