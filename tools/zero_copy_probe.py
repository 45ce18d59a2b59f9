"""Probe: the decode kernel reading its stream from / writing its output to PINNED HOST memory directly (no staging copies):
the wavefronts' 512-byte stream requests and 256-byte streaming stores cross PCIe themselves.  Compared with the staged paths
(hsrans_hpipe, upload-decode-download).  Run on the GPU box: python tools/zero_copy_probe.py [--size N]"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--block", type=int, default=1 << 18)
ap.add_argument("--interval", type=int, default=32)
args = ap.parse_args()
ctx = H.Context(0)
n = args.size
g = torch.Generator(device="cuda").manual_seed(11)
d_in = torch.rand(n, device="cuda", generator=g).pow_(6).mul_(205).to(torch.uint8)
d_enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
m, dplan = ctx.encode_device(H.MT, 64, 11, d_in, d_enc, block_size=args.block, index_interval=args.interval, want_plan=True)
plan = ctx.read_device_plan(dplan, capacity=1 << 30)
mp = (m + 15) // 16 * 16
host_stream = torch.zeros(mp + 64, dtype=torch.uint8).pin_memory()
host_stream[:m].copy_(d_enc[:m])
host_ref = d_in.cpu()
host_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
d_stream = d_enc
L = ctx.L
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def launch(p_in, p_out):
    rc = L.hsrans_decode_device(ctx.handle, dplan.handle, p_in, m, p_out, n, s)
    assert rc == 0, rc


def timed(fn, check=None):
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts)


cases = {"stream in HBM, output in HBM": (d_stream.data_ptr(), d_out.data_ptr()),
         "stream in HBM, output in pinned host memory": (d_stream.data_ptr(), host_out.data_ptr()),
         "stream in pinned host memory, output in HBM": (host_stream.data_ptr(), d_out.data_ptr()),
         "stream and output in pinned host memory (zero copy)": (host_stream.data_ptr(), host_out.data_ptr())}
for name, (pi, po) in cases.items():
    host_out.zero_()
    d_out.zero_()
    try:
        launch(pi, po)
        torch.cuda.synchronize()
        ok = bool(torch.equal(host_out if po == host_out.data_ptr() else d_out.cpu(), host_ref)) and ctx.status(dplan) == 0
        best, mean = timed(lambda: launch(pi, po))
        print(json.dumps({"mode": name, "size": n, "compressed": m, "ms_best": round(best * 1e3, 3), "ms_mean": round(mean * 1e3, 3),
                          "decoded_GB_s": round(n / best / 1e9, 1), "bit_exact": ok}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"mode": name, "error": repr(e)}), flush=True)

# the staged paths on the same buffers
hs = host_stream[:m]
pipeline.decode_from_host_unpipelined(ctx, plan, hs, host_out)
best, mean = timed(lambda: pipeline.decode_from_host_unpipelined(ctx, plan, hs, host_out))
print(json.dumps({"mode": "upload, decode, download one after the other", "ms_best": round(best * 1e3, 2), "decoded_GB_s": round(n / best / 1e9, 1)}), flush=True)
for k in (2, 4, 8):
    dec = pipeline.PipelinedHostDecoder(ctx, plan, n_slices=k)
    host_out.zero_()
    dec.decode(hs, host_out)
    ok = bool(torch.equal(host_out, host_ref))
    best, mean = timed(lambda: dec.decode(hs, host_out))
    print(json.dumps({"mode": f"hsrans_hpipe, {k} slices", "ms_best": round(best * 1e3, 2), "ms_mean": round(mean * 1e3, 2), "decoded_GB_s": round(n / best / 1e9, 1),
                      "bit_exact": ok}), flush=True)
    del dec
# plain copies, for the link's own rates: down alone, up alone, both at once on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        d_stream[:m].copy_(hs, non_blocking=True)
    with torch.cuda.stream(s2):
        host_out.copy_(d_out, non_blocking=True)


for name, fn in (("D2H copy of the output alone", lambda: host_out.copy_(d_out, non_blocking=True)),
                 ("H2D copy of the stream alone", lambda: d_stream[:m].copy_(hs, non_blocking=True)), ("both copies at once (two streams)", both)):
    best, mean = timed(fn)
    print(json.dumps({"mode": name, "ms_best": round(best * 1e3, 2), "ms_mean": round(mean * 1e3, 2)}), flush=True)
