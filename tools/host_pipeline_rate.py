"""End-to-end rate with the stream and the output in pinned host memory (BASELINE config 5 shape): upload / decode / download
overlapped over slices (hypersonic_rans_amd.pipeline) against the same work done one leg after the other.
Run on the GPU box: python tools/host_pipeline_rate.py [--size N] [--slices K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import pipeline, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
ctx = H.Context(0)
n = args.size
# the input is made and encoded on the device (the host encoder would take minutes at this size), then moved to pinned host memory
g = torch.Generator(device="cuda").manual_seed(11)
d_in = torch.rand(n, device="cuda", generator=g).pow_(6).mul_(205).to(torch.uint8)
d_enc = torch.empty(H.capacity(H.MT, 64, n), dtype=torch.uint8, device="cuda")
m, dplan = ctx.encode_device(H.MT, 64, 11, d_in, d_enc, block_size=1 << 18, index_interval=32, want_plan=True)
plan = ctx.read_device_plan(dplan, capacity=1 << 30)
host_stream = torch.empty(m, dtype=torch.uint8).pin_memory()
host_stream.copy_(d_enc[:m])
host_ref = d_in.cpu()
del d_enc, dplan
host_out = torch.empty(n, dtype=torch.uint8).pin_memory()


def timed(fn):
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('   runs ms:', ' '.join(f'{t * 1e3:.1f}' for t in ts), file=sys.stderr)
    return min(ts), sum(ts) / len(ts)


pipeline.decode_from_host_unpipelined(ctx, plan, host_stream, host_out)
assert torch.equal(host_out, host_ref)
best, mean = timed(lambda: pipeline.decode_from_host_unpipelined(ctx, plan, host_stream, host_out))
print(json.dumps({"mode": "upload, decode, download one after the other", "size": n, "compressed": m, "ms_best": round(best * 1e3, 2), "ms_mean": round(mean * 1e3, 2),
                  "decoded_GB_s": round(n / best / 1e9, 1)}), flush=True)
for k in (0, 2, 4, 8, 16, 32):  # 0 = the library's default slicing
    dec = pipeline.PipelinedHostDecoder(ctx, plan, n_slices=k)
    host_out.zero_()
    dec.decode(host_stream, host_out)
    ok = bool(torch.equal(host_out, host_ref))
    best, mean = timed(lambda: dec.decode(host_stream, host_out))
    print(json.dumps({"mode": f"pipelined, {k} slices" + (" (kernels store straight into the host buffer)" if os.environ.get("HSRANS_HPIPE_DIRECT") else ""), "size": n, "compressed": m, "ms_best": round(best * 1e3, 2),
                      "ms_mean": round(mean * 1e3, 2), "decoded_GB_s": round(n / best / 1e9, 1), "bit_exact": ok}), flush=True)
    del dec
