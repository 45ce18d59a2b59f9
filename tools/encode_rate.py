"""GPU encoder rate: hsrans_encode_device on the 100 MB enwik8-shaped input, HBM-resident.
Prints one JSON line per configuration.  Run on the GPU box: python tools/encode_rate.py [--size N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--cpu-sample", type=int, default=16 << 20, help="bytes the library's own scalar host encoder is timed on (0 = skip)")
args = ap.parse_args()
ctx = H.Context(0)


def sustained_ms(fn, launches=40, regions=5, settle_s=0.025):
    """per-launch time behind `settle_s` of sustained launches, median of `regions` regions: the first launches after an idle gap run
    slower; rounds 1-4 timed `reps` launches straight after the encode and quoted 55-60 us where the sustained figure is 48"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(regions):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / launches)
    return sorted(ts)[len(ts) // 2]


d = synth.enwik8_shaped(args.size, seed=1)
d_in = torch.from_numpy(d).cuda()
for states, bits, block in ((64, 11, 1 << 16), (64, 11, 1 << 15), (64, 11, 1 << 18), (32, 11, 1 << 16), (64, 15, 1 << 16)):
    d_out = torch.empty(H.capacity(H.MT, states, d.size), dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block)  # synchronises its stream
        ts.append(time.perf_counter() - t0)
    best, mean = min(ts), sum(ts) / len(ts)
    # the same encode with the sidecar plan (checkpoint every 32 groups) built on the device, then the decode it enables
    tp = []
    for _ in range(max(2, args.reps // 2)):
        t0 = time.perf_counter()
        n2, dplan = ctx.encode_device(H.MT, states, bits, d_in, d_out, block_size=block, index_interval=32, want_plan=True)
        tp.append(time.perf_counter() - t0)
    back = torch.empty(d.size, dtype=torch.uint8, device="cuda")
    ctx.decode_device(dplan, d_out, back, stream_length=n2)
    torch.cuda.synchronize()
    ok = n2 == n and ctx.status(dplan) == 0 and bool(torch.equal(back, d_in))
    dec_ms = sustained_ms(lambda: ctx.decode_device(dplan, d_out, back, stream_length=n2))
    print(json.dumps({"codec": f"mt_ rANS32x{states} 16w {bits}", "block": block, "size": args.size, "stream": n, "ratio": round(n / args.size, 4),
                      "ms_best": round(best * 1e3, 3), "ms_mean": round(mean * 1e3, 3), "GB_s_best": round(args.size / best / 1e9, 1),
                      "with_plan_G32_ms_best": round(min(tp) * 1e3, 3), "decode_with_that_plan_ms": round(dec_ms, 4),
                      "decode_MiB_s": round(args.size / 2**20 / (dec_ms * 1e-3)), "round_trip_bit_exact": ok}))

if args.cpu_sample:
    # the same stream from the library's scalar host encoder (hsrans_encode_ex, 1 core): the CPU side of the comparison
    sample = d[: args.cpu_sample]
    t0 = time.perf_counter()
    s_host = H.encode(H.MT, 64, 11, sample, block_size=1 << 16, independent_blocks=True)
    dt = time.perf_counter() - t0
    d_out = torch.empty(H.capacity(H.MT, 64, sample.size), dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(H.MT, 64, 11, d_in[: sample.size], d_out, block_size=1 << 16)
    same = n == s_host.size and bool((d_out[:n].cpu().numpy() == s_host).all())
    print(json.dumps({"codec": "mt_ rANS32x64 16w 11 host encoder (1 core)", "block": 1 << 16, "size": int(sample.size), "ms": round(dt * 1e3, 1),
                      "MiB_s": round(sample.size / dt / 2**20, 1), "gpu_stream_identical": same}))

# ---- the raw format: one dependent chain per coder state = one wavefront (hsrans_encode_device_raw) ----
for states, bits in ((64, 11), (32, 11), (64, 15)):
    d_out = torch.empty(H.capacity(H.RAW, states, d.size), dtype=torch.uint8, device="cuda")
    n = ctx.encode_device_raw(states, bits, d_in, d_out)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.encode_device_raw(states, bits, d_in, d_out)
        ts.append(time.perf_counter() - t0)
    groups = H.index_boundaries(states, bits, d.size, ctx)
    t0 = time.perf_counter()
    n2, dplan = ctx.encode_device_raw(states, bits, d_in, d_out, index_groups=groups, want_device_plan=True)
    t_plan = time.perf_counter() - t0
    back = torch.empty(d.size, dtype=torch.uint8, device="cuda")
    ctx.decode_device(dplan, d_out, back, stream_length=n2)
    torch.cuda.synchronize()
    ok = n2 == n and ctx.status(dplan) == 0 and bool(torch.equal(back, d_in))
    dec_ms = sustained_ms(lambda: ctx.decode_device(dplan, d_out, back, stream_length=n2))
    line = {"codec": f"raw rANS32x{states} 16w {bits}", "size": args.size, "stream": n, "ratio": round(n / args.size, 4), "ms_best": round(min(ts) * 1e3, 2),
            "MB_s_best": round(args.size / min(ts) / 1e6, 1), "with_wave_index_and_device_plan_ms": round(t_plan * 1e3, 2), "index_chains": dplan.launch_info()["chains"],
            "decode_with_that_plan_ms": round(dec_ms, 4), "round_trip_bit_exact": ok, "note": "one wavefront: the format is one dependent chain per coder state"}
    if args.cpu_sample and states == 64 and bits == 11:
        sample = d[: args.cpu_sample]
        t0 = time.perf_counter()
        s_host = H.encode(H.RAW, 64, 11, sample)
        dt = time.perf_counter() - t0
        line["host_encoder_1_core_MB_s"] = round(sample.size / dt / 1e6, 1)
        line["host_encoder_sample"] = int(sample.size)
    print(json.dumps(line), flush=True)
