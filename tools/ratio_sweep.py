#!/usr/bin/env python3
"""Throughput against compression ratio, one table.  Every other number in this repository is measured on data
that compresses to 0.64; the words a group consumes drive the ring refills, chunk crossings and waits, and the reference's own
tables quote files at 0.78-0.83 beside enwik8's 0.64 (README.md:32,81,130 of the reference).  For each distribution: the headline
launch (rANS32x64 16w, 11 bits, raw + one chain per wavefront), P streams rotated through HBM, microseconds per decode and the
fraction of 8 TB/s on the distribution's OWN algorithmic bytes (compressed read once + decoded written once); the same P streams
decoded by ONE launch (hsrans_decode_device_batch); every stream validated bit-exact, and a 1 MiB stream of each distribution
against the scalar CPU oracle.

    python tools/ratio_sweep.py [--size 100000000] [--pairs 4] [--out profiles/r05_ratio_sweep.jsonl]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from oracle_lib import RAW, Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100_000_000)
ap.add_argument("--bits", type=int, default=11)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--window", type=int, default=200)
ap.add_argument("--windows", type=int, default=5)
ap.add_argument("--only", default="")
ap.add_argument("--out", default="")
a = ap.parse_args()
n, S, bits, P = a.size, 64, a.bits, a.pairs

DISTS = [
    ("two symbols 15:1", lambda m: synth.two_symbol(m)),
    ("Zipf 1.8 over 256 symbols", lambda m: synth.zipf_bytes(m, 1.8)),
    ("Zipf 1.2 over 205 symbols (the benchmark's data)", lambda m: synth.enwik8_shaped(m)),
    ("Zipf 0.9 over 256 symbols", lambda m: synth.zipf_bytes(m, 0.9)),
    ("Zipf 0.6 over 256 symbols", lambda m: synth.zipf_bytes(m, 0.6)),
    ("uniform bytes", lambda m: synth.uniform_bytes(m)),
    ("non-stationary (Zipf segments under changing permutations + single-byte runs), ONE histogram", lambda m: synth.nonstationary(m)),
]
ctx = H.Context(0)
oracle = Oracle()
cal = ctx.calibrate(bits=bits)
run = n / S / 8192.0
cal["runs"] = [ctx.calibrate_runs(bits=bits, copies=c) for c in sorted({min(16, max(2, round(run / 96))), min(16, max(2, round(P * run / 96)))})]
groups = H.index_boundaries(S, bits, n, ctx)


def window(fn, count):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for t in range(count):
        fn(t)
    eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb) / count * 1e3


for name, gen in DISTS:
    if a.only and a.only not in name:
        continue
    # 1 MiB of the distribution against the oracle (the contract), through the same launch shape
    small = gen(1 << 20)
    s1, p1 = H.encode(H.RAW, S, bits, small, index_groups=H.index_boundaries(S, bits, small.size, ctx))
    r, want = oracle.decode(RAW, S, bits, s1, small.size)
    assert r == small.size and np.array_equal(want, small)
    assert np.array_equal(ctx.decode(H.RAW, S, bits, s1, plan=p1), want), f"{name}: GPU differs from the oracle at 1 MiB"
    base = gen(n)
    datas = [base if k == 0 else synth._permutation(1000 + k)[base] for k in range(P)]
    dplans, bplans, d_in, d_out, lens = [], [], [], [], []
    t0 = time.perf_counter()
    for k, data in enumerate(datas):
        s, p = H.encode(H.RAW, S, bits, data, index_groups=groups)
        lens.append(s.size)
        dplans.append(ctx.make_device_plan(p))
        bplans.append(ctx.make_device_plan(ctx.index_build_at(H.RAW, S, bits, s, H.index_boundaries_batch(S, bits, [n] * P, k, ctx))))
        d_in.append(torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda())
        d_out.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
    t_enc = time.perf_counter() - t0
    batch = ctx.make_batch(bplans)

    def check(tag):
        torch.cuda.synchronize()
        for k in range(P):
            assert np.array_equal(d_out[k].cpu().numpy(), datas[k]), f"{name}: {tag}: stream {k} is not bit-exact"
            d_out[k].zero_()

    def serial(t):
        k = t % P
        ctx.decode_device(dplans[k], d_in[k], d_out[k], stream_length=lens[k])

    def one_launch(t):
        ctx.decode_device_batch(batch, d_in, d_out, stream_lengths=lens)

    for t in range(P):
        serial(t)
    check("one launch per stream")
    one_launch(0)
    check("one launch for all")
    assert ctx.batch_status(batch) == [0] * P
    for _ in range(2):  # settle
        window(serial, a.window)
        window(one_launch, a.window // P)
    ser = [window(serial, a.window) for _ in range(a.windows)]
    bat = [window(one_launch, a.window // P) / P for _ in range(a.windows)]
    for t in range(P):
        serial(t)
    check("after the timed windows")
    alg = float(np.mean(lens)) + n
    row = {"distribution": name, "decoded_bytes": n, "compressed_bytes_mean": float(np.mean(lens)), "ratio": float(np.mean(lens)) / n, "algorithmic_bytes": alg,
           "words_per_group": float(np.mean(lens)) / 2 / (n / S), "streams_rotated": P, "bit_exact": True, "oracle_checked_at_1MiB": True,
           "us_per_decode": round(float(np.median(ser)), 2), "us_per_decode_windows": [round(x, 2) for x in ser], "frac_of_8TBs": alg / (np.median(ser) * 1e-6) / 8e12,
           "MiB_s": n / 2**20 / (np.median(ser) * 1e-6),
           "one_launch_us_per_stream": round(float(np.median(bat)), 2), "one_launch_frac_of_8TBs": alg / (np.median(bat) * 1e-6) / 8e12, "one_launch_imbalance": batch.info()["imbalance"],
           "host_encode_s_per_stream": t_enc / P}
    print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(json.dumps(row) + "\n")
    batch.close()
    del dplans, bplans, d_in, d_out
    torch.cuda.empty_cache()
