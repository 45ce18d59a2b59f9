#!/usr/bin/env python3
"""Many small streams: one launch each against ONE launch for all (hsrans_decode_device_batch) — the many-small-files case the
reference serves by taking file after file (src/main.cpp:841-898).  Raw streams (64 states, 11 bits, one-chain-per-wave index) take the
shared one-chain-per-wave launch, mt_ streams (64 KiB blocks, a checkpoint every 32 groups) the shared grouped launch.  Every stream is
validated bit-exact in both forms.   python tools/small_streams_probe.py [--out FILE]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--window", type=int, default=20)
ap.add_argument("--windows", type=int, default=5)
a = ap.parse_args()
ctx = H.Context(0)
ctx.calibrate(bits=11)
base = synth.enwik8_shaped(16_000_000, seed=77)


def case(container, count, size):
    ms = []
    sizes = [size] * count
    for k in range(count):
        data = synth._permutation(3000 + k)[base[:size]]
        if container == H.RAW:
            s, p = H.encode(H.RAW, 64, 11, data, index_groups=H.index_boundaries_batch(64, 11, sizes, k, ctx))
            s1, p1 = H.encode(H.RAW, 64, 11, data, index_groups=H.index_boundaries(64, 11, size, ctx))  # the index a launch of its own wants
        else:
            s, p = H.encode(H.MT, 64, 11, data, block_size=1 << 16, index_interval=32)
            p1 = p
        ms.append({"data": data, "len": s.size, "d_in": torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda(),
                   "d_out": torch.zeros(size, dtype=torch.uint8, device="cuda"), "dplan": ctx.make_device_plan(p), "dplan_alone": ctx.make_device_plan(p1)})
    batch = ctx.make_batch([m["dplan"] for m in ms])
    ins, outs, lens = [m["d_in"] for m in ms], [m["d_out"] for m in ms], [m["len"] for m in ms]

    def check(tag):
        torch.cuda.synchronize()
        for m in ms:
            assert np.array_equal(m["d_out"].cpu().numpy(), m["data"]), tag
            m["d_out"].zero_()

    def serial():
        for m in ms:
            ctx.decode_device(m["dplan_alone"], m["d_in"], m["d_out"], stream_length=m["len"])

    def one():
        ctx.decode_device_batch(batch, ins, outs, stream_lengths=lens)

    serial()
    check("one launch each")
    one()
    check("one launch for all")
    assert ctx.batch_status(batch) == [0] * count

    def window(fn):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(a.window):
            fn()
        eb.record()
        torch.cuda.synchronize()
        return ea.elapsed_time(eb) / a.window * 1e3

    for _ in range(2):
        window(serial), window(one)
    ser = [window(serial) for _ in range(a.windows)]
    bat = [window(one) for _ in range(a.windows)]
    alg = float(sum(lens)) + count * size
    info = batch.info()
    row = {"container": "raw" if container == H.RAW else "mt_", "streams": count, "bytes_each": size, "one_launch_each_us_total": round(float(np.median(ser)), 1),
           "one_launch_for_all_us_total": round(float(np.median(bat)), 1), "speedup": float(np.median(ser) / np.median(bat)),
           "one_launch_each_frac": alg / (np.median(ser) * 1e-6) / 8e12, "one_launch_for_all_frac": alg / (np.median(bat) * 1e-6) / 8e12,
           "launches_per_call": info["launches"], "direct_members": info["direct_members"], "grouped_members": info["grouped_members"], "bit_exact": True}
    print(json.dumps(row), flush=True)
    if a.out:
        open(a.out, "a").write(json.dumps(row) + "\n")
    batch.close()


for container in (H.RAW, H.MT):
    for count, size in ((32, 1_000_000), (32, 4_000_000), (16, 16_000_000), (8, 4_000_000)):
        case(container, count, size)
        torch.cuda.empty_cache()
