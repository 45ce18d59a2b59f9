#!/usr/bin/env python3
"""A/B of decode plans for one raw stream: uniform checkpoints (G groups) vs one chain per resident wave (hsrans_index_boundaries).
Prints one JSON line per variant: kernel time (HIP events around K launches), MiB/s, fraction of the 8 TB/s HBM roofline.

    python tools/ab_plans.py [--size N] [--bits B] [--states S] [--steps K] [--variants g32,direct,...]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=100_000_000)
    ap.add_argument("--bits", type=int, default=11)
    ap.add_argument("--states", type=int, default=64)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--variants", default="g32,direct")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    ctx = H.Context(0)
    dev = torch.device("cuda", 0)
    n, S, bits = a.size, a.states, a.bits
    data = synth.enwik8_shaped(n, seed=20241008)
    d_ref = torch.from_numpy(data).to(dev)
    stream = None
    for v in a.variants.split(","):
        if v.startswith("g"):
            st, plan = H.encode(H.RAW, S, bits, data, index_interval=int(v[1:]))
        else:
            st, plan = H.encode(H.RAW, S, bits, data, index_groups=H.index_boundaries(S, bits, n, ctx))
        if stream is None:
            stream = st
            pad = (-stream.size) % 16
            d_in = torch.from_numpy(np.concatenate([stream, np.zeros(pad, np.uint8)])).to(dev)
        assert np.array_equal(st, stream)
        d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
        dplan = ctx.make_device_plan(plan)
        ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
        torch.cuda.synchronize()
        assert ctx.status(dplan) == 0 and torch.equal(d_out, d_ref), v
        for _ in range(5):
            ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for rep in range(3):
            e0.record()
            for _ in range(a.steps):
                ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            best = ms if best is None else min(best, ms)
        assert torch.equal(d_out, d_ref)
        alg = stream.size + n
        print(json.dumps({"tag": a.tag, "variant": v, "bits": bits, "states": S, "n": n, "chains": H.plan_chain_count(plan), "plan_bytes": int(plan.size),
                          "kernel_us": best * 1e3, "MiB_s": n / 2**20 / (best * 1e-3), "frac_hbm": alg / (best * 1e-3) / 8e12,
                          "launch": dplan.launch_info()}), flush=True)


if __name__ == "__main__":
    main()
