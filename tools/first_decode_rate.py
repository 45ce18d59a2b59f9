"""A reference-emitted mt_ stream (64 KiB-class blocks, no index) that only exists in device memory: the device-side header walk,
the FIRST decode (one chain per block; hsrans_decode_device_indexing records the checkpoints on the way) and every LATER decode
with the plan that pass left behind, beside the plain unindexed decode.  Run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ctx = H.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = synth.enwik8_shaped(n)
stream = H.encode(H.MT, 64, 11, data)  # the reference's adaptive block policy, byte-identical to its encoder's stream
d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
d_ref = torch.from_numpy(data).cuda()
alg = stream.size + n


def timed(fn, reps=5):
    best, keep = 1e9, []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep.append(fn())  # (kept: dropping the previous result here would put its hipFree inside the next timed call)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, keep[-1]


def gpu_ms(fn, reps=10):
    """average of `reps` back-to-back launches by HIP events on torch's current stream (which the launches use)"""
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_k2, base = timed(lambda: ctx.make_device_plan_from_stream(H.MT, 64, 11, d_in, stream.size, n), 3)
out = torch.zeros(n, dtype=torch.uint8, device="cuda")
t_plain, _ = timed(lambda: ctx.decode_device(base, d_in, out, stream_length=stream.size))
assert torch.equal(out, d_ref)
ev_plain = gpu_ms(lambda: ctx.decode_device(base, d_in, out, stream_length=stream.size))
host_base = ctx.make_device_plan(H.plan_build(H.MT, 64, 11, stream))
ev_plain_host = gpu_ms(lambda: ctx.decode_device(host_base, d_in, out, stream_length=stream.size))
print(json.dumps({"unindexed_decode_ms_events": {"device_planned": round(ev_plain, 4), "host_planned": round(ev_plain_host, 4)},
                  "launch_device_planned": base.launch_info(), "launch_host_planned": host_base.launch_info()}), flush=True)
for interval in (32, 64):
    out.zero_()
    t_first, indexed = timed(lambda: ctx.decode_device_indexing(base, d_in, out, interval, stream_length=stream.size), 3)
    assert torch.equal(out, d_ref)
    out.zero_()
    t_later, _ = timed(lambda: ctx.decode_device(indexed, d_in, out, stream_length=stream.size), 10)
    assert torch.equal(out, d_ref) and ctx.status(indexed) == 0
    frac = lambda t: round(alg / t / 8e12, 3)
    ev_later = gpu_ms(lambda: ctx.decode_device(indexed, d_in, out, stream_length=stream.size), 20)
    print(json.dumps({"size": n, "stream": int(stream.size), "blocks": base.launch_info()["chains"], "interval": interval,
                      "k2_walk_ms": round(t_k2 * 1e3, 3), "plain_unindexed_decode_ms": round(t_plain * 1e3, 3), "plain_frac": frac(t_plain),
                      "first_decode_indexing_ms": round(t_first * 1e3, 3), "later_decode_ms": round(t_later * 1e3, 3), "later_decode_ms_events": round(ev_later, 4), "later_frac_events": frac(ev_later * 1e-3),
                      "indexed_chains": indexed.launch_info()["chains"],
                      "assembly": "host" if os.environ.get("HSRANS_INDEX_ASSEMBLE_ON_HOST") else "device",
                      "note": "wall clock around each call incl. its synchronisation (best of 3 / 5 / 10 calls); first_decode includes the recording pass and the plan assembly"}), flush=True)
