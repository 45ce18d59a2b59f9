#!/usr/bin/env python3
"""Headline benchmark: bit-exact rANS32x64 16w 11-bit (raw) decode of a 100 MB enwik8-shaped stream on MI355X.

A "step" = one decode of the whole stream (compressed input and decoded output resident in HBM).  With --gpus N
(launched by torch.distributed.run, one rank per GPU) every rank decodes its own 100 MB stream (weak scaling, no
data-path collective: the streams are independent objects); the timed region is bracketed by barrier + synchronize and
the max over ranks is reported.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--interval G] [--size BYTES] [--bits B] [--no-cpu]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable by a copy kernel


def cpu_baseline(stream: np.ndarray, data: np.ndarray, states: int, bits: int, budget_s: float = 12.0) -> dict:
    """Times the CPU decoder on this host, rank 0 only.  Prefers the REAL reference's fastest AVX2 decoder
    (oracle/_ref, kind "reference"); otherwise the scalar oracle restatement (kind "port").  Checker code only."""
    from oracle_lib import RAW, Oracle, Ref

    n = data.size
    if Ref.available():
        ref = Ref()
        variant = 1 if ref.L.hsref_has_avx2() else 0
        best, runs, t_total = None, 0, 0.0
        while runs < 3 or (t_total < min(budget_s, 4.0) and runs < 40):
            t0 = time.perf_counter()
            r, out = ref.decode(RAW, states, bits, stream, n, variant=variant)
            dt = time.perf_counter() - t0
            assert r == n
            if runs == 0:
                assert np.array_equal(out, data), "reference CPU decoder output differs from the original data"
            best = dt if best is None else min(best, dt)
            t_total += dt
            runs += 1
        name = ("rANS32x64_xmmShfl2_16w_decode_avx2_varC_%d" if bits <= 12 else "rANS32x64_xmmShfl2_16w_decode_avx2_varA_%d") % bits if variant else "rANS32x64_16w_decode_scalar_%d" % bits
        return {"value": n / 2**20 / best, "unit": "MiB/s", "cores": 1, "kind": "reference",
                "sample": f"whole {n}-byte stream, best of {runs} runs, {name} from oracle/_ref (real reference, clang -O3)",
                "cpu": _cpu_model(), "host_cores": os.cpu_count()}
    orc = Oracle()
    t0 = time.perf_counter()
    r, out = orc.decode(RAW, states, bits, stream, n)
    dt = time.perf_counter() - t0
    assert r == n and np.array_equal(out, data)
    return {"value": n / 2**20 / dt, "unit": "MiB/s", "cores": 1, "kind": "port",
            "sample": f"whole {n}-byte stream, 1 run, scalar oracle restatement (oracle/hsrans_oracle.c)", "cpu": _cpu_model(),
            "host_cores": os.cpu_count()}


def _pmc_traffic(n: int, states: int, bits: int, interval: int):
    """HBM bytes per launch from the committed rocprofv3 PMC run of this very workload (profiles/*_pmc.json, written by
    tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).
    PMC counters cannot be collected from inside this process, so this is null unless such a run matches the workload."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            j = json.load(open(f))
            cfg = j["bench_line_under_trace"]["config"]
            if (cfg["decoded_bytes"], cfg["states"], cfg["bits"], cfg["index_interval_groups"]) == (n, states, bits, interval):
                best = float(j["hbm_traffic_bytes_per_launch"]["total"])
        except (KeyError, ValueError, OSError):
            continue
    return best


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=100_000_000)
    ap.add_argument("--bits", type=int, default=11)
    ap.add_argument("--states", type=int, default=64)
    ap.add_argument("--interval", type=int, default=32, help="checkpoint interval of the sidecar plan, in groups of `states` symbols")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-single", action="store_true", help="skip the un-indexed single-wavefront measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # one process per GPU; if the launcher narrowed each rank's view to its own GPU (HIP_VISIBLE_DEVICES), index 0 is that GPU
    n_visible = torch.cuda.device_count()
    dev_index = local_rank % n_visible if n_visible else 0
    if distributed:
        import torch.distributed as dist

        torch.cuda.set_device(dev_index)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    if args.gpus != world and rank == 0 and distributed:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU decode path")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    ctx = H.Context(dev_index)

    # ---- synthetic workload (BASELINE.json configs[1]): enwik8-shaped, one stream per rank -------------------------
    n, S, bits = args.size, args.states, args.bits
    data = synth.enwik8_shaped(n, seed=20241008 + rank)
    t0 = time.perf_counter()
    stream, plan = H.encode(H.RAW, S, bits, data, index_interval=args.interval)
    t_enc = time.perf_counter() - t0
    chains = H.plan_chain_count(plan)
    pad = (-stream.size) % 16
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros(pad, np.uint8)])).to(dev)
    d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_ref = torch.from_numpy(data).to(dev)
    dplan = ctx.make_device_plan(plan)

    def step():
        ctx.decode_device(dplan, d_in, d_out, stream_length=stream.size)

    step()  # validation decode (not one of the W warm-up steps): the checks below idle the GPU, so they come before the warm-up
    torch.cuda.synchronize()
    assert ctx.status(dplan) == 0
    assert torch.equal(d_out, d_ref), "GPU output is not bit-exact"
    d_out.zero_()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ------------------------------------------
    # One HIP event pair brackets the K launches ON THE LAUNCH STREAM (hsrans_decode_device launches on torch's current
    # stream, which is where torch.cuda.Event records): span / K = the kernel's average launch duration for the roofline.
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.warmup):  # W untimed warm-up steps, immediately in front of the timed region
        step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_a.record()
    for _ in range(args.steps):
        step()
    ev_b.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms_span = ev_a.elapsed_time(ev_b) / args.steps

    # per-launch spread (outside the timed region; each pair adds event/launch latency, so only min/max are reported)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 20))]
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    assert ctx.status(dplan) == 0
    assert torch.equal(d_out, d_ref), "GPU output is not bit-exact after the timed region"
    sha = hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest()
    assert sha == hashlib.sha256(data.tobytes()).hexdigest()

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        info = dplan.launch_info()
        ms_per_step = elapsed * 1e3 / args.steps
        k_avg = float(kernel_ms_span)
        k_min = float(np.min(kernel_ms))
        alg_bytes = stream.size + n  # compressed bytes read once + decoded bytes written once
        achieved = alg_bytes / (k_avg * 1e-3) / 1e9
        result = {
            "metric": "decode MiB/s (bit-exact) on 100 MB stream",
            "value": world * n / 2**20 / elapsed * args.steps,
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 state / u16 word / u8 symbol (integer)",
            "data": "synthetic",
            "config": {
                "workload": f"rANS32x{S} 16w {bits}-bit (raw) decode, {n} B enwik8-shaped synthetic (Zipf1.2/205 symbols, seed 20241008+rank), "
                            f"one stream per GPU, sidecar plan with a checkpoint every {args.interval} groups ({chains} chains)",
                "container": "raw", "states": S, "bits": bits, "decoded_bytes": n, "compressed_bytes": int(stream.size),
                "ratio": stream.size / n, "plan_bytes": int(plan.size), "index_interval_groups": args.interval, "chains": chains,
                "launch": info, "bit_exact": True, "sha256": sha, "host_encode_s": t_enc,
            },
            # the reference harness prints min/mean throughput per decoder (src/main.cpp:72-118): same two numbers here
            "mib_s": {"mean_over_timed_region": world * n / 2**20 / elapsed * args.steps, "best_single_launch": n / 2**20 / (k_min * 1e-3),
                      "launches_in_timed_region": args.steps},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": _pmc_traffic(n, S, bits, args.interval),
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": k_avg, "kernel_ms_single_launch_min": k_min,
                "kernel": "hsrans::k_decode<%d, %s>" % (3 if info["shared_table"] and bits <= 11 else (2 if bits >= 13 else (1 if bits == 12 else 0)),
                                                        "true" if info["shared_table"] else "false"),
            },
        }

        # the same stream WITHOUT the sidecar plan: one wavefront, one dependent chain (SURVEY.md finding 2)
        if not args.no_single and world == 1:
            plan1 = H.plan_build(H.RAW, S, bits, stream)
            dplan1 = ctx.make_device_plan(plan1)
            d_out.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.decode_device(dplan1, d_in, d_out, stream_length=stream.size)
            b.record()
            torch.cuda.synchronize()
            assert ctx.status(dplan1) == 0 and torch.equal(d_out, d_ref)
            ms1 = a.elapsed_time(b)
            result["single_wavefront_no_plan"] = {"value": n / 2**20 / (ms1 * 1e-3), "unit": "MiB/s", "ms": ms1, "bit_exact": True,
                                                  "note": "raw format has no restart points: 1 wave64 = 1 dependent chain"}

        # the step before the path, informational: the same input through the GPU encoder (mt_ container,
        # 64 KiB independent blocks, sidecar plan built on the device) and the decode that plan enables
        if not args.no_single and world == 1:
            d_enc = torch.empty(H.capacity(H.MT, S, n), dtype=torch.uint8, device=dev)
            m, dplan_e = ctx.encode_device(H.MT, S, bits, d_ref, d_enc, block_size=1 << 16, index_interval=32, want_plan=True)  # warm-up
            t0 = time.perf_counter()
            m = ctx.encode_device(H.MT, S, bits, d_ref, d_enc, block_size=1 << 16)  # synchronises its stream
            t_gpu_enc = time.perf_counter() - t0
            d_out.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.decode_device(dplan_e, d_enc, d_out, stream_length=m)
            b.record()
            torch.cuda.synchronize()
            assert ctx.status(dplan_e) == 0 and torch.equal(d_out, d_ref)
            result["gpu_encoder"] = {"container": "mt_", "block_size": 1 << 16, "compressed_bytes": int(m), "encode_ms": t_gpu_enc * 1e3,
                                     "encode_GB_s": n / t_gpu_enc / 1e9, "decode_with_device_built_plan_MiB_s": n / 2**20 / (a.elapsed_time(b) * 1e-3),
                                     "round_trip_bit_exact": True}

    if rank == 0 and not args.no_cpu and world == 1:  # reported baseline, N=1 only
        result["cpu_baseline"] = cpu_baseline(stream, data, S, bits)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
