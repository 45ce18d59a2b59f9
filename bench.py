#!/usr/bin/env python3
"""Headline benchmark: bit-exact rANS32x64 16w 11-bit (raw) decode of a 100 MB enwik8-shaped stream on MI355X.

A "step" = one decode of one whole stream (compressed input and decoded output resident in HBM).  The timed loop rotates
over `--pairs` (default 4) DISTINCT (stream, output) buffer pairs, 657 MB in all: more than the 256 MiB Infinity Cache, so
every step reads its stream from and writes its output to HBM (the same pair replayed back to back — what round 1 timed —
is reported beside it as `roofline.warm`).  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank
decodes its own streams (weak scaling, no data-path collective: the streams are independent objects); the timed region is
bracketed by barrier + synchronize and the max over ranks is reported.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--index wave|G] [--size BYTES] [--bits B] [--pairs P] [--no-cpu]

`--workload sharded` is the strong-scaling form (BASELINE config 4): ONE 2^30-byte mt_ stream in 256 KiB blocks, its
chains sharded over the ranks with hsrans_plan_slice (device plans made once, outside the timed region), every rank
decodes its share from its window of the stream and the decoded ranges are exchanged over RCCL inside the timed region.
"""
from __future__ import annotations

import argparse
import glob
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
VALU_SLOT_NS = 1.9     # one wave64 VALU instruction of the loop's dominant (VOP3 / SGPR-operand) class per SIMD, measured
SIMDS = 1024           # 256 CUs x 4
COPY_PEAK_GBS = 6300.0 # what a device copy kernel reaches of the 8 TB/s (MI355X_MICROARCH.md)
SHADER_GHZ = 2.1       # shader clock under this load (per-wave s_memtime stamps against s_memrealtime: DESIGN.md 5)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(stream: np.ndarray, data: np.ndarray, states: int, bits: int, budget_s: float = 4.0) -> dict:
    """Times the CPU decoders on this host, rank 0 only: the REAL reference's AVX2 decoder named by the north star and its
    AVX-512 sibling where the host has AVX-512 (oracle/_ref, kind "reference"); else the scalar oracle restatement (kind
    "port").  Checker code only — nothing here is on the product path."""
    from oracle_lib import RAW, Oracle, Ref

    n = data.size
    if Ref.available():
        ref = Ref()
        variants = []
        if ref.L.hsref_has_avx2():
            variants.append((1, ("rANS32x64_xmmShfl2_16w_decode_avx2_varC_%d" if bits <= 12 else "rANS32x64_xmmShfl2_16w_decode_avx2_varA_%d") % bits))
        if ref.has_avx512():
            variants.append((3, ("rANS32x64_ymmShfl2_16w_decode_avx512_varC_%d" if bits <= 12 else "rANS32x64_ymmShfl2_16w_decode_avx512_varA_%d") % bits))
        if not variants:
            variants.append((0, "rANS32x64_16w_decode_scalar_%d" % bits))
        results = {}
        for variant, name in variants:
            best, r, out, runs = ref.timed_decode(RAW, states, bits, stream, n, variant=variant, budget_s=budget_s)
            assert r == n and np.array_equal(out, data), "reference CPU decoder output differs from the original data"
            results[name] = {"MiB_s": n / 2**20 / best, "runs": runs}
        fastest = max(results, key=lambda k: results[k]["MiB_s"])
        return {"value": results[fastest]["MiB_s"], "unit": "MiB/s", "cores": 1, "kind": "reference",
                "sample": f"whole {n}-byte stream, best of {results[fastest]['runs']} runs, {fastest} from oracle/_ref (real reference, clang -O3)",
                "decoders": results, "own_host_simd_port": _own_port(stream, data, RAW, states, bits, 1, budget_s), "cpu": _cpu_model(), "host_cores": os.cpu_count()}
    orc = Oracle()
    t0 = time.perf_counter()
    r, out = orc.decode(RAW, states, bits, stream, n)
    dt = time.perf_counter() - t0
    assert r == n and np.array_equal(out, data)
    return {"value": n / 2**20 / dt, "unit": "MiB/s", "cores": 1, "kind": "port",
            "sample": f"whole {n}-byte stream, 1 run, scalar oracle restatement (oracle/hsrans_oracle.c)", "cpu": _cpu_model(),
            "host_cores": os.cpu_count()}


def _own_port(stream, data, container, states, bits, threads, budget_s=4.0):
    """This library's own host SIMD decoder (csrc/hsrans_cpu.cpp, runtime dispatch) on the same stream: the `port` next to the
    reference calibration.  Not the product's GPU path — the CPU comparator and the single-chain route of the auto entries."""
    from hypersonic_rans_amd import api

    L = api.load_library()
    stream = np.ascontiguousarray(stream)
    out = np.full(data.size + 64, 0xCC, np.uint8)  # allocated and touched before the clock starts
    best, runs, t_total = None, 0, 0.0
    while runs < 3 or (t_total < budget_s and runs < 40):
        t0 = time.perf_counter()
        r = L.hsrans_decode_cpu(-1, threads, container, states, bits, api._p(stream), stream.size, api._p(out), data.size, None, 0)
        dt = time.perf_counter() - t0
        assert r == data.size
        if runs == 0:
            assert np.array_equal(out[:data.size], data)
        best = dt if best is None else min(best, dt)
        t_total += dt
        runs += 1
    return {"MiB_s": data.size / 2**20 / best, "level": api.CPU_LEVELS[api.cpu_level()], "threads": threads, "runs": runs, "kind": "port"}


def _mt_cpu_baseline(stream, data, states, bits):
    """mt_ stream on ALL host cores: the reference's mt_rANS32x64_16w_decode_mt_N on its thread pool (README.md:184 publishes
    16.2 GiB/s for it on a 7950X) and this library's host decoder on std::threads, same stream, bounded to a few seconds."""
    from oracle_lib import MT, Ref

    out = {"cpu": _cpu_model(), "host_cores": os.cpu_count()}
    threads = max(1, (os.cpu_count() or 2) - 1)  # main.cpp:165 sizes its pool the same way
    if Ref.available():
        ref = Ref()
        best, r, got, _runs = ref.timed_decode(MT, states, bits, stream, data.size, variant=2, threads=threads, runs=3, budget_s=3.0, max_runs=10)
        assert r == data.size and np.array_equal(got, data)
        out.update({"value": data.size / 2**20 / best, "unit": "MiB/s", "cores": int(ref.L.hsref_pool_threads()), "kind": "reference",
                    "sample": f"whole {data.size}-byte mt_ stream, best of 3, mt_rANS32x{states}_16w_decode_mt_{bits} from oracle/_ref on its thread pool"})
    out["own_host_simd_port"] = _own_port(stream, data, MT, states, bits, threads, budget_s=3.0)
    return out


def kernel_source_sha256() -> str:
    """SHA-256 over the decode kernels' sources (csrc/hsrans_kernels.hip and the headers of this repository it includes): the identity of the kernels a counter
    run was made with.  bench.py puts it into its line (config.kernel_source_sha256), tools/pmc_summary.py keeps that line next
    to the counters, and _pmc_profile refuses counters of other kernels."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "hypersonic_rans_amd", "csrc")
    files = [os.path.join(csrc, n) for n in ("hsrans_kernels.hip", "hsrans_kernels.h", "hsrans_plan.h")] + glob.glob(os.path.join(csrc, "kernels_*.h"))
    for f in sorted(files):  # (the decode kernels' translation unit and everything it includes of this repository; not the encoder, not the host code)
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _pmc_profile(n: int, states: int, bits: int, index: str, pairs: int = 0, units: int = 1):
    """The committed rocprofv3 PMC run of this very workload (profiles/*_pmc.json, written by tools/pmc_summary.py: FETCH_SIZE /
    WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  PMC counters cannot be collected
    from inside this process, so traffic is null unless such a run matches the workload AND was made with the kernels as they
    are now (config.kernel_source_sha256 of the line it recorded; VERDICT r3: a stale file must not be reported as this
    build's traffic).  Returns (path, json) of the match, or (path, None) for the newest run of this workload that is stale."""
    best, stale = None, None
    now = kernel_source_sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            j = json.load(open(f))
            cfg = j["bench_line_under_trace"]["config"]
            if (cfg["decoded_bytes"], cfg["states"], cfg["bits"], str(cfg.get("index", cfg.get("index_interval_groups"))), cfg.get("streams_per_launch", 1)) == (n, states, bits, index, units):
                if cfg.get("kernel_source_sha256") != now:
                    stale = (os.path.relpath(f, ROOT), None)
                    continue
                if best is None or cfg.get("pairs") == pairs:  # same rotation (cache state) preferred; counters carry FETCH_SIZE and WRITE_SIZE
                    if "hbm_traffic_bytes_per_launch" in j or best is None:
                        best = (os.path.relpath(f, ROOT), j)
        except (KeyError, ValueError, OSError):
            continue
    return best or stale


def _permuted(data: np.ndarray, k: int) -> np.ndarray:
    """Another stream of the same shape: the bytes of `data` under a fixed byte permutation (same order-0 statistics, other content)."""
    if k == 0:
        return data
    return synth._permutation(1000 + k)[data]


def headline(args, world, rank, dev, dev_index, ctx, dist):
    n, S, bits = args.size, args.states, args.bits
    # BASELINE config 2's generator for the 100 MB workload; multi-GiB runs (tools/profile.sh) tile it (the generator makes ~7 MB/s)
    base = synth.enwik8_shaped(n, seed=20241008 + rank) if n <= (1 << 28) else _tiled(n, seed=20241008 + rank)
    pairs = []
    t_enc = t_setup = 0.0
    # the one-chain-per-wave index is shaped for THIS device: hsrans_ctx_calibrate fits the chain lengths of the 8 wave classes
    # (about half a second, outside the timed region; the lengths used are in config.launch.class_weights either way)
    calibration = None
    if args.index == "wave" and S == 64 and bits <= 12 and not args.no_calibrate:
        try:
            calibration = ctx.calibrate(bits=bits)
            # ... and at the run lengths of this workload (the fit above is for runs of ~96 groups): a stream decoded alone, and
            # the P streams decoded by one launch (mib_s.independent_streams_one_launch)
            run = n / S / 8192.0
            calibration["runs"] = [ctx.calibrate_runs(bits=bits, copies=c) for c in sorted({min(16, max(2, round(run / 96))), min(16, max(2, round(max(1, args.pairs) * run / 96)))})]
        except H.HsransError as e:  # not the launch shape the classes are defined for (another device geometry): compiled-in lengths
            calibration = {"skipped": str(e)}
    groups = None if args.index != "wave" else H.index_boundaries(S, bits, n, ctx)
    for k in range(max(1, args.pairs)):
        data = _permuted(base, k)
        t0 = time.perf_counter()
        if args.index == "wave":
            stream, plan = H.encode(H.RAW, S, bits, data, index_groups=groups)
        else:
            stream, plan = H.encode(H.RAW, S, bits, data, index_interval=int(args.index))
        t_enc += time.perf_counter() - t0
        pad = (-stream.size) % 16
        t0 = time.perf_counter()
        dplan = ctx.make_device_plan(plan)  # hsrans_dplan_create: plan validation, host-built decode table, upload (synchronous)
        t_setup += time.perf_counter() - t0
        p = {"data": data, "stream": stream, "plan": plan,
             "d_in": torch.from_numpy(np.concatenate([stream, np.zeros(pad, np.uint8)])).to(dev),
             "d_out": torch.zeros(n, dtype=torch.uint8, device=dev), "dplan": dplan}
        pairs.append(p)
    P = len(pairs)
    chains = H.plan_chain_count(pairs[0]["plan"])

    # --one-launch: a step is ONE launch that decodes all P streams (hsrans_decode_device_batch; the streams' indexes shaped for that
    # launch, hsrans_index_boundaries_batch) — the profiling form of mib_s.independent_streams_one_launch (tools/profile.sh r05_batch --one-launch)
    units = 1
    one_batch = None
    if args.one_launch:
        assert S == 64 and bits <= 12 and P >= 2, "--one-launch: 64 states, 10..12 bits, at least two streams"
        shaped = [ctx.make_device_plan(ctx.index_build_at(H.RAW, S, bits, p["stream"], H.index_boundaries_batch(S, bits, [n] * P, k, ctx))) for k, p in enumerate(pairs)]
        one_batch = ctx.make_batch(shaped)
        units = P
        for p, d in zip(pairs, shaped):
            p["dplan"] = d
        b_in0, b_out0, b_len0 = [p["d_in"] for p in pairs], [p["d_out"] for p in pairs], [p["stream"].size for p in pairs]

    def step(i):
        if one_batch is not None:
            ctx.decode_device_batch(one_batch, b_in0, b_out0, stream_lengths=b_len0)
            return
        p = pairs[i % P]
        ctx.decode_device(p["dplan"], p["d_in"], p["d_out"], stream_length=p["stream"].size)

    # validation decode of every pair (not one of the W warm-up steps): the checks idle the GPU, so they come before the warm-up
    for i, p in enumerate(pairs):
        step(i)
        torch.cuda.synchronize()
        assert ctx.status(p["dplan"]) == 0
        assert np.array_equal(p["d_out"].cpu().numpy(), p["data"]), "GPU output is not bit-exact"
        p["d_out"].zero_()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ------------------------------------------
    # One HIP event pair brackets the K launches ON THE LAUNCH STREAM (hsrans_decode_device launches on torch's current
    # stream, which is where torch.cuda.Event records): span / K = the kernel's average launch duration for the roofline.
    # The region — W warm-up steps, barrier + synchronize, EXACTLY K timed steps, synchronize + barrier — runs R = --repeats
    # times in this process (one 0.9 ms window cannot resolve the 2-5 % a kernel change is worth; the reference's
    # harness reports min / mean / sigma over its runs too, src/main.cpp:72-118).  `value` is the MEDIAN region; the spread goes
    # into mib_s.  With several ranks every region's time is the maximum over the ranks.
    def timed_region(first_step):
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(args.warmup):  # W untimed warm-up steps, immediately in front of the timed region
            step(first_step + i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_a.record()
        for i in range(args.steps):
            step(first_step + args.warmup + i)
        ev_b.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ev_a.elapsed_time(ev_b) / args.steps

    # Settle first (untimed, after the validation and before the first region): the validation above copied 4 x 100 MB to the host
    # and compared them there, the GPU idled meanwhile, and a GPU that wakes from idle does not run at its sustained state at
    # once — measured with this very kernel (tools/settle_probe.py, profiles/r04_settle.txt): the first ~1 ms after the idle
    # gap reads 41 us per decode, the next 5-8 ms read 47-52 us, and only after about 12 ms of back-to-back launches the time is at
    # the 40-43 us it then keeps.  A production decoder is in that last state; --settle-ms 0 measures the wake-up instead.
    # (the method changed between rounds 3 and 4 — one region straight after the validation's idle gap then, a settled median of
    # 15 regions now — so the old method's figure is taken too, first, and reported beside the headline: mib_s.old_method_one_region_no_settle)
    old_method = timed_region(0) if (args.settle_ms > 0 and not args.timed_only) else None
    settle_launches = 0
    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for i in range(64):
                step(settle_launches + i)
            settle_launches += 64
            torch.cuda.synchronize()
    regions = [timed_region(r * (args.warmup + args.steps)) for r in range(max(1, args.repeats))]
    region_s = np.array([r[0] for r in regions])
    region_span_ms = np.array([r[1] for r in regions])
    elapsed = float(np.median(region_s))
    kernel_ms_span = float(np.median(region_span_ms))

    if args.timed_only:
        # tools/profile.sh: nothing but the rotation is launched, so that the rocprofv3 per-kernel average IS the timed region's
        warm_ms, warm_each, kernel_ms = float("nan"), [], [float(region_span_ms.min())]
    else:
        # warm: ONE pair replayed back to back (stream + output + index stay in the Infinity Cache); outside the timed region.
        # Every pair in turn: the replayed time depends on where the pair's buffers landed physically (measured 40.8 / 45.1 /
        # 42.2 / 41.2 us for the four pairs of one process), so the mean over the pairs is reported, and the best beside it.
        warm_each = []
        for k in range(P):
            wa, wb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                step(k)
            wa.record()
            for _ in range(args.steps):
                step(k)
            wb.record()
            torch.cuda.synchronize()
            warm_each.append(wa.elapsed_time(wb) / args.steps)
        warm_ms = float(np.mean(warm_each))
        # per-launch spread over the rotation (each event pair adds launch latency, so only min/max are reported)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 20))]
        for i, (a, b) in enumerate(ev):
            a.record()
            step(i)
            b.record()
        torch.cuda.synchronize()
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
    shas = []
    for p in pairs:
        assert ctx.status(p["dplan"]) == 0
        got = p["d_out"].cpu().numpy()
        assert np.array_equal(got, p["data"]), "GPU output is not bit-exact after the timed region"
        shas.append(hashlib.sha256(got.tobytes()).hexdigest())
        assert shas[-1] == hashlib.sha256(p["data"].tobytes()).hexdigest()

    # the P streams as INDEPENDENT work in flight together (SURVEY.md §7 hard part 1 asks for "N independent streams" beside "1 stream
    # + index"): the same K steps, launched alternately on two HIP streams, so that one launch's prologue and tail overlap its
    # neighbour's decode.  Not the headline: a step there is one launch with the GPU to itself.
    overlapped_ms = None
    if not args.timed_only and not args.one_launch and world == 1:
        side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        samples = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2 * args.steps):
                p = pairs[i % P]
                ctx.decode_device(p["dplan"], p["d_in"], p["d_out"], stream=side[i % 2], stream_length=p["stream"].size)
            torch.cuda.synchronize()
            samples.append((time.perf_counter() - t0) / (2 * args.steps) * 1e3)
        overlapped_ms = float(np.median(samples))
    # ... and as ONE launch (hsrans_decode_device_batch): the device's wave slots dealt to the P streams — whole workgroups, every
    # stream's chains cut into runs by the slots' age class — so that prologue, tail and kernel boundary are paid once for all P.
    # A step here is one launch that decodes all P streams (657 MB of stream + output: cold by construction).
    one_launch = None
    if not args.timed_only and not args.one_launch and world == 1 and P >= 2 and S == 64 and bits <= 12:
        b_in, b_out, b_len = [p["d_in"] for p in pairs], [p["d_out"] for p in pairs], [p["stream"].size for p in pairs]

        def batch_leg(dplans):
            batch = ctx.make_batch(dplans)
            for p in pairs:
                p["d_out"].zero_()
            ctx.decode_device_batch(batch, b_in, b_out, stream_lengths=b_len)
            torch.cuda.synchronize()
            assert ctx.batch_status(batch) == [0] * P
            for p in pairs:
                assert np.array_equal(p["d_out"].cpu().numpy(), p["data"]), "batch launch: GPU output is not bit-exact"
            launches = max(4, args.steps // P)
            # (the checks above idled the GPU: settled like the headline, but three times as long — this launch moves 657 MB and its samples
            # kept falling through the first ~55 ms of sustained launches, 161 -> 142 us in profiles/r05_bench_line.json's first take, where
            # the one-stream launch is steady after 30)
            t_settle = time.perf_counter()
            while (time.perf_counter() - t_settle) * 1e3 < max(3 * args.settle_ms, 1.0):
                for _ in range(launches):
                    ctx.decode_device_batch(batch, b_in, b_out, stream_lengths=b_len)
                torch.cuda.synchronize()
            samples = []
            for _ in range(max(5, args.repeats)):
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                for _ in range(launches):
                    ctx.decode_device_batch(batch, b_in, b_out, stream_lengths=b_len)
                eb.record()
                torch.cuda.synchronize()
                samples.append(ea.elapsed_time(eb) / launches)
            assert ctx.batch_status(batch) == [0] * P
            for p in pairs:
                assert np.array_equal(p["d_out"].cpu().numpy(), p["data"]), "batch launch: GPU output is not bit-exact after the timed launches"
            ms_batch = float(np.median(samples))
            return {"ms_per_launch": ms_batch, "ms_per_stream": ms_batch / P, "streams": P, "samples_ms_per_launch": [float(x) for x in samples],
                    "launches_per_sample": launches, "batch": batch.info()}

        # the streams' sidecars shaped for THIS launch, as the headline's is for its own (hsrans_index_boundaries_batch: one chain per
        # wave slot the batch deals the stream — a quarter of the headline's index for four streams; stream bytes unchanged) ...
        t0 = time.perf_counter()
        shaped = []
        for k, p in enumerate(pairs):
            bplan = ctx.index_build_at(H.RAW, S, bits, p["stream"], H.index_boundaries_batch(S, bits, [n] * P, k, ctx))
            shaped.append((bplan.size, ctx.make_device_plan(bplan)))
        t_shaped = time.perf_counter() - t0
        one_launch = batch_leg([d for _, d in shaped])
        one_launch["index"] = {"kind": "hsrans_index_boundaries_batch", "plan_bytes_per_stream": int(shaped[0][0]), "chains_per_stream": H.plan_chain_count(bplan),
                               "index_build_ms_per_stream": t_shaped / P * 1e3}
        # ... the same streams SUBMITTED ONE BY ONE to a queue (hsrans_queue: a caller that meets its streams one after the other — the reference's
        # loop over files, src/main.cpp:841-898 — and does not know K up front): every fourth submission flushes, the flush finds the batch it made
        # the first time around.  Host-side submission inside the timed span; same kernels, same indexes as the leg above.
        queue = H.api.Queue(ctx, max_members=P)
        shaped_plans = [d for _, d in shaped]

        def queued_round():
            for k, p in enumerate(pairs):
                queue.submit(shaped_plans[k], p["d_in"], p["d_out"], stream_length=p["stream"].size)

        for p in pairs:
            p["d_out"].zero_()
        queued_round()
        torch.cuda.synchronize()
        for k, p in enumerate(pairs):
            assert ctx.status(shaped_plans[k]) == 0 and np.array_equal(p["d_out"].cpu().numpy(), p["data"]), "queued launch: GPU output is not bit-exact"
        rounds = max(4, args.steps // P)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < max(3 * args.settle_ms, 1.0):
            for _ in range(rounds):
                queued_round()
            torch.cuda.synchronize()
        q_samples = []
        for _ in range(max(5, args.repeats)):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            for _ in range(rounds):
                queued_round()
            eb.record()
            torch.cuda.synchronize()
            q_samples.append(ea.elapsed_time(eb) / rounds)
        for k, p in enumerate(pairs):
            assert ctx.status(shaped_plans[k]) == 0 and np.array_equal(p["d_out"].cpu().numpy(), p["data"]), "queued launch: GPU output is not bit-exact after the timed launches"
        q_ms = float(np.median(q_samples))
        one_launch["queued"] = {"ms_per_stream": q_ms / P, "frac_of_hbm_peak": (int(np.mean(b_len)) + n) / (q_ms / P * 1e-3) / 1e9 / HBM_PEAK_GBS, "queue": queue.stats(),
                                "note": f"the {P} streams submitted one by one to an hsrans_queue (max_members = {P}: the last submission flushes); the flush reuses the batch it made first"}
        queue.close()
        # ... and with the sidecars the streams already have (made for a launch of their own)
        if args.index == "wave":
            own = batch_leg([p["dplan"] for p in pairs])
            one_launch["with_the_streams_own_indexes"] = {"ms_per_stream": own["ms_per_stream"], "frac_of_hbm_peak": (int(np.mean(b_len)) + n) / (own["ms_per_stream"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                          "batch": own["batch"]}
    if rank != 0:
        return None

    info = pairs[0]["dplan"].launch_info()
    stream, plan = pairs[0]["stream"], pairs[0]["plan"]
    ms_per_step = elapsed * 1e3 / args.steps
    k_avg = float(kernel_ms_span)
    to_mib_s = lambda seconds: float(world * units * n / 2**20 / seconds * args.steps)
    alg_bytes = units * (int(np.mean([p["stream"].size for p in pairs])) + n)  # compressed bytes read once + decoded bytes written once (--one-launch: of all P streams)
    achieved = alg_bytes / (k_avg * 1e-3) / 1e9
    groups_per_launch = units * n // S
    prof = _pmc_profile(n, S, bits, args.index, P, units)
    traffic, traffic_source, issue, traffic_stale = None, None, None, None
    if prof is not None and prof[1] is None:
        traffic_stale = prof[0]  # counters of this workload exist, but of kernels that have changed since: not reported
    elif prof is not None:
        traffic_source, j = prof
        traffic = float(j["hbm_traffic_bytes_per_launch"]["total"])
        valu = j["counters"].get("SQ_INSTS_VALU", {}).get("mean")
        if valu:
            per_group = valu / groups_per_launch
            issue = {"valu_per_group": per_group, "slot_ns": VALU_SLOT_NS, "simds": SIMDS, "us": per_group * VALU_SLOT_NS * groups_per_launch / SIMDS * 1e-3,
                     "note": "VALU instructions per 64-symbol group (SQ_INSTS_VALU / groups) x 1.9 ns per issue slot x groups / 1,024 SIMDs: the time the "
                             "vector issue alone needs; counters from " + traffic_source}
    lds_bound = None
    if prof is not None and prof[1] is not None:
        lds = prof[1]["counters"].get("SQ_LDS_IDX_ACTIVE", {}).get("mean")
        conf = prof[1]["counters"].get("SQ_LDS_BANK_CONFLICT", {}).get("mean")
        if lds:
            lds_bound = {"lds_cycles_per_group_and_cu": lds / groups_per_launch, "bank_conflict_share": (conf / lds) if conf else None, "cus": SIMDS // 4, "clock_ghz": SHADER_GHZ,
                         "us": lds / (SIMDS // 4) / (SHADER_GHZ * 1e3),
                         "note": "LDS-array cycles (SQ_LDS_IDX_ACTIVE, summed over the CUs) / 256 CUs / 2.1 GHz: the time the CUs' LDS pipelines alone need for the table gathers and word "
                                 "reads of one launch — with issue_bound the two units that actually bind this kernel; the HBM roofline does not (DESIGN.md 5)"}
    result = {
        "metric": "decode MiB/s (bit-exact) on 100 MB stream",
        "value": world * units * n / 2**20 / elapsed * args.steps,
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
            "workload": f"rANS32x{S} 16w {bits}-bit (raw + sidecar index) decode, {n} B enwik8-shaped synthetic (Zipf1.2/205 symbols, seed 20241008+rank), "
                        f"{P} distinct streams per GPU rotated through the timed loop ({P * (alg_bytes) / 2**20:.0f} MiB of stream + output: beyond the Infinity Cache), "
                        + ("index with one chain per resident wavefront" if args.index == "wave" else f"index with a checkpoint every {args.index} groups")
                        + f" ({chains} chains)" + (f"; --one-launch: a step is ONE launch decoding all {P} streams (hsrans_decode_device_batch)" if args.one_launch else ""),
            "container": "raw", "states": S, "bits": bits, "decoded_bytes": n, "compressed_bytes": int(stream.size),
            "ratio": stream.size / n, "index": args.index, "plan_bytes": int(plan.size), "plan_over_compressed": plan.size / stream.size,
            "effective_ratio_with_index": (stream.size + plan.size) / n, "chains": chains, "pairs": P, "streams_per_launch": units,
            "batch": one_batch.info() if one_batch is not None else None,
            "launch": info, "bit_exact": True, "sha256": shas[0], "host_encode_s": t_enc / P,
            # once per (stream, plan), outside the timed region: hsrans_dplan_create = plan validation + host-built table + upload of the index
            "plan_setup_ms": t_setup / P * 1e3,
            "calibration": calibration,
            "kernel_source_sha256": kernel_source_sha256(),
            "settle": {"ms": args.settle_ms, "launches": settle_launches,
                       "note": "untimed launches of the same rotation between the validation (which idles the GPU) and the first timed region: the sustained state, not the wake-up"},
        },
        # the reference harness prints min/mean throughput per decoder (src/main.cpp:72-118): same two numbers here
        "mib_s": {"median": to_mib_s(np.median(region_s)), "min": to_mib_s(region_s.max()), "max": to_mib_s(region_s.min()), "p10": to_mib_s(np.percentile(region_s, 90)),
                  "p90": to_mib_s(np.percentile(region_s, 10)), "mean": to_mib_s(region_s.mean()), "repeats": int(region_s.size),
                  "per_repeat": [to_mib_s(x) for x in region_s], "ms_per_step_per_repeat": [float(x / args.steps * 1e3) for x in region_s],
                  "note": "each repeat = W warm-up steps + barrier/synchronize + K timed steps + synchronize/barrier; `value` is the median repeat",
                  "old_method_one_region_no_settle": None if old_method is None else
                  {"value": to_mib_s(old_method[0]), "ms_per_step": old_method[0] / args.steps * 1e3,
                   "note": "rounds 1-3 measured this: ONE region (W warm-up + K timed steps) straight after the validation's idle gap, no settling, no repeats"},
                  "best_single_launch": units * n / 2**20 / (float(np.min(kernel_ms)) * 1e-3),
                  "warm_one_pair_replayed": units * n / 2**20 / (warm_ms * 1e-3), "launches_in_timed_region": args.steps,
                  # "1 stream, no index" = single_wavefront_no_plan below, "1 stream + index" = value, and "N independent streams":
                  "independent_streams_overlapped": None if overlapped_ms is None else
                  {"value": n / 2**20 / (overlapped_ms * 1e-3), "ms_per_stream": overlapped_ms, "frac_of_hbm_peak": alg_bytes / (overlapped_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "note": f"the same {P} streams as independent work: launches alternate between two HIP streams, so a launch's prologue and tail overlap its "
                           "neighbour's decode (wall clock over 2K launches, median of 5); a step of the headline has the GPU to itself"},
                  "independent_streams_queued": None if one_launch is None or "queued" not in one_launch else
                  dict(one_launch["queued"], value=n / 2**20 / (one_launch["queued"]["ms_per_stream"] * 1e-3)),
                  "independent_streams_one_launch": None if one_launch is None else
                  dict(one_launch, value=n / 2**20 / (one_launch["ms_per_stream"] * 1e-3), frac_of_hbm_peak=alg_bytes / (one_launch["ms_per_stream"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       note=f"the same {P} streams decoded by ONE launch (hsrans_decode_device_batch: wave slots dealt to the streams, prologue / tail / kernel "
                            "boundary paid once); HIP events around the launches of a sample, median sample; value = decoded MiB/s per GPU")},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # the same fraction from the wall clock of the timed region (B_alg / ms_per_step): what the driver's number implies
            "frac_wall": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": (traffic_source + " (rocprofv3 --pmc run of this workload with these kernels, not this process)") if traffic_source else None,
            "traffic_stale": (traffic_stale + ": counters of this workload, but the kernel sources have changed since (sha-256 mismatch): not reported") if traffic_stale else None,
            "kernel_ms_avg_spread": {"min": float(region_span_ms.min()), "max": float(region_span_ms.max()), "p10": float(np.percentile(region_span_ms, 10)),
                                     "p90": float(np.percentile(region_span_ms, 90)), "repeats": int(region_span_ms.size)},
            "algorithmic_bytes_per_launch": alg_bytes, "index_bytes_read_per_launch": int(plan.size),
            "kernel_ms_avg": k_avg, "kernel_ms_single_launch_min": float(np.min(kernel_ms)),
            "cache_state": f"cold: {P} (stream, output) pairs rotated, working set {P * alg_bytes / 2**20:.0f} MiB > 256 MiB Infinity Cache",
            "warm": {"kernel_ms_avg": float(warm_ms), "frac": alg_bytes / (warm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms_per_pair": [float(x) for x in warm_each],
                     "cache_state": "one pair replayed back to back (Infinity-Cache resident, as BENCH_r01 measured)"},
            "issue_bound": issue,
            "lds_bound": lds_bound, "lds_bound_us": None if lds_bound is None else lds_bound["us"],
            # against what a plain copy kernel reaches on this part (~6.3 TB/s, MI355X_MICROARCH.md) instead of the 8 TB/s pin rate
            "frac_of_copy_peak": achieved / COPY_PEAK_GBS,
            "binds": "vector issue + LDS gather (issue_bound / lds_bound), not HBM: traffic is ~1.03 x algorithmic and the kernel time is ~2 x the HBM time",
            "kernel": "hsrans::k_decode_batch<3>" if args.one_launch else ("hsrans::k_decode_dual<%d>" % info["table_mode"]) if info["chains_per_wave"] == 2 else
                      ("hsrans::k_decode_direct<%d>" % info["table_mode"]) if args.index == "wave" else
                      "hsrans::k_decode<%d, %s>" % (info["table_mode"], "true" if info["shared_table"] else "false"),
        },
    }

    # the same stream WITHOUT the sidecar index: one wavefront, one dependent chain (SURVEY.md finding 2)
    d_in, d_out, d_ref = pairs[0]["d_in"], pairs[0]["d_out"], torch.from_numpy(pairs[0]["data"]).to(dev)
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
        # ... and the raw format (this workload's own container) through the one coding wavefront the format has work for: the
        # stream it writes is the host encoder's byte for byte, so is the one-chain-per-wavefront index it records on the way
        if S == 64 and n <= (1 << 30):
            d_raw = torch.empty(H.capacity(H.RAW, S, n), dtype=torch.uint8, device=dev)
            t0 = time.perf_counter()
            kw = {"index_groups": groups} if groups is not None else {"index_interval": int(args.index)}
            m_raw, plan_raw = ctx.encode_device_raw(S, bits, d_ref, d_raw, want_plan=True, **kw)
            t_raw = time.perf_counter() - t0
            same = m_raw == pairs[0]["stream"].size and bool(np.array_equal(d_raw[:m_raw].cpu().numpy(), pairs[0]["stream"])) and \
                bool(np.array_equal(plan_raw, pairs[0]["plan"]))
            result["gpu_encoder"]["raw_format"] = {"encode_ms": t_raw * 1e3, "encode_GB_s": n / t_raw / 1e9, "stream_and_index_identical_to_host_encoder": same,
                                                   "note": "one dependent chain per coder state: one wavefront"}

    if not args.no_cpu and world == 1:  # reported baseline, N=1 only
        result["cpu_baseline"] = cpu_baseline(stream, pairs[0]["data"], S, bits)
    # the other BASELINE configurations' 1-GPU legs, in the same driver-timed run (after everything the headline measures: they re-use and then
    # release its buffers)
    if world == 1 and not args.no_configs and not args.timed_only and not args.one_launch and S == 64 and bits == 11 and n == 100_000_000:
        t0 = time.perf_counter()
        result["configs"] = config_legs(args, ctx, dev, pairs, n)
        result["configs_wall_s"] = time.perf_counter() - t0
        c4 = next((c for c in result["configs"] if c.get("config") == 4 and "MiB_s" in c), None)
        if c4 is not None:  # (the N > 1 line is another workload — this is the N = 1 value to set it against)
            result["sharded_workload_one_gpu"] = {"MiB_s": c4["MiB_s"], "us": c4["us"], "frac_of_hbm_peak": c4["frac_of_hbm_peak"],
                                                  "note": "`bench.py --gpus N` (N > 1) decodes ONE 2^30-byte mt_ stream sharded over N ranks: compare its `value` with THIS figure, not with the headline's"}
    return result


def config_legs(args, ctx, dev, pairs, n: int, budget_s: float = 90.0) -> list:
    """The 1-GPU legs of BASELINE.json's OTHER configurations, timed in the driver's own run of this file (until round 6 only the
    headline was driver-witnessed, twenty figures were builder-run files under profiles/).  The reference's harness walks every codec over one
    input in one process and prints a line each (src/main.cpp:841-898); this walks the configurations:
      config 3   rANS32x64 16w raw at 12 / 14 / 15 bits on the headline's own four 100 MB inputs: one launch per stream (rotated), and the four
                 streams in ONE launch (hsrans_decode_device_batch); at 12 bits also with the decode table left in global memory
                 ("LDS-resident vs spilled table": HSRANS_TABLE_SPILL=1 is read when a device plan is made)
      config 4   ONE 2^30-byte mt_ stream in 256 KiB blocks (+ a checkpoint every --configs-interval groups), GPU-encoded, decode only — what
                 `--gpus N` shards; this is the N = 1 figure of that workload
      config 5   the same 2^30 bytes from / to page-locked HOST memory through the C ABI's pipeline (PCIe-inclusive, marked so; one GPU)
    Every leg: streams written by the GPU encoders (byte-identical to the host encoder's and, raw, to the reference's: tests/), output
    compared with the source bytes before and after the timed launches, HIP events on the launch stream, the median sample behind a
    short settle.  Fractions are of 8 TB/s on the algorithmic bytes (compressed read once + decoded written once).  Stops adding legs when
    `budget_s` is spent (a leg that did not run is listed as skipped, never silently missing)."""
    t_start = time.perf_counter()
    legs = []
    S = 64
    P = len(pairs)
    d_src = [torch.from_numpy(p["data"]).to(dev) for p in pairs]

    def timed(fn, launches, samples=5, settle_ms=20.0):
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < settle_ms:
            for _ in range(launches):
                fn()
            torch.cuda.synchronize()
        out = []
        for _ in range(samples):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(launches):
                fn()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / launches)
        return float(np.median(out)), [float(x) for x in out]

    def kernel_name(info, batch=False):
        if batch:
            return "hsrans::k_decode_batch_dual<%d>" % info if isinstance(info, int) else "hsrans::k_decode_batch<3>"
        if info["chains_per_wave"] == 2:
            return "hsrans::k_decode_dual<%d>" % info["table_mode"]
        return "hsrans::k_decode_direct<%d>" % info["table_mode"]

    def raw_streams(bits, groups_of):
        """the four inputs as raw streams of this width with the given index (GPU raw encoder: one wavefront, ~0.1 s per 100 MB)"""
        out = []
        for k in range(P):
            d_raw = torch.empty(H.capacity(H.RAW, S, n) + 16, dtype=torch.uint8, device=dev)
            m, dplan = ctx.encode_device_raw(S, bits, d_src[k], d_raw, index_groups=groups_of(k), want_device_plan=True)
            out.append((d_raw, m, dplan))
        return out

    def check(outs):
        return all(bool(torch.equal(o, d)) for o, d in zip(outs, d_src))

    outs = [p["d_out"] for p in pairs]
    for bits in (12, 14, 15):
        if time.perf_counter() - t_start > budget_s:
            legs.append({"config": 3, "bits": bits, "skipped": "time budget"})
            continue
        try:
            groups = H.index_boundaries(S, bits, n, ctx)
            st = raw_streams(bits, lambda k: groups)
            k = [0]

            def one():
                i = k[0] % P
                k[0] += 1
                ctx.decode_device(st[i][2], st[i][0], outs[i], stream_length=st[i][1])

            for o in outs:
                o.zero_()
            for _ in range(P):
                one()
            torch.cuda.synchronize()
            ok = check(outs) and all(ctx.status(x[2]) == 0 for x in st)
            ms, samples = timed(one, 4 * P)
            ok = ok and check(outs)
            alg = int(np.mean([x[1] for x in st])) + n
            info = st[0][2].launch_info()
            leg = {"config": 3, "workload": f"rANS32x64 16w {bits}-bit raw, {n} B, {P} streams rotated, one launch per stream, one chain per resident wavefront",
                   "bits": bits, "us_per_stream": ms * 1e3, "MiB_s": n / 2**20 / (ms * 1e-3), "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "kernel": kernel_name(info), "table": "LDS", "compressed_bytes": int(st[0][1]), "bit_exact": bool(ok), "samples_ms": samples}
            # the four streams in ONE launch, indexes shaped for it
            sb = raw_streams(bits, lambda k: H.index_boundaries_batch(S, bits, [n] * P, k, ctx))
            batch = ctx.make_batch([x[2] for x in sb])
            b_in, b_len = [x[0] for x in sb], [x[1] for x in sb]
            for o in outs:
                o.zero_()
            ctx.decode_device_batch(batch, b_in, outs, stream_lengths=b_len)
            torch.cuda.synchronize()
            okb = check(outs) and ctx.batch_status(batch) == [0] * P
            msb, _ = timed(lambda: ctx.decode_device_batch(batch, b_in, outs, stream_lengths=b_len), 6)
            okb = okb and check(outs)
            leg["four_streams_one_launch"] = {"us_per_stream": msb / P * 1e3, "frac_of_hbm_peak": alg / (msb / P * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches": batch.info()["launches"],
                                              "kernel": "hsrans::k_decode_batch<3>" if bits <= 12 else "hsrans::k_decode_batch_dual<%d>" % (3 if bits == 13 else 4), "bit_exact": bool(okb)}
            if bits == 12:  # LDS-resident vs spilled table, same streams, same process
                os.environ["HSRANS_TABLE_SPILL"] = "1"
                try:
                    sp = [ctx.make_device_plan(ctx.read_device_plan(x[2], capacity=1 << 26)) for x in st]
                finally:
                    os.environ.pop("HSRANS_TABLE_SPILL", None)
                j = [0]

                def spilled():
                    i = j[0] % P
                    j[0] += 1
                    ctx.decode_device(sp[i], st[i][0], outs[i], stream_length=st[i][1])

                for o in outs:
                    o.zero_()
                for _ in range(P):
                    spilled()
                torch.cuda.synchronize()
                oks = check(outs) and all(ctx.status(x) == 0 for x in sp)
                mss, _ = timed(spilled, P, samples=3, settle_ms=5.0)
                leg["table_in_global_memory"] = {"us_per_stream": mss * 1e3, "frac_of_hbm_peak": alg / (mss * 1e-3) / 1e9 / HBM_PEAK_GBS, "slower_by": mss / ms,
                                                 "kernel": "hsrans::k_decode_direct<5>", "table_mode": sp[0].launch_info()["table_mode"], "bit_exact": bool(oks)}
                del sp
            legs.append(leg)
            del st, sb, batch
        except Exception as e:  # a leg that fails is reported, the headline stands
            legs.append({"config": 3, "bits": bits, "failed": repr(e)})
        torch.cuda.empty_cache()

    # ---- configs 4 and 5: ONE 2^30-byte mt_ stream ------------------------------------------------------------------------------------
    big = 1 << 30
    if time.perf_counter() - t_start > budget_s:
        legs += [{"config": 4, "skipped": "time budget"}, {"config": 5, "skipped": "time budget"}]
        return legs
    try:
        d_big = torch.cat(d_src * ((big + P * n - 1) // (P * n)))[:big].contiguous()  # the four inputs (four byte permutations) tiled: every 100 MB another histogram
        del d_src
        enc = torch.empty(H.capacity(H.MT, S, big), dtype=torch.uint8, device=dev)
        t0 = time.perf_counter()
        m, dplan = ctx.encode_device(H.MT, S, 11, d_big, enc, block_size=1 << 18, index_interval=args.configs_interval, want_plan=True)
        t_enc = time.perf_counter() - t0
        out = torch.zeros(big, dtype=torch.uint8, device=dev)
        ctx.decode_device(dplan, enc, out, stream_length=m)
        torch.cuda.synchronize()
        ok = ctx.status(dplan) == 0 and bool(torch.equal(out, d_big))
        ms, samples = timed(lambda: ctx.decode_device(dplan, enc, out, stream_length=m), 8)
        ok = ok and bool(torch.equal(out, d_big))
        info = dplan.launch_info()
        legs.append({"config": 4, "workload": f"mt_rANS32x64 16w 11-bit, ONE {big}-byte stream in 256 KiB blocks + a checkpoint every {args.configs_interval} groups, GPU-encoded, "
                                              "decode only, one GPU: the N = 1 figure of `bench.py --gpus N` (which shards this stream over N ranks)",
                     "us": ms * 1e3, "MiB_s": big / 2**20 / (ms * 1e-3), "frac_of_hbm_peak": (m + big) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "compressed_bytes": int(m),
                     "kernel": "hsrans::k_decode_dealt" if info["spread"] == 2 else "hsrans::k_decode_spread<3>" if info["spread"] == 1 else "hsrans::k_decode_grouped<%d, true>" % info["table_mode"],
                     "launch": {k: info[k] for k in ("grid", "block", "lds_bytes", "chains", "dynamic_groups", "spread")}, "gpu_encode_ms": t_enc * 1e3, "bit_exact": bool(ok), "samples_ms": samples})
    except Exception as e:
        legs.append({"config": 4, "failed": repr(e)})
        return legs
    if time.perf_counter() - t_start > budget_s:
        legs.append({"config": 5, "skipped": "time budget"})
        return legs
    try:
        from hypersonic_rans_amd import pipeline
        plan = ctx.read_device_plan(dplan, capacity=1 << 30)
        host_stream = torch.empty(m, dtype=torch.uint8).pin_memory()
        host_stream.copy_(enc[:m])
        host_ref = d_big.cpu()
        del enc, out, dplan, d_big
        torch.cuda.empty_cache()
        host_out = torch.empty(big, dtype=torch.uint8).pin_memory()
        dec = pipeline.PipelinedHostDecoder(ctx, plan)
        dec.decode(host_stream, host_out)
        ok = bool(torch.equal(host_out, host_ref))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            dec.decode(host_stream, host_out)  # synchronous: the decoded bytes are in host memory on return
            ts.append(time.perf_counter() - t0)
        ok = ok and bool(torch.equal(host_out, host_ref))
        t = float(np.median(ts))
        legs.append({"config": 5, "workload": f"the same {big}-byte mt_ stream from page-locked host memory into page-locked host memory: hsrans_hpipe (upload, decode and download of "
                                              "consecutive slices overlapped on three HIP streams), one GPU",
                     "pcie_inclusive": True, "ms": t * 1e3, "decoded_GB_s": big / t / 1e9, "MiB_s": big / 2**20 / t, "frac_of_pcie_d2h_63GBs": big / t / 1e9 / 63.0, "bit_exact": bool(ok),
                     "note": "PCIe-inclusive: never the headline's value; the roofline that bounds it is the link (63 GB/s per direction), not HBM"})
        dec.close()
    except Exception as e:
        legs.append({"config": 5, "failed": repr(e)})
    return legs


def _tiled(n: int, seed: int) -> np.ndarray:
    """n bytes of enwik8-shaped data, generated as 2^24-byte tiles of one base block under per-tile byte permutations (the
    integer generator makes ~7 MB/s per core: a GiB of it would take minutes; the tiles keep its order-0 statistics)."""
    tile = 1 << 24
    base = synth.enwik8_shaped(min(tile, n), seed=seed)
    out = np.empty(n, np.uint8)
    for k, s in enumerate(range(0, n, tile)):
        c = min(tile, n - s)
        out[s:s + c] = _permuted(base, k % 7)[:c]
    return out


class _Dev:
    """What a rank's legs need from its device: the real one (HIP: torch.cuda + hsrans_ctx) or, for --rehearse only, none at
    all (CPU tensors; the decode is the library's host SIMD decoder, the exchange runs over gloo)."""

    def __init__(self, rehearse: bool, dev_index: int):
        self.rehearse = rehearse
        if rehearse:
            self.dev, self.ctx = torch.device("cpu"), None
        else:
            self.dev = torch.device("cuda", dev_index)
            torch.cuda.set_device(self.dev)
            self.ctx = H.Context(dev_index)

    def sync(self):
        if not self.rehearse:
            torch.cuda.synchronize()

    def mark(self):
        if self.rehearse:
            return time.perf_counter()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def ms(self, a, b) -> float:
        return (b - a) * 1e3 if self.rehearse else a.elapsed_time(b)


def _reduce_max(dist, dv: _Dev, x: float) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=dv.dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_workload(args, world, rank, dv: _Dev, dist):
    """ONE stream over all ranks (strong scaling, BASELINE config 4): every rank decodes its chains from its window of the stream;
    the decoded ranges are exchanged INSIDE the timed region, pipelined behind the decode.  Device plans, window uploads and
    buffers are made once, outside it.  Legs (all in this one run, each validated bit-exact before it is timed):
        none   every rank keeps its range (the consumer is sharded too): decode only
        all    every rank ends with the whole output; pipelined over --parts sub-runs, and unpipelined beside it
        root   rank 0 ends with the whole output; its share of the chains is balanced against its inbound links
               (measured by an unweighted, unpipelined step first); pipelined  — this is `value` for N > 1
        one    rank 0 decodes the whole stream alone (the N = 1 figure of the same workload, same run)
    and, beside them, `replicas`: every rank decoding its own 100 MB raw streams (the N = 1 headline workload, weak scaling)."""
    from hypersonic_rans_amd import sharded

    n, S, bits = args.size, args.states, args.bits
    data = _tiled(n, seed=20241008)
    t0 = time.perf_counter()
    container = H.BLOCK if args.container == "block" else H.MT
    cname = "block_" if container == H.BLOCK else "mt_"
    stream, plan = H.encode(container, S, bits, data, block_size=args.block, index_interval=args.interval)
    t_enc = time.perf_counter() - t0
    d_ref = torch.from_numpy(data).to(dv.dev)
    alg = stream.size + n

    def make(parts=1, weights=None, root=None):
        if dv.rehearse:
            return sharded.HostRehearsalDecoder(plan, container, S, bits, parts=parts, weights=weights, root=root)
        return sharded.ShardedDecoder(dv.ctx, plan, parts=parts, weights=weights, root=root)

    def check(dec, out, gathered: bool):
        """bit-exact: the bytes this rank must hold after a step (everything, or its own range)"""
        assert dec.global_status() == 0, "a rank's kernel reported a malformed stream"
        if gathered and (dec.root is None or rank == dec.root):
            ok = torch.equal(out[:n], d_ref)
        else:
            b, e = dec.ranges[rank]
            ok = torch.equal(out[b - dec.out_base:e - dec.out_base], d_ref[b:e])
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dv.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == 0, "sharded decode is not bit-exact"

    def leg(name, parts=1, weights=None, root=None, gather=True, split_legs=False, steps=None):
        """Times `steps` steps of one configuration; max over ranks.  split_legs: decode and exchange timed separately
        (unpipelined, one synchronisation per step) — used for calibration and for the per-leg figures."""
        steps = steps or args.steps
        dec = make(parts, weights, root)
        d_window = dec.upload_window(stream, dv.dev)
        out = dec.alloc_out(dv.dev)
        dv.sync()
        dec.step(d_window, out, gather=gather)
        dv.sync()
        check(dec, out, gather)
        for _ in range(args.warmup):
            dec.step(d_window, out, gather=gather)
        if not dv.rehearse and args.settle_ms > 0:  # (the validation above idled the GPU: see headline())
            t_settle = time.perf_counter()
            while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
                for _ in range(8):
                    dec.step(d_window, out, gather=gather)
                dv.sync()
        dist.barrier()
        dv.sync()
        t0 = time.perf_counter()
        dec_ms = gat_ms = 0.0
        if split_legs:
            for _ in range(steps):
                a = dv.mark()
                dec.step(d_window, out, gather=False)
                b = dv.mark()
                if gather and dv.rehearse:
                    sharded.gather_ranges(out, dec.ranges, None, dec.root, dec.out_base)
                elif gather:
                    dec.exchange(d_window, out)  # hsrans_decode_sharded(HSRANS_SHARD_EXCHANGE_ONLY): the same RCCL groups as the pipelined step's
                c = dv.mark()
                dv.sync()
                dec_ms += dv.ms(a, b) / steps
                gat_ms += dv.ms(b, c) / steps
        else:
            a = dv.mark()
            for _ in range(steps):
                dec.step(d_window, out, gather=gather)
            b = dv.mark()
            dv.sync()
            dec_ms = dv.ms(a, b) / steps
        dist.barrier()
        elapsed = _reduce_max(dist, dv, time.perf_counter() - t0)
        check(dec, out, gather)
        mine = {"rank": rank, "chains": dec.count, "range": list(dec.ranges[rank]), "window_bytes": dec.window[1] - dec.window[0],
                "out_bytes_held": dec.out_len, "stream_ms" if not split_legs else "decode_ms": dec_ms}
        if split_legs:
            mine["gather_ms"] = gat_ms
        if not dv.rehearse and dec.launch_info() is not None:
            mine["launch"] = dec.launch_info()  # kernel geometry of what step() launched: the rank's first sub-run (grid, LDS, class weights, dynamic block order)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        ms = elapsed * 1e3 / steps
        if rank == 0:  # (stderr: if a later leg dies on hardware this run has never seen, the log still holds the legs before it)
            print(f"bench.py sharded: leg {name}: {ms:.4f} ms per step over {world} rank(s)", file=sys.stderr, flush=True)
        return {"leg": name, "ms_per_step": ms, "MiB_s": n / 2**20 / (ms * 1e-3), "parts": parts, "gather": ("none" if not gather else "all" if root is None else f"root={root}"),
                "shares": [r[1] - r[0] for r in dec.ranges], "pipelined": bool(gather and parts > 1), "per_rank": per_rank,
                "sub_runs_in_one_launch": None if dv.rehearse else bool(dec.c.info.get("one_launch", 0))}

    legs = {}
    legs["none"] = leg("none", gather=False, split_legs=True)
    decode_ms = max(r["decode_ms"] for r in legs["none"]["per_rank"])
    root_share = 1.0
    if world > 1:
        legs["all_unpipelined"] = leg("all_unpipelined", gather=True, split_legs=True)
        legs["all"] = leg("all", parts=args.parts, gather=True)
        # calibration of the root's share: an unweighted, unpipelined gather to rank 0 gives one GPU's decode rate D (decoded bytes
        # per second of the slowest rank) and the root's inbound rate B (the other ranks' bytes over the root's exchange time)
        cal = leg("root_unweighted_unpipelined", root=0, gather=True, split_legs=True)
        legs["root_unweighted_unpipelined"] = cal
        r0 = cal["per_rank"][0]
        share = cal["shares"]
        D = max(share) / (max(r["decode_ms"] for r in cal["per_rank"]) * 1e-3)
        B = (n - share[0]) / max(r0["gather_ms"] * 1e-3, 1e-9)
        root_share = args.root_share if args.root_share else sharded.balanced_root_share(world, D, B)
        # every rank must cut the chains identically: rank 0's measurement decides
        t = torch.tensor([root_share], dtype=torch.float64, device=dv.dev)
        dist.broadcast(t, 0)
        root_share = float(t.item())
        legs["root"] = leg("root", parts=args.parts, weights=sharded.root_weights(world, 0, root_share), root=0, gather=True)
        legs["root"].update({"root_share": root_share, "decode_bytes_per_s": D, "root_inbound_bytes_per_s": B})
        legs["one"] = leg("one", weights=sharded.root_weights(world, 0, 1.0), root=0, gather=True)
    else:
        legs["one"] = leg("one", gather=False)  # back-to-back launches, no per-step synchronisation
    main = legs["root"] if world > 1 else legs["one"]

    replicas = None
    if world > 1 and not args.no_replicas and not dv.rehearse:
        import copy

        a2 = copy.copy(args)
        a2.size, a2.no_single, a2.no_cpu, a2.timed_only, a2.repeats, a2.settle_ms = 100_000_000 if args.size >= (1 << 28) else args.size, True, True, True, 3, 10.0
        r = headline(a2, world, rank, dv.dev, dv.dev.index, dv.ctx, dist)
        if r is not None:
            replicas = {"value": r["value"], "unit": "MiB/s", "ms_per_step": r["ms_per_step"], "scaling": "weak", "roofline_frac_per_gpu": r["roofline"]["frac"],
                        "workload": r["config"]["workload"]}

    seen = torch.ones(1, dtype=torch.int32, device=dv.dev)
    dist.all_reduce(seen)
    if rank != 0:
        return None
    cpu = _mt_cpu_baseline(stream, data, S, bits) if (world == 1 and not args.no_cpu and not dv.rehearse and container == H.MT) else None
    one_ms = legs["one"]["ms_per_step"] if world > 1 else main["ms_per_step"]
    result = {
        "metric": "decode MiB/s (bit-exact), one stream sharded over the GPUs, decoded ranges gathered to rank 0" if world > 1 else
                  f"decode MiB/s (bit-exact), one {cname} stream, one GPU",
        "value": main["MiB_s"], "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 state / u16 word / u8 symbol (integer)", "data": "synthetic",
        "config": {"workload": f"{cname}rANS32x{S} 16w {bits}-bit, ONE {n}-byte stream in {args.block}-byte blocks + index every {args.interval} groups, chains sharded "
                               f"over {world} rank(s) by hsrans_plan_slice, decoded ranges gathered to rank 0 point-to-point over "
                               f"{'RCCL' if not dv.rehearse else 'gloo'}, exchange pipelined behind the decode in {args.parts} sub-runs (one launch per rank, a completion word "
                               f"per sub-run), root share {root_share:.3f}.  STRONG scaling of this one stream: compare `value` with `one_gpu_same_stream` (measured in this run) "
                               "or with `sharded_workload_one_gpu` of the N = 1 line (`python bench.py`) — NOT with the N = 1 line's `value`, which is the 100 MB raw headline",
                   "container": cname, "states": S, "bits": bits, "decoded_bytes": n, "compressed_bytes": int(stream.size), "ratio": stream.size / n,
                   "plan_bytes": int(plan.size), "plan_over_compressed": plan.size / stream.size, "chains": H.plan_chain_count(plan), "block": args.block,
                   "index_interval_groups": args.interval, "gather": "root" if world > 1 else "none", "parts": args.parts, "root_share": root_share,
                   "backend": dist.get_backend(), "n_ranks_seen": int(seen.item()), "host_encode_s": t_enc, "bit_exact": True},
        "gather": legs,
        "decode_only_MiB_s": legs["none"]["MiB_s"],
        "one_gpu_same_stream": {"ms_per_step": one_ms, "MiB_s": n / 2**20 / (one_ms * 1e-3)},
        "sub_runs_in_one_launch": main.get("sub_runs_in_one_launch"),
        "speedup_vs_one_gpu": {k: one_ms / v["ms_per_step"] for k, v in legs.items() if k != "one"},
        "replicas": replicas,
        "per_rank": main["per_rank"],
        "cpu_baseline": cpu,
        "roofline": {"bound": "hbm", "achieved": alg / (decode_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": alg / (decode_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None, "algorithmic_bytes_per_launch": alg,
                     "frac_wall": alg / (legs["none"]["ms_per_step"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                     "note": "decode kernels of the `none` leg (slowest rank, HIP events on the launch stream), against the HBM peak of all ranks; "
                             "the exchange is xGMI-bound: gather.*"},
    }
    if dv.rehearse:
        result["rehearsal"] = "host SIMD decoder over gloo: exercises rank spawn, sharding and the pipelined exchange; NOT a measurement of the product"
    return result


def host_workload(args, world, rank, dv: _Dev, dist):
    """BASELINE config 5's shape: every rank decodes its OWN mt_ stream from page-locked host memory into page-locked host memory
    (upload, decode and download of consecutive slices overlapped on three streams: hsrans_hpipe) — independent
    streams, no data-path collective, weak scaling.  PCIe-inclusive by construction, so this is never the headline `value`: it is
    its own workload with its own metric.  The stream is written by the GPU encoder (the scalar host encoder would take minutes)."""
    from hypersonic_rans_amd import pipeline

    n, S, bits = args.size, args.states, args.bits
    ctx = dv.ctx
    g = torch.Generator(device=dv.dev).manual_seed(11 + rank)
    d_in = torch.rand(n, device=dv.dev, generator=g).pow_(6).mul_(205).to(torch.uint8)
    d_enc = torch.empty(H.capacity(H.MT, S, n), dtype=torch.uint8, device=dv.dev)
    m, dplan = ctx.encode_device(H.MT, S, bits, d_in, d_enc, block_size=args.block, index_interval=args.interval, want_plan=True)
    plan = ctx.read_device_plan(dplan, capacity=1 << 30)
    host_stream = torch.empty(m, dtype=torch.uint8).pin_memory()
    host_stream.copy_(d_enc[:m])
    host_ref = d_in.cpu()
    del d_enc, dplan, d_in
    host_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    dec = pipeline.PipelinedHostDecoder(ctx, plan)
    dec.decode(host_stream, host_out)
    assert torch.equal(host_out, host_ref), "pipelined host decode is not bit-exact"
    for _ in range(args.warmup):
        dec.decode(host_stream, host_out)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dec.decode(host_stream, host_out)  # synchronous: the decoded bytes are in host memory on return
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    assert torch.equal(host_out, host_ref), "pipelined host decode is not bit-exact after the timed region"
    if dist is not None:
        elapsed = _reduce_max(dist, dv, elapsed)
    if rank != 0:
        return None
    ms = elapsed * 1e3 / args.steps
    link = 63.0  # PCIe Gen5 x16, GB/s per direction (MI355X_MICROARCH.md): the output leg is the longer one
    return {
        "metric": "decode MiB/s (bit-exact), stream and output in page-locked HOST memory (PCIe-inclusive), one mt_ stream per GPU",
        "value": world * n / 2**20 / (elapsed / args.steps), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 state / u16 word / u8 symbol (integer)", "data": "synthetic",
        "config": {"workload": f"mt_rANS32x{S} 16w {bits}-bit, one {n}-byte stream per GPU in {args.block}-byte blocks + index every {args.interval} groups, "
                               "compressed stream and decoded output in page-locked host memory, hsrans_hpipe (upload, decode and download of consecutive "
                               "slices overlapped)",
                   "container": "mt_", "states": S, "bits": bits, "decoded_bytes": n, "compressed_bytes": int(m), "ratio": m / n, "plan_bytes": int(plan.size),
                   "block": args.block, "index_interval_groups": args.interval, "bit_exact": True},
        "decoded_GB_s": world * n / (elapsed / args.steps) / 1e9,
        "roofline": {"bound": "pcie", "achieved": n / (ms * 1e-3) / 1e9, "peak": link, "unit": "GB/s", "frac": n / (ms * 1e-3) / 1e9 / link, "traffic": None,
                     "note": "per GPU: decoded bytes written to host memory over the link's device-to-host direction (63 GB/s spec); the kernels are PCIe-bound "
                             "here, not HBM-bound: this line never stands in for the headline"},
        "cpu_baseline": None,
    }


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list[str], result_fd: int) -> int:
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts N ranks (one per GPU) through
    `python -m torch.distributed.run` — exactly the command line the driver uses — BEFORE anything here has touched the GPU
    (a process that has initialised HIP must not fork/exec GPU children), relays rank 0's single JSON line and passes a
    failure of any rank on as its own exit code."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    out, _ = p.communicate()
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {n}-rank launch failed (exit code {p.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return p.returncode or 1
    json.loads(lines[0])
    os.write(result_fd, (lines[0] + "\n").encode())
    return 0


def main() -> None:
    # stdout carries exactly ONE line, the JSON result: everything else a library prints there (RCCL's start-up banner, for one)
    # goes to stderr
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=30.0, help="headline: untimed back-to-back launches for this long before the first timed region (0 = none)")
    ap.add_argument("--repeats", type=int, default=15, help="headline: the (W warm-up + K timed steps) region is run this many times; `value` is the median region")
    ap.add_argument("--workload", choices=("headline", "sharded", "host"), default=None,
                    help="default: headline (100 MB raw stream per GPU) on one GPU, sharded (ONE 2^30-byte mt_ stream over all GPUs) on several; "
                         "host: one 2^30-byte mt_ stream per GPU from / to page-locked host memory (BASELINE config 5's shape, PCIe-inclusive)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--bits", type=int, default=11)
    ap.add_argument("--states", type=int, default=64)
    ap.add_argument("--index", default="wave", help="'wave' = one chain per resident wavefront (hsrans_index_boundaries); or G = a checkpoint every G groups")
    ap.add_argument("--pairs", type=int, default=4, help="distinct (stream, output) pairs rotated through the timed loop")
    ap.add_argument("--block", type=int, default=1 << 18, help="sharded: block size in bytes")
    ap.add_argument("--container", choices=("mt", "block"), default="mt", help="sharded: the stream's container (BASELINE config 4 names block_; same kernel, same plan shape)")
    ap.add_argument("--interval", type=int, default=64, help="sharded: checkpoint interval inside the blocks, in groups (64: a rank of 8 holds 4 chains per wavefront — what the "
                    "host-dealt one-round launch wants, tools/shard_projection.py: 57.3 us per rank against 59.9 at 256)")
    ap.add_argument("--parts", type=int, default=4, help="sharded: sub-runs per rank; sub-run k's exchange overlaps sub-run k+1's decode")
    ap.add_argument("--root-share", type=float, default=0.0, help="sharded: the root's share of the decoded bytes (0 = balance it against the measured inbound rate)")
    ap.add_argument("--no-configs", action="store_true", help="headline: skip the 1-GPU legs of BASELINE configs 3, 4 and 5 (result['configs'])")
    ap.add_argument("--configs-interval", type=int, default=64, help="configs 4 / 5: checkpoint interval of the 2^30-byte mt_ stream, in groups")
    ap.add_argument("--no-replicas", action="store_true", help="sharded, N > 1: skip the weak-scaling replicas leg")
    ap.add_argument("--no-calibrate", action="store_true", help="headline: shape the index with the compiled-in class lengths instead of fitting them to this device")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-single", action="store_true", help="skip the un-indexed single-wavefront and GPU-encoder legs")
    ap.add_argument("--one-launch", action="store_true", help="a step = ONE launch that decodes all --pairs streams (hsrans_decode_device_batch), their indexes shaped for it")
    ap.add_argument("--timed-only", action="store_true", help="launch nothing but validation, warm-up and the timed rotation (profiling runs)")
    ap.add_argument("--rehearse", action="store_true",
                    help="NO GPU: ranks over gloo, sub-runs decoded by the library's host decoder — rehearses rank spawn, sharding and the pipelined "
                         "exchange where there is no GPU (tests); the line is marked as not a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: be the launcher (nothing in this process has touched the GPU yet)
        rc = spawn_ranks(args.gpus, sys.argv[1:], result_fd)
        os.close(result_fd)
        sys.exit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload is None:
        args.workload = "headline" if world == 1 else "sharded"
    if args.steps is None:
        args.steps = 50 if args.workload == "headline" else 20 if args.workload == "sharded" else 8
    if args.size is None:
        args.size = 100_000_000 if args.workload == "headline" else 1 << 30
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    distributed = world > 1 or args.workload == "sharded"
    # one process per GPU; if the launcher narrowed each rank's view to its own GPU (HIP_VISIBLE_DEVICES), index 0 is that GPU
    n_visible = torch.cuda.device_count()
    dev_index = local_rank % n_visible if n_visible else 0
    if args.rehearse:
        if args.workload != "sharded":
            raise SystemExit("--rehearse exists for the sharded workload only")
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU decode path")
    dv = _Dev(args.rehearse, dev_index)
    dist = None
    if distributed:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dv.dev)

    if args.workload == "headline":
        result = headline(args, world, rank, dv.dev, dev_index, dv.ctx, dist)
    elif args.workload == "host":
        result = host_workload(args, world, rank, dv, dist)
    else:
        result = sharded_workload(args, world, rank, dv, dist)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
