#!/usr/bin/env python3
"""Mean per launch of every counter tools/debug/cold_pmc.sh collected for the decode kernel, warm next to cold."""
import csv, glob, os, sys, collections
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "cold_pmc")
res = {"warm": {}, "cold": {}}
for mode in res:
    for f in glob.glob(os.path.join(root, mode + "_*", "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_decode" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 3:]  # the first launches are warm-up
            res[mode][k] = sum(v) / len(v)
print(f"{'counter':44s} {'warm':>16s} {'cold':>16s} {'cold/warm':>10s}")
for k in sorted(set(res["warm"]) | set(res["cold"])):
    w, c = res["warm"].get(k, float("nan")), res["cold"].get(k, float("nan"))
    print(f"{k:44s} {w:16.1f} {c:16.1f} {c / w if w else float('nan'):10.2f}")
