#!/bin/bash
# LDS counters of dual-kernel variants through the sweep tool (works for variants whose output is wrong on purpose)
mkdir -p gpurun_out/s21
export TMPDIR=/tmp
for v in "$@"; do
export HSRANS_LIB=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_$v.so
rm -rf gpurun_out/s21/pmc_$v
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/s21/pmc_$v -- python3 tools/sweep_configs.py --bits 15 --tag $v > gpurun_out/s21/sweep_$v.jsonl 2> gpurun_out/s21/err_$v.txt
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/s21/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode_dual" in r["Kernel_Name"] and r["Grid_Size"] == "262144":
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
g = 100_000_000 / 64
print("$v", {k: round(sum(v) / len(v) / g, 3) for k, v in acc.items()}, [ (json.loads(l)["interval"], json.loads(l)["ms"]) for l in open("gpurun_out/s21/sweep_$v.jsonl")])
PY
done
