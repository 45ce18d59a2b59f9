#!/bin/bash
# usage: session20.sh <variant> ; sweep at 14/15 bits + LDS counters at 15 bits
mkdir -p gpurun_out/s20
export TMPDIR=/tmp
v=$1
export HSRANS_LIB=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_$v.so
timeout 300 python tools/sweep_configs.py --bits 13,14,15 --tag $v > gpurun_out/s20/sweep_$v.jsonl 2> gpurun_out/s20/err_$v.txt
rm -rf gpurun_out/s20/pmc_$v
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/s20/pmc_$v -- python3 bench.py --steps 12 --warmup 4 --no-cpu --no-single --timed-only --bits 15 --no-calibrate > gpurun_out/s20/pmc_$v.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/s20/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"] and "calibrate" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
g = 100_000_000 / 64
print({k: round(sum(v) / len(v) / g, 3) for k, v in acc.items()})
PY
