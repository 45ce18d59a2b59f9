#!/bin/bash
mkdir -p gpurun_out/s17
V=lib/variants
O=gpurun_out/s17/ab.jsonl
: > $O
ab() { echo "{\"workload\": \"$1\"}" >> $O; shift; python tools/ab_probe.py --rounds 4 --variant new --variant prev:$V/libhsrans_hip_prev.so "$@" >> $O 2>> gpurun_out/s17/err.txt; }
ab headline
ab raw64_b14 --bits 14
ab raw64_b15 --bits 15
ab raw64_b13 --bits 13
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s17/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/s17/pytest.log | tail -2
cut -c1-200 $O
