#!/bin/bash
mkdir -p gpurun_out/s16
V=lib/variants
python tools/ab_probe.py --rounds 5 --variant base --variant prosplit:$V/libhsrans_hip_prosplit.so --variant ahead2:$V/libhsrans_hip_ahead2.so > gpurun_out/s16/ab.jsonl 2> gpurun_out/s16/err.txt
cut -c1-200 gpurun_out/s16/ab.jsonl
HSRANS_DEBUG_STAMPS=1 python tools/rot_probe.py --tag base_stamps > gpurun_out/s16/stamps.jsonl 2>> gpurun_out/s16/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu --no-single > gpurun_out/s16/bench.json 2> gpurun_out/s16/bench.err
