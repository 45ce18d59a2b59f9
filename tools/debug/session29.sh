#!/bin/bash
mkdir -p gpurun_out/s29
rm -f gpurun_out/s29/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_mstrict.so
for i in 1 2; do
python tools/sweep_configs.py --bits 13,14,15 --tag exact >> gpurun_out/s29/sweep_exact.jsonl 2> gpurun_out/s29/err1
HSRANS_LIB=$V python tools/sweep_configs.py --bits 13,14,15 --tag strict >> gpurun_out/s29/sweep_strict.jsonl 2> gpurun_out/s29/err2
python tools/sweep_configs.py --bits 11 --states 32 --tag exact >> gpurun_out/s29/sweep_exact.jsonl 2> gpurun_out/s29/err1
HSRANS_LIB=$V python tools/sweep_configs.py --bits 11 --states 32 --tag strict >> gpurun_out/s29/sweep_strict.jsonl 2> gpurun_out/s29/err2
done
for i in 1 2 3; do
python bench.py --no-cpu --no-single --steps 40 --bits 15 > gpurun_out/s29/b15_exact_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --steps 40 --bits 15 > gpurun_out/s29/b15_strict_$i.json 2>/dev/null
python bench.py --no-cpu --no-single --steps 40 --states 32 > gpurun_out/s29/s32_exact_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --steps 40 --states 32 > gpurun_out/s29/s32_strict_$i.json 2>/dev/null
done
