#!/bin/bash
mkdir -p gpurun_out/s27
rm -f gpurun_out/s27/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_strict.so
python tools/sweep_configs.py --bits 11,12 --tag exact > gpurun_out/s27/sweep_exact.jsonl 2> gpurun_out/s27/err1
HSRANS_LIB=$V python tools/sweep_configs.py --bits 11,12 --tag strict > gpurun_out/s27/sweep_strict.jsonl 2> gpurun_out/s27/err2
python tools/sweep_configs.py --bits 11,12 --tag exact >> gpurun_out/s27/sweep_exact.jsonl 2> gpurun_out/s27/err1
HSRANS_LIB=$V python tools/sweep_configs.py --bits 11,12 --tag strict >> gpurun_out/s27/sweep_strict.jsonl 2> gpurun_out/s27/err2
