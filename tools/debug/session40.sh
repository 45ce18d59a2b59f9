#!/bin/bash
mkdir -p gpurun_out/s40
rm -f gpurun_out/s40/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_pairexec.so
for i in 1 2; do
python tools/sweep_configs.py --bits 11 --states 32 --tag select >> gpurun_out/s40/sweep.jsonl 2>/dev/null
HSRANS_LIB=$V python tools/sweep_configs.py --bits 11 --states 32 --tag exec >> gpurun_out/s40/sweep.jsonl 2>/dev/null
done
for i in 1 2 3; do
python bench.py --no-cpu --no-single --no-calibrate --steps 40 --states 32 > gpurun_out/s40/select_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --no-calibrate --steps 40 --states 32 > gpurun_out/s40/exec_$i.json 2>/dev/null
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s40/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s40/pytest.log
