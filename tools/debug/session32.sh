#!/bin/bash
mkdir -p gpurun_out/s32
rm -f gpurun_out/s32/*
python tools/sweep_configs.py --bits 13,14,15 --tag refit > gpurun_out/s32/sweep.jsonl 2> gpurun_out/s32/err
python tools/sweep_configs.py --bits 11,14 --states 32 --tag refit >> gpurun_out/s32/sweep.jsonl 2>> gpurun_out/s32/err
python tools/sweep_configs.py --bits 13,14,15 --tag refit >> gpurun_out/s32/sweep.jsonl 2>> gpurun_out/s32/err
python tools/sweep_configs.py --bits 11,14 --states 32 --tag refit >> gpurun_out/s32/sweep.jsonl 2>> gpurun_out/s32/err
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s32/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s32/pytest.log
