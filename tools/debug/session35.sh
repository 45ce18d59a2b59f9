#!/bin/bash
mkdir -p gpurun_out/s35
rm -f gpurun_out/s35/*
timeout 600 python tools/debug/session34.py > gpurun_out/s35/grouped_wide.jsonl 2>/dev/null
timeout 300 python tools/sweep_configs.py --bits 14,15 --tag rankasm > gpurun_out/s35/sweep.jsonl 2>/dev/null
HSRANS_DUAL=0 timeout 300 python tools/sweep_configs.py --bits 14,15 --tag rankasm_dual0 >> gpurun_out/s35/sweep.jsonl 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_harness.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/s35/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s35/pytest.log
tail -3 gpurun_out/s35/pytest.log
