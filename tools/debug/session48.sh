#!/bin/bash
mkdir -p gpurun_out/s48
rm -f gpurun_out/s48/*
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_harness.py -m gpu -x -q > gpurun_out/s48/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s48/pytest.log
python tools/sweep_configs.py --bits 14,15 --states 32 --tag pairrank > gpurun_out/s48/sweep.jsonl 2>/dev/null
python tools/sweep_configs.py --bits 14,15 --states 32 --tag pairrank >> gpurun_out/s48/sweep.jsonl 2>/dev/null
