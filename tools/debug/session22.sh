#!/bin/bash
mkdir -p gpurun_out/s22
timeout 300 python tools/sweep_configs.py --bits 13,14,15 --tag rank > gpurun_out/s22/sweep.jsonl 2> gpurun_out/s22/err.txt
HSRANS_NO_RANK_TABLE=1 timeout 300 python tools/sweep_configs.py --bits 14,15 --tag norank > gpurun_out/s22/sweep_norank.jsonl 2> gpurun_out/s22/err2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/s22/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s22/pytest.log
tail -3 gpurun_out/s22/pytest.log
