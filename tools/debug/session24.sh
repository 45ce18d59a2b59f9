#!/bin/bash
mkdir -p gpurun_out/s24
timeout 300 python tools/sweep_configs.py --bits 13,14,15 --states 32 --tag rank32 > gpurun_out/s24/sweep.jsonl 2> gpurun_out/s24/err.txt
HSRANS_NO_RANK_TABLE=1 timeout 300 python tools/sweep_configs.py --bits 14,15 --states 32 --tag norank > gpurun_out/s24/sweep_norank.jsonl 2> gpurun_out/s24/err2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s24/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s24/pytest.log
tail -3 gpurun_out/s24/pytest.log
