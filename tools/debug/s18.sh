#!/bin/bash
mkdir -p gpurun_out/s18
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s18/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/s18/pytest.log | tail -2
python tools/ab_probe.py --rounds 4 --states 32 --variant new --variant prev:lib/variants/libhsrans_hip_prev.so > gpurun_out/s18/ab_s32.jsonl 2> gpurun_out/s18/err.txt; cut -c1-160 gpurun_out/s18/ab_s32.jsonl
bash tools/regen_round.sh r04 > gpurun_out/s18/regen.log 2>&1; tail -5 gpurun_out/s18/regen.log
