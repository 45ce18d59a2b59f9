#!/bin/bash
mkdir -p gpurun_out/s50
rm -f gpurun_out/s50/*
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s50/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s50/pytest.log
python tools/sweep_configs.py --bits 14,15 --states 64,32 --tag persistrank > gpurun_out/s50/sweep.jsonl 2>/dev/null
python tools/sweep_configs.py --bits 14,15 --states 64,32 --tag persistrank >> gpurun_out/s50/sweep.jsonl 2>/dev/null
