#!/bin/bash
mkdir -p gpurun_out/s6
python -m pytest tests -m gpu -x -q > gpurun_out/s6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s6/pytest.log
python tools/sweep_configs.py > gpurun_out/s6/config_sweep.jsonl 2> gpurun_out/s6/sweep.err
tail -3 gpurun_out/s6/pytest.log
