#!/bin/bash
mkdir -p gpurun_out/s17
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/s17/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s17/pytest.log
timeout 300 python -m pytest tests/test_gpu_calibrate.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s17/pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s17/pytest2.log
tail -3 gpurun_out/s17/pytest.log; tail -2 gpurun_out/s17/pytest2.log
