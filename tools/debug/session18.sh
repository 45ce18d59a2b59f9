#!/bin/bash
mkdir -p gpurun_out/s18
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "first_decode or index_build or device_side" > gpurun_out/s18/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s18/pytest.log
tail -25 gpurun_out/s18/pytest.log
