#!/bin/bash
mkdir -p gpurun_out
timeout 500 python tools/debug/s20.py 2>&1 | grep -c "equal True"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_capi_device.py -x -q -m gpu 2>&1 | tail -6
HSRANS_INDEXING_TRACE=1 timeout 600 python tools/first_decode_rate.py > gpurun_out/s21_first.jsonl 2> gpurun_out/s21_first.err
cat gpurun_out/s21_first.jsonl | tail -2
