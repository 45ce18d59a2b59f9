#!/bin/bash
mkdir -p gpurun_out/s16
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/s16/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s16/pytest.log
python tools/sweep_configs.py > gpurun_out/s16/sweep_fine.jsonl 2> gpurun_out/s16/sweep.err
HSRANS_GROUP_FINE_SPLIT=0 python tools/sweep_configs.py > gpurun_out/s16/sweep_coarse.jsonl 2>> gpurun_out/s16/sweep.err
python tools/encode_rate.py > gpurun_out/s16/encode_rate.jsonl 2> gpurun_out/s16/encode.err
tail -3 gpurun_out/s16/pytest.log
