#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plain_host_decode or first_decode or decode_host" 2>&1 | tail -8
timeout 900 python tools/host_loop_rate.py > gpurun_out/s24_loop.jsonl 2> gpurun_out/s24_loop.err
cat gpurun_out/s24_loop.jsonl; tail -3 gpurun_out/s24_loop.err
