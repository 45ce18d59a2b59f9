#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python tools/first_decode_rate.py > gpurun_out/s22_first.jsonl 2> gpurun_out/s22_first.err
HSRANS_INDEX_ASSEMBLE_ON_HOST=1 timeout 600 python tools/first_decode_rate.py >> gpurun_out/s22_first.jsonl 2>> gpurun_out/s22_first.err
cat gpurun_out/s22_first.jsonl
