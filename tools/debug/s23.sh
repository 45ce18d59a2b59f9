#!/bin/bash
mkdir -p gpurun_out
HSRANS_DEBUG_STAMPS=1 timeout 900 python tools/encode_rate.py > gpurun_out/s23_enc.jsonl 2> gpurun_out/s23_enc.err
cat gpurun_out/s23_enc.jsonl | cut -c1-700; grep "raw encode" gpurun_out/s23_enc.err | tail -4
