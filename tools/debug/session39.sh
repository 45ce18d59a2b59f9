#!/bin/bash
mkdir -p gpurun_out/s39
rm -f gpurun_out/s39/*
python tools/sweep_configs.py --bits 13 --states 64,32 --tag default > gpurun_out/s39/sweep.jsonl 2>/dev/null
HSRANS_PACK64_MAX_BITS=12 python tools/sweep_configs.py --bits 13 --states 64,32 --tag rank13 >> gpurun_out/s39/sweep.jsonl 2>/dev/null
python tools/sweep_configs.py --bits 13 --states 64,32 --tag default >> gpurun_out/s39/sweep.jsonl 2>/dev/null
HSRANS_PACK64_MAX_BITS=12 python tools/sweep_configs.py --bits 13 --states 64,32 --tag rank13 >> gpurun_out/s39/sweep.jsonl 2>/dev/null
