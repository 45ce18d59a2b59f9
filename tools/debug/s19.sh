#!/bin/bash
# device-side index assembly: parity of the plan it leaves, then the first-decode timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_decode or index_build" 2>&1 | tail -15 > gpurun_out/s19_tests.log
HSRANS_INDEXING_TRACE=1 timeout 600 python tools/first_decode_rate.py > gpurun_out/s19_first.jsonl 2> gpurun_out/s19_first.err
HSRANS_INDEX_ASSEMBLE_ON_HOST=1 HSRANS_INDEXING_TRACE=1 timeout 600 python tools/first_decode_rate.py > gpurun_out/s19_first_host.jsonl 2> gpurun_out/s19_first_host.err
tail -5 gpurun_out/s19_tests.log; cat gpurun_out/s19_first.jsonl; tail -8 gpurun_out/s19_first.err; cat gpurun_out/s19_first_host.jsonl | tail -2
