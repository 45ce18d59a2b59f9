#!/bin/bash
mkdir -p gpurun_out/s33
export HSRANS_DEBUG_STAMPS=1
t() { name=$1; shift; timeout 500 python tools/tune_weights.py "$@" > gpurun_out/s33/$name.txt 2>&1; tail -1 gpurun_out/s33/$name.txt; }
t slot --bits 11 --iters 6 --index 32 --var HSRANS_SLOT_WEIGHTS --start 1328,1268,1211,1145,1018,875,665,490 --cold 4
t direct --bits 11 --iters 5 --var HSRANS_DIRECT_WEIGHTS --start 1403,1348,1262,1147,946,794,628,472 --cold 4
