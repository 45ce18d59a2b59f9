#!/bin/bash
# re-fit of the 32-state pair class lengths (diagnostic build: make -C hypersonic_rans_amd/csrc stamps)
mkdir -p gpurun_out/s31
export HSRANS_DEBUG_STAMPS=1
timeout 500 python tools/tune_weights.py --bits 11 --states 32 --iters 6 --var HSRANS_DIRECT_WEIGHTS_PAIR --start 1737,1597,1391,1144,856,609,405,260 --cold 4 > gpurun_out/s31/p11.txt 2>&1
tail -2 gpurun_out/s31/p11.txt
