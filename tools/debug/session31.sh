#!/bin/bash
# re-fit of the class lengths after the loops' bookkeeping change (diagnostic build: make -C hypersonic_rans_amd/csrc stamps)
mkdir -p gpurun_out/s31
export HSRANS_DEBUG_STAMPS=1
t() { name=$1; shift; timeout 500 python tools/tune_weights.py "$@" > gpurun_out/s31/$name.txt 2>&1; tail -1 gpurun_out/s31/$name.txt; }
t d15 --bits 15 --iters 6 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1170,1083,953,795,1170,1083,953,795 --cold 4
t d14 --bits 14 --iters 5 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1170,1083,953,795,1170,1083,953,795 --cold 4
t d13 --bits 13 --iters 6 --var HSRANS_DUAL_WEIGHTS --start 1249,1118,925,708,1249,1118,925,708 --cold 4
t p11 --bits 11 --states 32 --iters 6 --var HSRANS_DIRECT_WEIGHTS_PAIR --start 1847,1695,1471,1174,780,512,317,204 --cold 4
