#!/bin/bash
# where do the slow waves of a rotated launch lose their time?
mkdir -p gpurun_out/s4
V=$PWD/hypersonic_rans_amd/lib/variants
O=gpurun_out/s4/diag.jsonl
HSRANS_DEBUG_STAMPS=1 python tools/rot_probe.py --tag stamps_wait > $O 2> gpurun_out/s4/err.txt
HSRANS_DEBUG_STAMPS=1 HSRANS_LIB=$V/libhsrans_hip_stampst.so python tools/rot_probe.py --tag stamps_wait_store >> $O 2>> gpurun_out/s4/err.txt
HSRANS_LIB=$V/libhsrans_hip_nostore.so python tools/rot_probe.py --tag nostore --no-check >> $O 2>> gpurun_out/s4/err.txt
python tools/rot_probe.py --tag base >> $O 2>> gpurun_out/s4/err.txt
cut -c1-1500 $O
