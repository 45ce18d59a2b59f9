#!/bin/bash
mkdir -p gpurun_out/s6
V=$PWD/hypersonic_rans_amd/lib/variants
O=gpurun_out/s6/linear.jsonl
python tools/rot_probe.py --tag base > $O 2> gpurun_out/s6/err.txt
HSRANS_LIB=$V/libhsrans_hip_linst.so python tools/rot_probe.py --tag linear_stores --no-check >> $O 2>> gpurun_out/s6/err.txt
HSRANS_LIB=$V/libhsrans_hip_nostore.so python tools/rot_probe.py --tag nostore --no-check >> $O 2>> gpurun_out/s6/err.txt
python tools/rot_probe.py --tag base_again >> $O 2>> gpurun_out/s6/err.txt
cut -c1-160 $O
