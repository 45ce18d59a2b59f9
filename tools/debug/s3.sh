#!/bin/bash
# existing (unprefetched) dynamic tail, bigger pools than round 3 tried
mkdir -p gpurun_out/s3
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_dyn.so
O=gpurun_out/s3/dyn.jsonl
python tools/rot_probe.py --tag base > $O 2> gpurun_out/s3/err.txt
for pm in 100 150 200 300; do for g in 32 64; do
  HSRANS_LIB=$V HSRANS_DIRECT_DYN_PERMILLE=$pm HSRANS_DIRECT_DYN_GROUPS=$g HSRANS_DIRECT_DYN_MAX=32768 python tools/rot_probe.py --tag dyn_${pm}_${g} >> $O 2>> gpurun_out/s3/err.txt
done; done
HSRANS_LIB=$V HSRANS_DIRECT_DYN_PERMILLE=150 HSRANS_DIRECT_DYN_GROUPS=16 HSRANS_DIRECT_DYN_MAX=32768 python tools/rot_probe.py --tag dyn_150_16 >> $O 2>> gpurun_out/s3/err.txt
python tools/rot_probe.py --tag base_again >> $O 2>> gpurun_out/s3/err.txt
cut -c1-200 $O
