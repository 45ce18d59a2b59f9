#!/bin/bash
# cache-policy bits of the output stores, sustained rotated / replayed
mkdir -p gpurun_out/s5
V=$PWD/hypersonic_rans_amd/lib/variants
O=gpurun_out/s5/store_policy.jsonl
python tools/rot_probe.py --tag base_nt > $O 2> gpurun_out/s5/err.txt
for v in st_plain st_sc1 st_sc0sc1 st_sc1nt st_sc0 st_sc0nt st_all; do
  HSRANS_LIB=$V/libhsrans_hip_$v.so python tools/rot_probe.py --tag $v >> $O 2>> gpurun_out/s5/err.txt
done
python tools/rot_probe.py --tag base_nt_again >> $O 2>> gpurun_out/s5/err.txt
cut -c1-160 $O
