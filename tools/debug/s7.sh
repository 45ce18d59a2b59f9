#!/bin/bash
# in-process A/B: store shapes and policies on the same buffers
mkdir -p gpurun_out/s7
V=lib/variants
python tools/ab_probe.py --rounds 5 --no-check wide,nostore,linst \
  --variant base --variant wide:$V/libhsrans_hip_wide.so --variant nostore:$V/libhsrans_hip_nostore.so --variant linst:$V/libhsrans_hip_linst.so \
  --variant plain:$V/libhsrans_hip_st_plain.so --variant sc1:$V/libhsrans_hip_st_sc1.so --variant sc0sc1:$V/libhsrans_hip_st_sc0sc1.so --variant sc1nt:$V/libhsrans_hip_st_sc1nt.so \
  --variant dyn150:$V/libhsrans_hip_dyn.so:HSRANS_DIRECT_DYN_PERMILLE=150,HSRANS_DIRECT_DYN_GROUPS=32,HSRANS_DIRECT_DYN_MAX=32768 \
  > gpurun_out/s7/ab.jsonl 2> gpurun_out/s7/err.txt
cut -c1-220 gpurun_out/s7/ab.jsonl; grep -v amdgpu.ids gpurun_out/s7/err.txt | tail -5
