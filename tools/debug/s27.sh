#!/bin/bash
mkdir -p gpurun_out
export HSRANS_LIB=$PWD/hypersonic_rans_amd/lib/libhsrans_hip_stamps.so
for cfg in "" "HSRANS_GROUP_PART_CHAINS=64 HSRANS_GROUP_OVERLAP=1" "HSRANS_GROUP_PART_CHAINS=64 HSRANS_GROUP_OVERLAP=1 HSRANS_WAVES_PER_WG=8"; do
  echo "== $cfg"
  env $cfg timeout 600 python tools/stamps_grouped.py --size 100000000 --block 262144 --interval 32 2>&1 | grep -v amdgpu.ids
done > gpurun_out/s27_stamps.txt
cat gpurun_out/s27_stamps.txt
