#!/bin/bash
mkdir -p gpurun_out/s4
for cfg in "w16:HSRANS_GROUP_OVERLAP=0" "w8:HSRANS_GROUP_OVERLAP=0 HSRANS_WAVES_PER_WG=8" "w8ovl:HSRANS_GROUP_OVERLAP=1 HSRANS_WAVES_PER_WG=8" "w8static:HSRANS_GROUP_OVERLAP=0 HSRANS_WAVES_PER_WG=8 HSRANS_GROUP_STATIC=1"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  for rep in 1 2; do
    env $envs python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s4/sharded_${tag}_$rep.json 2> gpurun_out/s4/sharded_${tag}_$rep.err
  done
done
HSRANS_GROUP_OVERLAP=0 HSRANS_WAVES_PER_WG=8 python tools/stamps_grouped.py > gpurun_out/s4/stamps_grouped_w8.txt 2>&1
# 64 KiB blocks (what the reference's encoder emits) with checkpoints, and 1 MiB blocks
for blk in 65536 1048576; do
  for cfg in "w16:HSRANS_GROUP_OVERLAP=0" "w8:HSRANS_GROUP_OVERLAP=0 HSRANS_WAVES_PER_WG=8"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    env $envs python bench.py --workload sharded --no-cpu --steps 20 --block $blk --interval $((blk/64/16)) > gpurun_out/s4/sharded_b${blk}_${tag}.json 2> gpurun_out/s4/sharded_b${blk}_${tag}.err
  done
done
