#!/bin/bash
mkdir -p gpurun_out/s11
for rep in 1 2; do
for prio in 0 300 500 700 1000; do
  HSRANS_GROUP_PRIO=$prio python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s11/sharded_prio${prio}_$rep.json 2> gpurun_out/s11/sharded_prio${prio}_$rep.err
done
done
HSRANS_GROUP_PRIO=500 python tools/stamps_grouped.py > gpurun_out/s11/stamps_prio500.txt 2>&1
