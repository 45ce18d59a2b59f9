#!/bin/bash
# the few-large-blocks case as a bench line (one launch per step) and under rocprofv3
mkdir -p gpurun_out/spread
export TMPDIR=/tmp
python bench.py --workload sharded --no-cpu --size 100000000 --block 262144 --interval 32 --parts 1 > gpurun_out/spread/bench_sharded_100mb_256k_parts1_line.json 2> gpurun_out/spread/err.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/spread/trace -- python3 bench.py --workload sharded --no-cpu --size 100000000 --block 262144 --interval 32 --parts 1 --steps 50 > gpurun_out/spread/trace.log 2>&1
find gpurun_out/spread/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/spread/kernel_stats.csv
head -5 gpurun_out/spread/kernel_stats.csv; tail -c 900 gpurun_out/spread/bench_sharded_100mb_256k_parts1_line.json
