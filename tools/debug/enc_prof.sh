#!/bin/bash
# rocprofv3 kernel trace of the GPU encoders (tools/encode_rate.py)
export TMPDIR=/tmp
mkdir -p gpurun_out/encprof
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/encprof/trace -- python3 tools/encode_rate.py --cpu-sample 0 > gpurun_out/encprof/log.txt 2>&1
find gpurun_out/encprof/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/encprof/kernel_stats.csv
cut -c1-200 gpurun_out/encprof/kernel_stats.csv | head -12
