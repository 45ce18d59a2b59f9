#!/bin/bash
mkdir -p gpurun_out/s44
rm -f gpurun_out/s44/*
HSRANS_GROUP_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/s44/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s44/pytest.log
for i in 1 2 3; do
for pf in 0 1; do
HSRANS_GROUP_PREFETCH=$pf python bench.py --workload sharded --no-cpu --steps 10 --block 65536 --interval 64 > gpurun_out/s44/k64_pf${pf}_$i.json 2>/dev/null
HSRANS_GROUP_PREFETCH=$pf python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s44/k256_pf${pf}_$i.json 2>/dev/null
BITS=11 HSRANS_GROUP_PREFETCH=$pf python tools/debug/session34.py 2>/dev/null | head -1 >> gpurun_out/s44/mt100_pf${pf}.jsonl
done
done
