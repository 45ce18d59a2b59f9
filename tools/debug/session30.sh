#!/bin/bash
mkdir -p gpurun_out/s30
rm -f gpurun_out/s30/*
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rank_table" > gpurun_out/s30/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s30/pytest.log
for i in 1 2; do
python bench.py --workload sharded --no-cpu --steps 10 --block 65536 --interval 64 > gpurun_out/s30/k64_base_$i.json 2>/dev/null
HSRANS_GROUP_OVERLAP=1 python bench.py --workload sharded --no-cpu --steps 10 --block 65536 --interval 64 > gpurun_out/s30/k64_overlap_$i.json 2>/dev/null
python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s30/k256_base_$i.json 2>/dev/null
HSRANS_GROUP_OVERLAP=1 python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s30/k256_overlap_$i.json 2>/dev/null
done
