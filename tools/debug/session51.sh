#!/bin/bash
mkdir -p gpurun_out/s51
rm -f gpurun_out/s51/*
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s51/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s51/pytest.log
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "renormalise_on_nearly or random_sweep or rank_table" -p no:cacheprovider >> gpurun_out/s51/soak.log 2>&1; echo "rc=$?" >> gpurun_out/s51/soak.log; done
for i in 1 2 3; do python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s51/bench_$i.json 2>/dev/null; done
python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s51/sharded.json 2>/dev/null
