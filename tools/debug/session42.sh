#!/bin/bash
mkdir -p gpurun_out/s42
rm -f gpurun_out/s42/*
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s42/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s42/pytest.log
for i in 1 2 3; do python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s42/bench_$i.json 2>/dev/null; done
python bench.py --no-cpu --no-single --steps 40 --pairs 1 > gpurun_out/s42/warm.json 2>/dev/null
for i in 1 2; do python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s42/sharded_$i.json 2>/dev/null; done
python bench.py --workload sharded --no-cpu --steps 10 --block 65536 --interval 64 > gpurun_out/s42/sharded64k.json 2>/dev/null
python tools/sweep_configs.py --bits 11 --states 64,32 > gpurun_out/s42/sweep.jsonl 2>/dev/null
