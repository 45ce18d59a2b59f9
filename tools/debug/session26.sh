#!/bin/bash
mkdir -p gpurun_out/s26
rm -f gpurun_out/s26/*
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/s26/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s26/pytest.log
tail -2 gpurun_out/s26/pytest.log
python tools/sweep_configs.py > gpurun_out/s26/sweep.jsonl 2> gpurun_out/s26/sweep.err
for i in 1 2; do python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s26/bench_$i.json 2>/dev/null; done
python bench.py --no-cpu --no-single --steps 40 --pairs 1 > gpurun_out/s26/bench_warm.json 2>/dev/null
python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s26/bench_sharded.json 2>/dev/null
