#!/bin/bash
mkdir -p gpurun_out/s13 gpurun_out/r03
python -m pytest tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/s13/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s13/pytest.log
python bench.py --workload sharded > gpurun_out/r03/bench_sharded_line.json 2> gpurun_out/r03/bench_sharded.err
python bench.py --workload sharded --no-cpu --block 65536 --interval 64 > gpurun_out/r03/bench_sharded_64k_line.json 2> gpurun_out/r03/bench_sharded_64k.err
python tools/sweep_configs.py > gpurun_out/r03/config_sweep.jsonl 2> gpurun_out/r03/sweep.err
HSRANS_TABLE_SPILL=1 python tools/sweep_configs.py --only-raw --tag "HSRANS_TABLE_SPILL=1 (tables left in global memory)" >> gpurun_out/r03/config_sweep.jsonl 2>> gpurun_out/r03/sweep.err
HSRANS_DUAL=0 python tools/sweep_configs.py --only-raw --tag "HSRANS_DUAL=0 (one chain per wave at 13-15 bits)" >> gpurun_out/r03/config_sweep.jsonl 2>> gpurun_out/r03/sweep.err
python tools/stamps_grouped.py 2>/dev/null | grep -v amdgpu > gpurun_out/r03/stamps_grouped_1gib.txt
tail -3 gpurun_out/s13/pytest.log
