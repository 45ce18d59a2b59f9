#!/bin/bash
# round 3, GPU session 1: tests, baseline bench lines, sharded N=1 through nccl, zero-copy probe
mkdir -p gpurun_out/s1
python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
python bench.py --workload sharded --no-cpu > gpurun_out/s1/sharded.json 2> gpurun_out/s1/sharded.err
python tools/zero_copy_probe.py > gpurun_out/s1/zero_copy.jsonl 2> gpurun_out/s1/zero_copy.err
python tools/stamps_grouped.py > gpurun_out/s1/stamps_grouped.txt 2>&1
tail -3 gpurun_out/s1/pytest.log
