#!/bin/bash
mkdir -p gpurun_out/s15 gpurun_out/r03
python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/s15/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s15/pytest.log
python bench.py --workload host > gpurun_out/r03/bench_host_line.json 2> gpurun_out/r03/bench_host.err
tail -3 gpurun_out/s15/pytest.log
