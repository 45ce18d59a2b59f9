#!/bin/bash
mkdir -p gpurun_out/s10 gpurun_out/r03
python -m pytest tests/test_gpu_parity.py tests/test_gpu_calibrate.py -m gpu -x -q > gpurun_out/s10/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s10/pytest.log
bash tools/profile.sh r03
bash tools/profile.sh r03_warm --pairs 1
ONLY_TRACE=1 STEPS=20 bash tools/profile.sh r03_1gib --size 1073741824 --pairs 1
python bench.py > gpurun_out/r03/bench_line.json 2> gpurun_out/r03/bench.err
python bench.py --workload sharded > gpurun_out/r03/bench_sharded_line.json 2> gpurun_out/r03/bench_sharded.err
python tools/stamps.py --index wave 2>/dev/null | grep -v amdgpu > gpurun_out/r03/stamps_wave_warm.txt
python tools/stamps.py --index wave --cold 4 2>/dev/null | grep -v amdgpu > gpurun_out/r03/stamps_wave_cold.txt
tail -3 gpurun_out/s10/pytest.log
