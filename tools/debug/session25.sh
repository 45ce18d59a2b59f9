#!/bin/bash
mkdir -p gpurun_out/s25
rm -f gpurun_out/s25/*.json
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_exactwait.so
C=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_constwait.so
for i in 1 2 3; do
  python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s25/new_$i.json 2>/dev/null
  HSRANS_LIB=$V python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s25/old_$i.json 2>/dev/null
done
python bench.py --no-cpu --no-single --steps 40 --pairs 1 > gpurun_out/s25/new_warm.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --steps 40 --pairs 1 > gpurun_out/s25/old_warm.json 2>/dev/null
HSRANS_LIB=$C python bench.py --no-cpu --no-single --steps 40 --pairs 1 > gpurun_out/s25/const_warm.json 2>/dev/null
python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s25/new_sharded.json 2>/dev/null
HSRANS_LIB=$V python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s25/old_sharded.json 2>/dev/null
HSRANS_LIB=$C python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s25/const_sharded.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s25/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s25/pytest.log
tail -2 gpurun_out/s25/pytest.log
