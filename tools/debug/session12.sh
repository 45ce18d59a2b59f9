#!/bin/bash
mkdir -p gpurun_out/s12
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s12/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s12/pytest.log
for rep in 1 2; do
  python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s12/sharded_$rep.json 2> gpurun_out/s12/sharded_$rep.err
  HSRANS_GROUP_PRIO=0 python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s12/sharded_prio0_$rep.json 2> gpurun_out/s12/sharded_prio0_$rep.err
  python bench.py --workload sharded --no-cpu --steps 20 --block 65536 --interval 64 > gpurun_out/s12/sharded64k_$rep.json 2> gpurun_out/s12/sharded64k_$rep.err
  HSRANS_GROUP_PRIO=0 python bench.py --workload sharded --no-cpu --steps 20 --block 65536 --interval 64 > gpurun_out/s12/sharded64k_prio0_$rep.json 2> gpurun_out/s12/sharded64k_prio0_$rep.err
done
python tools/sweep_configs.py > gpurun_out/s12/config_sweep.jsonl 2> gpurun_out/s12/sweep.err
STEPS=10 bash tools/profile.sh r03_sharded --workload sharded
tail -3 gpurun_out/s12/pytest.log
