#!/bin/bash
mkdir -p gpurun_out/s5
python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s5/pytest.log
for cfg in "tail125:" "tail0:HSRANS_GROUP_TAIL_PERMILLE=0" "tail250:HSRANS_GROUP_TAIL_PERMILLE=250" "tail60:HSRANS_GROUP_TAIL_PERMILLE=60"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  for rep in 1 2; do
    env $envs python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s5/sharded_${tag}_$rep.json 2> gpurun_out/s5/sharded_${tag}_$rep.err
  done
done
python tools/stamps_grouped.py > gpurun_out/s5/stamps_grouped.txt 2>&1
python bench.py --workload sharded --no-cpu --steps 20 --block 65536 --interval 64 > gpurun_out/s5/sharded_b65536.json 2> gpurun_out/s5/sharded_b65536.err
python bench.py --workload sharded --no-cpu --steps 20 --interval 64 > gpurun_out/s5/sharded_i64.json 2> gpurun_out/s5/sharded_i64.err
python bench.py > gpurun_out/s5/bench.json 2> gpurun_out/s5/bench.err
tail -3 gpurun_out/s5/pytest.log
