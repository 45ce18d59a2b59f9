#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s3/pytest.log
for cfg in "dyn_ovl:" "static_ovl:HSRANS_GROUP_STATIC=1" "dyn_noovl:HSRANS_GROUP_OVERLAP=0" "static_noovl:HSRANS_GROUP_STATIC=1 HSRANS_GROUP_OVERLAP=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  for rep in 1 2; do
    env $envs python bench.py --workload sharded --no-cpu --steps 30 > gpurun_out/s3/sharded_${tag}_$rep.json 2> gpurun_out/s3/sharded_${tag}_$rep.err
  done
done
python tools/stamps_grouped.py > gpurun_out/s3/stamps_grouped_dyn.txt 2>&1
HSRANS_GROUP_STATIC=1 HSRANS_GROUP_OVERLAP=0 python tools/stamps_grouped.py > gpurun_out/s3/stamps_grouped_static.txt 2>&1
python tools/host_decoder_vs_reference.py --size 100000000 --budget 2.0 --cases 32:11,32:12,32:13,32:14,32:15,64:11,64:12,64:13,64:14,64:15 > gpurun_out/s3/host_decoder.jsonl 2> gpurun_out/s3/host_decoder.err
tail -3 gpurun_out/s3/pytest.log
