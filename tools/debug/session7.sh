#!/bin/bash
mkdir -p gpurun_out/s7
python -m pytest tests/test_gpu_parity.py tests/test_harness.py -m gpu -x -q > gpurun_out/s7/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s7/pytest.log
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 1 --decode-runs 8 --test > gpurun_out/s7/harness_100mb_11bit.txt 2>&1
python tools/host_pipeline_rate.py > gpurun_out/s7/host_pipeline_1gib.jsonl 2> gpurun_out/s7/pipeline.err
# the launcher forms: one rank under torchrun; two ranks on ONE GPU (expected to be refused by RCCL: the launcher must pass the failure on)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --workload sharded --no-cpu --steps 5 --size 268435456 > gpurun_out/s7/torchrun_n1.json 2> gpurun_out/s7/torchrun_n1.err; echo "rc=$?" >> gpurun_out/s7/torchrun_n1.err
timeout 600 python bench.py --gpus 2 --no-cpu --steps 3 --size 134217728 --no-replicas > gpurun_out/s7/two_ranks_one_gpu.json 2> gpurun_out/s7/two_ranks_one_gpu.err; echo "rc=$?" >> gpurun_out/s7/two_ranks_one_gpu.err
tail -3 gpurun_out/s7/pytest.log
