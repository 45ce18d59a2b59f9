#!/bin/bash
mkdir -p gpurun_out/s15
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s15/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/s15/pytest.log | tail -2
python tools/ab_probe.py --rounds 4 --variant new --variant old_nt:lib/variants/libhsrans_hip_st_nt.so > gpurun_out/s15/ab.jsonl 2> gpurun_out/s15/err.txt
cut -c1-200 gpurun_out/s15/ab.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 tests/rccl_worker.py > gpurun_out/s15/rccl_worker_n1.txt 2>&1; tail -2 gpurun_out/s15/rccl_worker_n1.txt | cut -c1-300
python bench.py --steps 20 --warmup 5 > gpurun_out/s15/bench.json 2> gpurun_out/s15/bench.err; tail -c 600 gpurun_out/s15/bench.json
