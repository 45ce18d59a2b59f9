#!/bin/bash
mkdir -p gpurun_out/s8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1; tail -3 gpurun_out/s8/pytest.log
NT=lib/variants/libhsrans_hip_st_nt.so
O=gpurun_out/s8/policy_workloads.jsonl
: > $O
ab() { echo "{\"workload\": \"$1\"}" >> $O; shift; python tools/ab_probe.py --rounds 4 --variant sc0sc1 --variant nt:$NT "$@" >> $O 2>> gpurun_out/s8/err.txt; }
ab raw64_g32 --index 32
ab raw32_wave --states 32
ab raw64_b14 --bits 14
ab raw64_b15 --bits 15
ab raw64_b13 --bits 13
ab mt_100mb_64k --container mt --block 65536 --index 64
ab mt_1gib_256k --container mt --size 1073741824 --pairs 2 --window 30 --index 256
python bench.py --steps 20 --warmup 5 > gpurun_out/s8/bench.json 2> gpurun_out/s8/bench.err
cut -c1-200 $O
