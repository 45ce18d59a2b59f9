#!/bin/bash
# round 4 session 1: baseline on this box + pair probe + cold stamps
mkdir -p gpurun_out/s1
python bench.py --steps 20 --warmup 5 > gpurun_out/s1/bench_a.json 2> gpurun_out/s1/bench_a.err
python tools/pair_probe.py > gpurun_out/s1/pair_probe.txt 2> gpurun_out/s1/pair_probe.err
python tools/stamps.py --index wave --cold 4 2>/dev/null | grep -v amdgpu > gpurun_out/s1/stamps_cold.txt
python bench.py --steps 20 --warmup 5 --no-cpu --no-single > gpurun_out/s1/bench_b.json 2> gpurun_out/s1/bench_b.err
tail -c 300 gpurun_out/s1/bench_a.json; cat gpurun_out/s1/pair_probe.txt | tail -20
