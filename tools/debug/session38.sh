#!/bin/bash
# "nt" against "nt sc1" output stores: rotated headline (compiled-in class lengths), replayed, grouped 2^30-byte launch
mkdir -p gpurun_out/s38
rm -f gpurun_out/s38/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_p_ntsc1.so
for i in 1 2 3 4 5 6; do
python bench.py --no-cpu --no-single --no-calibrate --steps 60 > gpurun_out/s38/nt_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --no-calibrate --steps 60 > gpurun_out/s38/ntsc1_$i.json 2>/dev/null
done
for i in 1 2; do
python bench.py --no-cpu --no-single --no-calibrate --steps 60 --pairs 1 > gpurun_out/s38/warm-nt_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --no-calibrate --steps 60 --pairs 1 > gpurun_out/s38/warm-ntsc1_$i.json 2>/dev/null
python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s38/sharded-nt_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --workload sharded --no-cpu --steps 10 > gpurun_out/s38/sharded-ntsc1_$i.json 2>/dev/null
done
