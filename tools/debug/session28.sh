#!/bin/bash
mkdir -p gpurun_out/s28
rm -f gpurun_out/s28/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_strict.so
for i in 1 2 3 4 5 6; do
python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s28/exact_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s28/strict_$i.json 2>/dev/null
done
