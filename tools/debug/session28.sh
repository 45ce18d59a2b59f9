#!/bin/bash
# strict (constant) against exact crossing wait in the one-chain-per-wave launch, rotated, compiled-in class lengths (no calibration noise)
mkdir -p gpurun_out/s28
rm -f gpurun_out/s28/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_strict.so
for i in 1 2 3 4 5 6 7 8; do
python bench.py --no-cpu --no-single --no-calibrate --steps 60 > gpurun_out/s28/exact_$i.json 2>/dev/null
HSRANS_LIB=$V python bench.py --no-cpu --no-single --no-calibrate --steps 60 > gpurun_out/s28/strict_$i.json 2>/dev/null
done
