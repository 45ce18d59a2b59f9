#!/bin/bash
mkdir -p gpurun_out/s36
rm -f gpurun_out/s36/*
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_searchbuild.so
for i in 1 2 3; do
timeout 300 python tools/debug/session34.py 2>/dev/null | head -1 >> gpurun_out/s36/runs.jsonl
HSRANS_LIB=$V timeout 300 python tools/debug/session34.py 2>/dev/null | head -1 >> gpurun_out/s36/search.jsonl
done
