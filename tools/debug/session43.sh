#!/bin/bash
mkdir -p gpurun_out/s43
rm -f gpurun_out/s43/*
for i in 1 2 3 4 5 6 7 8; do python bench.py --no-cpu --no-single --steps 40 > gpurun_out/s43/b_$i.json 2>/dev/null; done
