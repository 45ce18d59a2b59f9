#!/bin/bash
mkdir -p gpurun_out/s9
NT=lib/variants/libhsrans_hip_st_nt.so
O=gpurun_out/s9/policy_workloads.jsonl
: > $O
ab() { echo "{\"workload\": \"$1\"}" >> $O; shift; python tools/ab_probe.py --rounds 4 --variant sc0sc1 --variant nt:$NT "$@" >> $O 2>> gpurun_out/s9/err.txt; }
ab headline
ab raw64_g32 --index 32
ab raw32_wave --states 32
ab raw64_b14 --bits 14
ab raw64_b12 --bits 12
ab mt_1gib_256k --container mt --size 1073741824 --pairs 2 --window 30 --index 256
ab headline_again
cut -c1-200 $O
