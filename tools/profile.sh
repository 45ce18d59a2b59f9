#!/bin/bash
# Profiling recipe for the decode kernel (run on the GPU box through gpurun, from the repo root):
#   bash tools/profile.sh <tag> [bench.py args...]
# 1. kernel trace + stats (rocprofv3 --kernel-trace --stats) of the default bench command;
# 2. PMC counters in SEPARATE passes (never combined with tracing; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# Results land in gpurun_out/prof_<tag>/; tools/pmc_summary.py turns them into profiles/<tag>_*.{csv,json}.
set -u
TAG=${1:-r04}; shift || true
export TMPDIR=/tmp
BARGS="$*"
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
# STEPS=<n> overrides the step count; ONLY_TRACE=1 skips the counter passes (multi-GiB runs: the host encode dominates every pass)
# (every rocprofv3 run sits under a timeout: a counter set the hardware cannot collect makes it abort and then hang in its signal handler)
timeout -k 5 ${TRACE_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu --no-single --timed-only $BARGS > $OUT/trace.log 2>&1
if [ "${ONLY_TRACE:-0}" = "1" ]; then tail -2 $OUT/trace.log; exit 0; fi
# (counter passes serialise the launches: no settle phase, two repeats of the region — the counters do not depend on the GPU's clock state)
pmc() { name=$1; shift; timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 12 --warmup 4 --repeats 2 --settle-ms 0 --no-cpu --no-single --no-calibrate --timed-only $BARGS > $OUT/pmc_$name.log 2>&1; }
pmc a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
pmc c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVE_CYCLES
pmc d FETCH_SIZE
pmc e WRITE_SIZE
pmc f GRBM_GUI_ACTIVE GRBM_COUNT
pmc g TCC_HIT_sum TCC_MISS_sum
tail -2 $OUT/trace.log
