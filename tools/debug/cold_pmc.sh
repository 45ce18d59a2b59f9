# Memory-path counters of the headline decode, warm (one pair replayed) against cold (4 pairs rotated).  Run through gpurun; the
# passes are separate rocprofv3 --pmc runs (never combined with tracing).  Summarise with tools/debug/cold_pmc_summary.py.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cold_pmc
mkdir -p $OUT
run() { mode=$1; name=$2; shift 2; pairs=4; [ $mode = warm ] && pairs=1
  # (timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its signal handler)
  timeout -k 5 100 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${mode}_$name -- python3 bench.py --steps 12 --warmup 4 --no-cpu --no-single --timed-only --pairs $pairs > $OUT/${mode}_$name.log 2>&1
  echo "$mode $name rc=$?"; }
for mode in warm cold; do
  run $mode a1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
  run $mode a2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
  run $mode b1 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
  run $mode b2 TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
  run $mode b3 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
  run $mode c1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
  run $mode e1 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
done
find $OUT -name "*.csv" -size +100M -delete
python3 tools/debug/cold_pmc_summary.py
