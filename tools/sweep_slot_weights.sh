#!/bin/bash
# Tuning aid: bench.py under different HSRANS_SLOT_WEIGHTS (per-mille static run length of the 8 wave classes)
# and HSRANS_STATIC_PERCENT; arguments are "weights[:percent]".
for a in "$@"; do
  w=${a%%:*}; p=100; [[ "$a" == *:* ]] && p=${a##*:}
  printf "%s static %s%%  " "$w" "$p"
  HSRANS_SLOT_WEIGHTS=$w HSRANS_STATIC_PERCENT=$p python bench.py --steps 100 --warmup 10 --no-cpu --no-single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
