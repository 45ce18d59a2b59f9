# Does the shipped (stamp-free) headline kernel want more or less skew in the per-class chain lengths than the fitted table?
# w' = 1000 + a (w - 1000) for a few a; replayed 100 MB decode, 2 runs each.  Run through gpurun.
base="1424 1371 1283 1165 920 768 606 464"
for a in 80 90 100 110 120; do
  w=$(python -c "print(','.join(str(int(round(1000+$a/100*(x-1000)))) for x in map(int,'$base'.split())))")
  for rep in 1 2; do
    HSRANS_DIRECT_WEIGHTS=$w bash tools/debug/bits_quick.sh 11 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('skew $a% [$w]', r['kernel_us'], r['frac'])"
  done
done
