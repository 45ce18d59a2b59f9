# Re-fits every per-class chain-length table (tools/tune_weights.py) on the GPU; prints the last iteration of each.  Run through gpurun.
O=gpurun_out/tune; mkdir -p $O
t() { name=$1; shift; timeout 400 python tools/tune_weights.py "$@" > $O/$name.txt 2>&1; echo "== $name: $*"; tail -3 $O/$name.txt; }
t b11 --bits 11 --iters 6
t b11cold --bits 11 --iters 5 --cold 4
t b13dual --bits 13 --iters 5 --var HSRANS_DUAL_WEIGHTS --start 1105,1052,977,867,1104,1051,976,867
HSRANS_DUAL=0 t b13single --bits 13 --iters 5 --var HSRANS_DIRECT_WEIGHTS4 --start 1067,1038,983,911,1068,1038,985,911
t b14dual --bits 14 --iters 5 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1105,1052,977,867,1104,1051,976,867
HSRANS_DUAL=0 t b14single --bits 14 --iters 5 --var HSRANS_DIRECT_WEIGHTS3 --start 1040,1018,989,952,1039,1018,989,953
t b15dual --bits 15 --iters 5 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1105,1052,977,867,1104,1051,976,867
HSRANS_DUAL=0 t b15single --bits 15 --iters 5 --var HSRANS_DIRECT_WEIGHTS6 --start 1124,1102,1076,1047,984,941,891,834
t s32 --bits 11 --states 32 --iters 5
