O=gpurun_out/tune; mkdir -p $O
t() { name=$1; shift; timeout 400 python tools/tune_weights.py "$@" > $O/$name.txt 2>&1; echo "== $name: $*"; tail -2 $O/$name.txt; }
t d13 --bits 13 --iters 6 --var HSRANS_DUAL_WEIGHTS --start 1207,1093,940,760,1207,1093,940,760
t d14 --bits 14 --iters 6 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1207,1093,940,760,1207,1093,940,760
t d15 --bits 15 --iters 6 --var HSRANS_DUAL_WEIGHTS_WIDE --start 1207,1093,940,760,1207,1093,940,760
