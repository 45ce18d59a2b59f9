O=gpurun_out/tune; mkdir -p $O
t() { name=$1; shift; timeout 400 python tools/tune_weights.py "$@" > $O/$name.txt 2>&1; echo "== $name: $*"; tail -4 $O/$name.txt; }
t s32b11 --bits 11 --states 32 --iters 6 --var HSRANS_DIRECT_WEIGHTS_PAIR --start 1898,1721,1464,1139,729,491,329,229
