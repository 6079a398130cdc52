set -u
run() { TGP_LIB=$1 timeout 200 python bench.py --m-per-gpu 262144 --steps 3 --no-cpu-baseline --no-acquire 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'kernel_ms', round(o['roofline']['kernel_ms'],2), 'frac', round(o['roofline']['frac'],4), o['config']['best_index'])"; }
run "" "shipped dma"
for T in "$@"; do run $PWD/tools/exp/libtgp_$T.so "$T"; done
