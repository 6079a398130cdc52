set -u
run() { TGP_LIB=$1 timeout 200 python bench.py --precision i8x4 --m-per-gpu 262144 --steps 3 --no-cpu-baseline --no-acquire 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'kernel_ms', round(o['roofline']['kernel_ms'],2), 'frac', round(o['roofline']['frac'],4), o['config']['best_index'])"; }
run "" "i8 shipped"
run $PWD/tools/exp/libtgp_x1024.so "no slab loads"
run $PWD/tools/exp/libtgp_x2048.so "no W-plane loads"
run $PWD/tools/exp/libtgp_x3072.so "no loads"
run $PWD/tools/exp/libtgp_x4096.so "no barrier"
run $PWD/tools/exp/libtgp_x8192.so "no LDS staging stores"
run $PWD/tools/exp/libtgp_x16384.so "no generation"
