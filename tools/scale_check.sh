#!/bin/bash
# One-command scaling check for the day a multi-GPU node is at hand.  For every workload in
# {headline, c3, c5} and both multi-GPU forms (ranks: one process per GPU over torch.distributed/RCCL; group: one
# process, tgp_group_*) it runs bench.py at N = 1, 2, 4, 8 GPUs (those that exist) with a FIXED total candidate set
# (--scaling strong: the winner must not depend on how the candidates are sharded), checks that the winning
# (value, index) is identical across N and across the two forms, and prints one table with the whole-job rate, the
# speed-up over N = 1 and every rank's own kernel time.  It makes no scaling claim by itself: it prints what it measured.
#   usage: tools/scale_check.sh [--gpus "1 2 4 8"] [--workloads "headline c3 c5"] [--modes "ranks group"] [--steps K]
#          [--m-per-gpu M]   (M: units per GPU of the 8-GPU job; default = the workload's own)
set -u
cd "$(dirname "$0")/.."
GPUS="1 2 4 8"; WORKLOADS="headline c3 c5"; MODES="ranks group"; STEPS=3; MPG=0
while [ $# -gt 0 ]; do
  case "$1" in
    --gpus) GPUS="$2"; shift 2;; --workloads) WORKLOADS="$2"; shift 2;; --modes) MODES="$2"; shift 2;;
    --steps) STEPS="$2"; shift 2;; --m-per-gpu) MPG="$2"; shift 2;; *) echo "unknown option $1"; exit 2;;
  esac
done
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
OUT=${SCALE_OUT:-gpurun_out/scale_check}; mkdir -p "$OUT"
rc=0
for w in $WORKLOADS; do for mode in $MODES; do for n in $GPUS; do
  [ "$n" -gt "$HAVE" ] && { echo "skip $w $mode N=$n: only $HAVE GPU(s) visible"; continue; }
  f="$OUT/${w}_${mode}_${n}.json"
  extra=""; [ "$MPG" != "0" ] && extra="--m-per-gpu $MPG"
  python bench.py --gpus "$n" --steps "$STEPS" --warmup 1 --workload "$w" --mode "$mode" --scaling strong \
         --no-cpu-baseline --no-secondary --no-acquire $extra > "$f" 2> "$f.err" || { echo "FAILED $w $mode N=$n (see $f.err)"; rc=1; }
done; done; done
python - "$OUT" <<'PY' || rc=1
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    w, mode, n = os.path.basename(f)[:-5].rsplit("_", 2)
    rows.setdefault(w, []).append((mode, int(n), j))
bad = 0
print(f"{'workload':9s} {'mode':6s} {'N':>2s} {'value':>12s} {'unit':28s} {'x N=1':>6s} {'ms/step':>8s}  winner (value, index)            kernel ms per rank")
for w, rs in rows.items():
    winners = {(j['config']['best_value'], j['config']['best_index']) for _, _, j in rs}
    base = {mode: j['value'] for mode, n, j in rs if n == 1}
    for mode, n, j in sorted(rs, key=lambda r: (r[0], r[1])):
        c = j['config']
        sp = j['value'] / base[mode] if mode in base else float('nan')
        print(f"{w:9s} {mode:6s} {n:2d} {j['value']:12.4e} {j['unit']:28s} {sp:6.2f} {j['ms_per_step']:8.2f}  "
              f"({c['best_value']:.12e}, {c['best_index']})  {[round(x, 2) for x in c['kernel_ms_per_rank']]} rccl_ranks={c['rccl_ranks']}")
    if len(winners) != 1:
        bad += 1
        print(f"  !! {w}: the winner depends on the sharding / the form: {sorted(winners)}")
    else:
        print(f"  ok {w}: one winner across {len(rs)} run(s)")
sys.exit(1 if bad or not rows else 0)
PY
exit $rc
