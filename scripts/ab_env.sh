#!/bin/bash
# A/B environment settings of bench.py on ONE box, alternating runs (the boxes of the pool differ by ~2 %).
#   bash scripts/ab_env.sh OUT_FILE "FRX_X=0" "FRX_X=1" ["FRX_X=1 FRX_Y=2" ...]      (each argument: one variant's settings)
set -e
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
for r in 1 2 3; do
  for v in "$@"; do
    echo -n "[$v] " >> "$OUT"
    env $v python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> "$OUT"
  done
done
cat "$OUT"
