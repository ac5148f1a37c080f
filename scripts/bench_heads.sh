#!/bin/bash
# One bench.py line per margin head (headline shape: C=10575, bs 256, bf16, hipGraph) -> gpurun_out/bench_heads.log
set -e
: > gpurun_out/bench_heads.log
for h in arcface cosface sphereface curricular mv_am mv_arc adaface elastic_arc elastic_cos magface vpl_arcface; do
  extra=""; [ "$h" = magface ] && extra="--lambda-g 35"
  python bench.py --head $h $extra --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$h', d['value'], d['ms_per_step'], d['config'].get('final_loss'))" >> gpurun_out/bench_heads.log
done
cat gpurun_out/bench_heads.log
