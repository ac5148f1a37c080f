#!/bin/bash
# A/B two builds of libfrx.so on one box: scripts/_ab/libfrx_base.so vs scripts/_ab/libfrx_new.so (alternating runs).
set -e
L=face-recognition-models_amd/frx/libfrx.so
for r in 1 2 3; do
  for v in base new; do
    cp scripts/_ab/libfrx_$v.so $L
    echo -n "$v " >> gpurun_out/ab.log
    python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab.log
  done
done
cp scripts/_ab/libfrx_new.so $L
cat gpurun_out/ab.log
