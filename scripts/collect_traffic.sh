#!/bin/bash
# HBM traffic of the bench's kernels from the L2 memory-side counters, one counter per pass
# (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass).
# Run on the GPU box from the repo root:  bash scripts/collect_traffic.sh   -> gpurun_out/traffic/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/scripts/traffic_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json | head -c 1500
