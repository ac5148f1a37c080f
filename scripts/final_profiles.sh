#!/bin/bash
# The per-round measurement set behind profiles/rNN_*: run on the GPU box from the repo root, outputs under gpurun_out/final/.
#   bash scripts/final_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.log
python scripts/layer_times.py > $O/layer_times.txt 2>&1
python scripts/extra_measurements.py > $O/extra_measurements.json 2> $O/extra.log
bash scripts/bench_heads.sh > /dev/null 2>&1; cp gpurun_out/bench_heads.log $O/bench_heads.log
python scripts/gemm_reference_points.py > $O/gemm_reference_points.txt 2>&1
python scripts/stem_bwd_probe.py > $O/stem_bwd_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/kstats.log 2>&1
cd $R
bash scripts/collect_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/summary.json $O/traffic_summary.json
ls $O
