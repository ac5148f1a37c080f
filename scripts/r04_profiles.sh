#!/bin/bash
# The round-4 measurement set behind profiles/r04_*: run on the GPU box from the repo root, outputs under gpurun_out/final/.
#   bash scripts/r04_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
bash scripts/boxinfo.sh > $O/boxinfo.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.log
FRX_BN_DETERMINISTIC=1 python bench.py --no-cpu-baseline > $O/bench_deterministic_bn.json 2> /dev/null
python scripts/layer_times.py > $O/layer_times.txt 2>&1
for cfg in "cosface 10575 256" "curricular 85000 128" "curricular 85742 128"; do
  set -- $cfg
  python bench.py --head $1 --classes $2 --batch $3 --steps 30 --no-cpu-baseline 2>/dev/null >> $O/bench_configs.jsonl
done
python bench.py --split --steps 20 --no-cpu-baseline > $O/bench_split_one_rank.json 2> $O/bench_split.log
for k in "arcface 256 10575 50" "cosface 256 10575 50" "curricular 128 85000 20"; do python scripts/head_bench.py $k >> $O/head_bench.txt 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/kstats.log 2>&1
f=$(find $O/kstats -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats.csv; rm -rf $O/kstats
cd $R
bash scripts/collect_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/summary.json $O/traffic_summary.json
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE
ls $O
