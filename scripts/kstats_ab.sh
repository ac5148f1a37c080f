#!/bin/bash
# rocprofv3 kernel statistics of the bench under two environments (A/B of a code path selected by an environment variable).
#   bash scripts/kstats_ab.sh NAME_A "ENV_A=.." NAME_B "ENV_B=.."     -> gpurun_out/kab/{NAME}_kernel_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/$name.log 2>&1 )
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${name}_kernel_stats.csv
  rm -rf $O/$name
}
run "$1" "$2"
run "$3" "$4"
ls $O
