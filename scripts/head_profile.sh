R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/head; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python3 $R/scripts/head_bench.py arcface 256 10575 50 > $O/hp.log 2>&1
f=$(find $O/hp -name "*kernel_stats.csv" | head -1); cp "$f" $O/hp_kernel_stats.csv; rm -rf $O/hp
