R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in f32 bf16x3; do
  export FRX_HEAD_GEMM=$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp_$k -- python3 $R/scripts/head_bench.py arcface 256 10575 50 > $O/hp_$k.log 2>&1
  f=$(find $O/hp_$k -name "*kernel_stats.csv" | head -1); cp "$f" $O/hp_${k}_kernel_stats.csv; rm -rf $O/hp_$k
done
