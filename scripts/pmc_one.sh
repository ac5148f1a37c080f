#!/bin/bash
# PMC counters of ONE conv shape (scripts/one_conv.py), one counter group per pass.  Run on the GPU box from the repo root:
#   bash scripts/pmc_one.sh TAG Ci Co k stride Hi mode     -> gpurun_out/pmc_TAG/*.csv + gpurun_out/pmc_TAG.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/scripts/one_conv.py "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp" >> $OUT/fail.txt
done
python3 - "$OUT" <<'PY' > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_igemm" not in k and "k_wgrad" not in k: continue
        acc[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
cat $R/gpurun_out/pmc_$TAG.txt
