#!/bin/bash
# SQ counters of ONE igemm launch shape (scripts/one_conv.py), separate rocprofv3 --pmc passes of <= 8 SQ counters each.
# Usage: bash scripts/pmc_conv.sh OUTDIR Ci Co k stride Hi kind
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/one_conv.py "$@" 10 > $R/$OUT/time.txt 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM? SQ_INSTS_WAVE32_LDS"; do
  C=$(echo $C | tr -d '?')
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/$OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/one_conv.py "$@" 3 > $R/$OUT/p$i.log 2>&1 || echo "pass $i failed" >> $R/$OUT/time.txt
done
python3 - "$R/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_igemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/time.txt").read())
    for k, cs in acc.items():
        o.write(k + "\n")
        for c, v in sorted(cs.items()):
            o.write(f"  {c:34s} {sum(v) / len(v):16.0f}  (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
