#!/bin/bash
# Runs ON THE GPU BOX: SQ/LDS counters of the canopy kernels on a chosen tree.
# usage: scripts/profile_counters.sh <tag> <tune_gpu.py args...>
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/cnt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SUCHTREE_AMD_AUTOTUNE=0      # (the timing launches of host_tune.h would be counted under the profiled kernel's name)
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" ; do
  N=$(echo $C | cut -d' ' -f1)
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $REPO/scripts/tune_gpu.py "$@" > $OUT/$N.log 2>&1
  echo "pmc $N rc=$?"
done
cd $REPO
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float)); name={}
    for r in csv.DictReader(open(f)):
        if "st::k_" not in r["Kernel_Name"]: continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]]=r["Kernel_Name"].split("(")[0]
    for d,c in per.items():
        for k,v in c.items(): acc[name[d]][k].append(v)
for kn,c in acc.items():
    print(kn)
    for k,v in sorted(c.items()): print("   %-26s %.4g (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
