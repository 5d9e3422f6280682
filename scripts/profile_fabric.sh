#!/bin/bash
# Runs ON THE GPU BOX: fabric read requests / L2 hits and misses of the st:: kernels of one tune_gpu.py run.
# usage: scripts/profile_fabric.sh <tag> <tune_gpu.py args...>
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/fab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SUCHTREE_AMD_AUTOTUNE=0      # (the timing launches of host_tune.h would be counted under the profiled kernel's name)
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
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
