#!/bin/bash
# Runs ON THE GPU BOX: arbitrary PMC counter sets (one pass per quoted set) over one tune_gpu.py run.
# Every pass runs under `timeout` (a counter set the hardware cannot collect makes rocprofv3 abort and then hang).
# usage: COUNTERS="A B;C D" scripts/profile_pmc.sh <tag> <tune_gpu.py args...>
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SUCHTREE_AMD_AUTOTUNE=0      # (the timing launches of host_tune.h would be counted under the profiled kernel's name)
IFS=';' read -ra SETS <<< "$COUNTERS"
for C in "${SETS[@]}"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $REPO/scripts/tune_gpu.py "$@" > $OUT/$N.log 2>&1
  echo "pmc $N rc=$?"; grep -i "error\|invalid\|not found" $OUT/$N.log | head -2
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
    for k,v in sorted(c.items()): print("   %-34s %.4g (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
