#!/bin/bash
mkdir -p gpurun_out scripts/micro/bin
hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/sector_pair scripts/micro/sector_pair.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for C in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sp_$N -- $GRAFT_REPO_ROOT/scripts/micro/bin/sector_pair > $GRAFT_REPO_ROOT/gpurun_out/sp_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list)); order=[]
for f in sorted(glob.glob("gpurun_out/sp_*/**/*counter_collection.csv", recursive=True)):
    per=collections.defaultdict(lambda: collections.defaultdict(float)); name={}
    for r in csv.DictReader(open(f)):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])]=r["Kernel_Name"].split("(")[0]
    for d in sorted(per):
        key=(name[d], (d-1)//16)
        for k,v in per[d].items(): acc[(name[d], d)][k]=v
lanes=256*8*256*256.0
for (kn,d),c in sorted(acc.items(), key=lambda x:x[0][1]):
    print(d, kn, {k: round(v/lanes,3) for k,v in c.items()})
PY
