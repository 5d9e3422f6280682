#!/bin/bash
# Runs ON THE GPU BOX (round 6): the counters of the random-sector microbenchmark (the ceiling of roofline.secondary_ceiling) at the
# headline kernel's 20 MiB footprint, to compare kernel and ceiling in ONE unit: L1-miss requests (TCP_TCC_READ_REQ), fabric
# requests (TCC_EA0_RDREQ), TA busy, TCP stalled on pending misses -- per lane read and per second.
# usage: scripts/sector_ceiling_counters.sh [MiB] [round tag]  -> profiles/ceiling_counters_<tag>.json
MIB=${1:-20}
TAG=${2:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/ceiling_cnt
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $REPO/scripts/sector_ceiling_probe.py $MIB > $OUT/$N.log 2>&1
  echo "pmc $N rc=$?"
done
cd $REPO
grep "G reads/s" $OUT/TCP_TCC_READ_REQ_sum_TCC_EA0_RDREQ_sum.log
python3 scripts/sector_ceiling_summary.py $OUT $MIB $TAG
mkdir -p gpurun_out/round_profiles; cp profiles/ceiling_counters_${TAG}.json gpurun_out/round_profiles/
