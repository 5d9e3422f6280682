#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + the PMC passes the
# roofline.traffic figure comes from, and the calibration passes that give the bytes per
# fabric request for the two access patterns involved.  Each counter set gets its own run,
# with --kernel-trace only (no sys/hip/hsa tracing next to --pmc).
# usage: scripts/profile_gpu.sh <round-tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
ARGS=${@:-"--steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-microbench --no-other-configs"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT      # (raw outputs of an earlier call must not be averaged in)
mkdir -p $OUT $REPO/scripts/micro/bin
# the calibration program (known traffic) is built on demand; binaries are not kept in git
[ -x $REPO/scripts/micro/bin/calib_requests ] || hipcc --offload-arch=gfx950 -O3 -o $REPO/scripts/micro/bin/calib_requests $REPO/scripts/micro/calib_requests.hip
cd /tmp && export TMPDIR=/tmp
export SUCHTREE_AMD_AUTOTUNE=0      # (the timing launches of host_tune.h would be counted under the profiled kernel's name)
timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
# (the last two sets, round 6: the CU side of the launch -- L1-miss requests, TA busy, TCP stalled on pending misses, miss latency)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum" ; do
  N=$(echo $C | tr ' ' '_')
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?"
  # the same counters on two launches of exactly known traffic (scripts/micro/calib_requests.hip)
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$N -- $REPO/scripts/micro/bin/calib_requests > $OUT/cal_$N.log 2>&1
  echo "cal $N rc=$?"
done
cd $REPO
python3 scripts/summarize_profile.py $OUT $TAG || true
mkdir -p gpurun_out/round_profiles
cp profiles/kernel_stats_$TAG.csv profiles/traffic_$TAG.json profiles/calibration_$TAG.json gpurun_out/round_profiles/ 2>/dev/null
