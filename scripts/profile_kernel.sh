#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace (--stats) + PMC passes of ONE kernel of a scripts/tune_gpu.py
# run (any tree / strategy), condensed into profiles/kernel_stats_<tag>.csv and profiles/traffic_<tag>.json.
# Every pass runs under `timeout`; each counter set gets its own run with --kernel-trace only.
# usage: scripts/profile_kernel.sh <tag> <kernel-name-substring> <pairs> <tune_gpu.py args...>
set -u
TAG=$1; KERN=$2; PAIRS=$3; shift 3
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SUCHTREE_AMD_AUTOTUNE=0      # (the timing launches of host_tune.h would count as launches of the profiled kernel)
T=${PMC_TIMEOUT:-240}
timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/scripts/tune_gpu.py --pairs $PAIRS --rounds 4 "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?"
# (PMC_SETS: semicolon-separated counter sets, one pass each, instead of the default nine)
DEFAULT_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum;TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum;GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
IFS=';' read -ra SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for C in "${SETS[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout $T rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/scripts/tune_gpu.py --pairs $PAIRS --rounds 3 "$@" > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?"
done
cd $REPO
python3 scripts/summarize_profile.py $OUT $TAG $KERN $PAIRS > $OUT/summary.json || true
python3 - <<PY
import json
d=json.load(open("profiles/traffic_$TAG.json"))
k=d.get("kernel_full_name"); print(k, d["kernels"].get(k))
print({a: round(b, 3) for a, b in d.get("counters_per_pair", {}).items()})
PY
