#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r03_counters_avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -c . gpurun_out/r03_counters_avail.txt
grep -o "Name:[[:space:]]*\(TA_\|TCP_\|TD_\|SQ_\)[A-Za-z0-9_]*" gpurun_out/r03_counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000
