#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash scripts/gpu_session.sh <step> ...'): named measurement steps, each writing
# gpurun_out/<step>.log.  One parameterised script instead of one file per experiment.
#   tests [pytest args]     the -m gpu suite (or the files / -k expression given), log kept
#   choice <tree> ...       scripts/kernel_choice_probe.py for every tree named (ml nj s70 s80 s85 <leaves>:<skew>)
#   bench [args]            python bench.py [args]
#   py <script> [args]      any script under scripts/
set -u
mkdir -p gpurun_out
STEP=$1; shift
case $STEP in
  tests)    timeout 3000 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tee gpurun_out/tests.log | tail -15 ;;
  choice)   for T in "$@"; do timeout 900 python scripts/kernel_choice_probe.py $T 2>&1 | grep -v Warning; done | tee gpurun_out/choice.log ;;
  bench)    timeout 1500 python bench.py "$@" 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-1500 ;;
  py)       S=$1; shift; timeout 1500 python scripts/$S "$@" 2>&1 | tee gpurun_out/$(basename $S .py).log ;;
  *) echo "unknown step $STEP"; exit 2 ;;
esac
