#!/bin/bash
for T in bigdeep ml nj; do timeout 200 python scripts/tune_gpu.py --tree $T --pairs 10000000 --strategy walk --rounds 4 --opt walk_climb_a=0,1 2>&1 | grep "median"; done
