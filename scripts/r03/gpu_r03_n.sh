#!/bin/bash
set -u
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r03n_pytest_all.log 2>&1
tail -8 gpurun_out/r03n_pytest_all.log
