#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c23
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time LCC_SKIP_SLOW=1 timeout 560 python -m pytest tests -m gpu -q --timeout 500 --durations=12 ) > $O/fast_serial.log 2>&1
grep -E "passed|failed|s call|s setup|real" $O/fast_serial.log | cut -c1-200
