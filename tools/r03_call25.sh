#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c25
mkdir -p $O
export TMPDIR=/tmp
cd $R
LCC_GEMM_SCHED=1 timeout 120 python tools/gemm_checksum.py > $O/sum1.txt 2>$O/sum1.err
LCC_GEMM_SCHED=6 timeout 120 python tools/gemm_checksum.py > $O/sum6.txt 2>$O/sum6.err
paste $O/sum1.txt $O/sum6.txt; cmp $O/sum1.txt $O/sum6.txt && echo "CHECKSUMS IDENTICAL"
( time LCC_GEMM_SCHED=6 LCC_SKIP_SLOW=1 timeout 330 python -m pytest tests -m gpu -q --timeout 300 -x ) > $O/fast_serial_sched6.log 2>&1
grep -E "passed|failed|real" $O/fast_serial_sched6.log | cut -c1-200
