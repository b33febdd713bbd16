#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c21
mkdir -p $O
export TMPDIR=/tmp
cd $R
( LCC_GEMM_SCHED=6 LCC_TALL_SCHED=2 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -x -k "gemm or linear or tall" ) > $O/gemm_tests.log 2>&1
tail -n 4 $O/gemm_tests.log
for S in 1 6; do LCC_GEMM_SCHED=$S timeout 200 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed "s/^/gemm_sched$S /" | tee -a $O/gemm_sched.txt; done
for S in 0 1 2; do LCC_TALL_SCHED=$S timeout 200 python tools/bench_tall.py 2>/dev/null | grep '^{' | grep -E '"tall"' | sed "s/^/tall_sched$S /" | tee -a $O/gemm_sched.txt; done
B="timeout 500 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_1s_base.log 2>&1
( LCC_TALL_SCHED=2 LCC_GEMM_SCHED=6 $B --steps 3 --warmup 1 ) > $O/bench_1s_spread.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_base.log 2>&1
( LCC_TALL_SCHED=2 LCC_GEMM_SCHED=6 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_spread.log 2>&1
for f in bench_1s_base bench_1s_spread bench_8s_base bench_8s_spread; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
