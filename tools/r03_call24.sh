#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c24
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_1s_base.log 2>&1
( LCC_GEMM_SCHED=6 $B --steps 3 --warmup 1 ) > $O/bench_1s_spread.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_base.log 2>&1
( LCC_GEMM_SCHED=6 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_spread.log 2>&1
for f in bench_1s_base bench_1s_spread bench_8s_base bench_8s_spread; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
