#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c14
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_decode_v2.py -m gpu -q --timeout 800 ) > $O/decode_v2_tests.log 2>&1
tail -n 4 $O/decode_v2_tests.log
B="timeout 600 python bench.py --cpu-baseline off --parity off"
( LCC_RESID_WAVES16=0 $B --steps 2 --warmup 1 --weights fp8 ) > $O/bench_7b_fp8_w8.log 2>&1
( $B --steps 2 --warmup 1 --weights fp8 ) > $O/bench_7b_fp8_w16.log 2>&1
( LCC_RESID_WAVES16=0 $B --steps 2 --warmup 1 --config qwen2vl-2b ) > $O/bench_2b_w8.log 2>&1
( $B --steps 2 --warmup 1 --config qwen2vl-2b ) > $O/bench_2b_w16.log 2>&1
( LCC_RESID_WAVES16=0 $B --steps 1 --warmup 1 --config qwen2vl-72b --weights fp8 --frames 16 ) > $O/bench_72b_fp8_w8.log 2>&1
( $B --steps 1 --warmup 1 --config qwen2vl-72b --weights fp8 --frames 16 ) > $O/bench_72b_fp8_w16.log 2>&1
for f in bench_7b_fp8_w8 bench_7b_fp8_w16 bench_2b_w8 bench_2b_w16 bench_72b_fp8_w8 bench_72b_fp8_w16; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"avg_launch_us": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
