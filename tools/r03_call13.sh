#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c13
mkdir -p $O
export TMPDIR=/tmp
cd $R
( LCC_ATTN_FUSED_WAVES=8 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --timeout 500 -k "fused or multi_stream or batch" ) > $O/fused_tests.log 2>&1
tail -n 4 $O/fused_tests.log
B="timeout 500 python bench.py --cpu-baseline off --parity off"
( LCC_ATTN_FUSED_WAVES=4 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_w4.log 2>&1
( LCC_ATTN_FUSED_WAVES=8 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_w8.log 2>&1
( LCC_ATTN_FUSED_WAVES=8 LCC_ATTN_FUSED_BLOCKS=512 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_w8_fb512.log 2>&1
( LCC_ATTN_FUSED_WAVES=8 LCC_ATTN_FUSED_BLOCKS=128 $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_w8_fb128.log 2>&1
( LCC_ATTN_FUSED_WAVES=8 $B --steps 2 --warmup 1 --streams-per-gpu 4 ) > $O/bench_4s_w8.log 2>&1
( LCC_ATTN_FUSED_WAVES=4 $B --steps 2 --warmup 1 --streams-per-gpu 4 ) > $O/bench_4s_w4.log 2>&1
for f in bench_8s_w4 bench_8s_w8 bench_8s_w8_fb512 bench_8s_w8_fb128 bench_4s_w8 bench_4s_w4; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
