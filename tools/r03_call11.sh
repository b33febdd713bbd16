#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c11
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -k "vit_rope_attention" ) > $O/vit_tests.log 2>&1
tail -n 5 $O/vit_tests.log
( timeout 400 python tools/bench_dgemv_intercept.py ) > $O/dgemv_intercept.jsonl 2> $O/dgemv_intercept.err
tail -n 20 $O/dgemv_intercept.jsonl; tail -n 3 $O/dgemv_intercept.err
B="timeout 500 python bench.py --cpu-baseline off --parity off"
( LCC_VIT32_MIN_BLOCKS4=128 $B --steps 3 --warmup 1 ) > $O/bench_1s_vit32w4.log 2>&1
( LCC_VIT32_MIN_BLOCKS4=1000000 $B --steps 3 --warmup 1 ) > $O/bench_1s_vit_old.log 2>&1
( LCC_VIT32_MIN_BLOCKS4=128 $B --steps 2 --warmup 1 --no-prefetch ) > $O/bench_1s_vit32w4_nopf.log 2>&1
( LCC_VIT32_MIN_BLOCKS4=1000000 $B --steps 2 --warmup 1 --no-prefetch ) > $O/bench_1s_vit_old_nopf.log 2>&1
for FB in 256 512 1024; do
  ( LCC_ATTN_FUSED_BLOCKS=$FB $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_fb$FB.log 2>&1
done
for f in bench_1s_vit32w4 bench_1s_vit_old bench_1s_vit32w4_nopf bench_1s_vit_old_nopf bench_8s_fb256 bench_8s_fb512 bench_8s_fb1024; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
