#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c12
mkdir -p $O
export TMPDIR=/tmp
cd $R
for W in 0 2; do
  ( LCC_RESID_WAVES16=$W timeout 300 python tools/bench_dgemv_intercept.py down o --bf16 ) > $O/dgemv_intercept_w16_$W.jsonl 2> $O/dgemv_intercept_$W.err
  echo "== LCC_RESID_WAVES16=$W"; cat $O/dgemv_intercept_w16_$W.jsonl; tail -n 2 $O/dgemv_intercept_$W.err
done
( LCC_RESID_WAVES16=2 timeout 600 python -m pytest tests/test_gpu_decode_v2.py -m gpu -q --timeout 500 -k "resid or layer or pipeline" ) > $O/v2_tests.log 2>&1
tail -n 4 $O/v2_tests.log
B="timeout 500 python bench.py --cpu-baseline off --parity off"
for W in 0 1 2; do
  ( LCC_RESID_WAVES16=$W $B --steps 3 --warmup 1 ) > $O/bench_1s_w16_$W.log 2>&1
done
for FB in 128 64; do
  ( LCC_ATTN_FUSED_BLOCKS=$FB $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_fb$FB.log 2>&1
done
for f in bench_1s_w16_0 bench_1s_w16_1 bench_1s_w16_2 bench_8s_fb128 bench_8s_fb64; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
