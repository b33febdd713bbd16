#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c3
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or attention or prefill" --timeout 500 ) > $O/attn_tests.log 2>&1
tail -n 4 $O/attn_tests.log
( timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -k "prefill_attention_long_cache" --timeout 500 ) > $O/attn_long_tests.log 2>&1
tail -n 4 $O/attn_long_tests.log
timeout 300 python tools/bench_attn.py 2>$O/bench_attn.err | grep '^{' > $O/attn_prefill_microbench.jsonl
grep -E '"variant": 3' $O/attn_prefill_microbench.jsonl | grep -v long_cache
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 2 --warmup 1 --no-prefetch --attn-variant 3 ) > $O/bench_1s_v3.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 --attn-variant 3 ) > $O/bench_8s_v3.log 2>&1
( $B --steps 1 --warmup 1 --workload oneshot480 --attn-variant 3 ) > $O/bench_oneshot480_v3.log 2>&1
for f in bench_1s_v3 bench_8s_v3 bench_oneshot480_v3; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; done
