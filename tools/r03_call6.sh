#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c6
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_decode_v2.py -m gpu -q -x --timeout 500 -k chained ) > $O/decode_v2_tests.log 2>&1
tail -n 3 $O/decode_v2_tests.log
B="timeout 400 python bench.py --cpu-baseline off --parity off --steps 2 --warmup 1 --no-prefetch"
( $B --decode-chain 0 ) > $O/bench_chain0.log 2>&1
for D in 0 10 16 20 24; do
  ( LCC_CHAIN_DELAY_US=$D $B --decode-chain 1 ) > $O/bench_chain1_d$D.log 2>&1
done
( $B --decode-chain 1 ) > $O/bench_chain1_auto.log 2>&1
for f in bench_chain0 bench_chain1_d0 bench_chain1_d10 bench_chain1_d16 bench_chain1_d20 bench_chain1_d24 bench_chain1_auto; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | head -1)"; done
