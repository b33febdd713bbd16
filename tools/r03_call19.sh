#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c19
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -x -k "tall" ) > $O/tall_tests.log 2>&1
tail -n 6 $O/tall_tests.log
for RING in 4 2; do LCC_TALL_RING=$RING timeout 200 python tools/bench_tall.py 2>/dev/null | grep '^{' | grep -E '"auto|"tall' | sed "s/^/ring$RING /" | tee -a $O/gemm_tall_ring.txt; done
B="timeout 500 python bench.py --cpu-baseline off --parity off"
( LCC_TALL_RING=4 $B --steps 3 --warmup 1 ) > $O/bench_1s_ring4.log 2>&1
( LCC_TALL_RING=2 $B --steps 3 --warmup 1 ) > $O/bench_1s_ring2.log 2>&1
for f in bench_1s_ring4 bench_1s_ring2; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
