#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c15
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())" 2>/dev/null
B="timeout 500 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_1s_default.log 2>&1
( $B --steps 3 --warmup 1 --main-stream-priority high ) > $O/bench_1s_high.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_default.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 8 --main-stream-priority high ) > $O/bench_8s_high.log 2>&1
for f in bench_1s_default bench_1s_high bench_8s_default bench_8s_high; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | tr '\n' ' ') $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; tail -n 2 $O/$f.log | grep -v '^{' | cut -c1-300; done
