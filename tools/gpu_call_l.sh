#!/bin/bash
# round-2 GPU call L: occupancy fix of the v2 GEMVs (single-round grids), > 16-stream decode, stepped threshold
set -x
mkdir -p gpurun_out/r02l
export TMPDIR=/tmp
O=gpurun_out/r02l
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_decode_v2.py tests/test_gpu_facade.py "tests/test_gpu_e2e.py" -m gpu -q -x --timeout 900 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_default.log 2>&1
( $B --steps 3 --warmup 1 --no-prefetch ) > $O/bench_noprefetch.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 2 ) > $O/bench_2streams.log 2>&1
( $B --steps 1 --warmup 0 --streams-per-gpu 32 ) > $O/bench_32streams.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off --no-prefetch > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/stats.err
find $GRAFT_REPO_ROOT/$O/stats -name '*kernel_trace.csv' -delete
cd $GRAFT_REPO_ROOT
tail -n 5 $O/tests.log
for f in bench_default bench_noprefetch bench_2streams bench_32streams; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"us_per_layer": [0-9.]*' $O/$f.log) $(grep -o '"avg_launch_us": [0-9.]*' $O/$f.log)"; tail -n 2 $O/$f.log | cut -c1-300; done
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-160
