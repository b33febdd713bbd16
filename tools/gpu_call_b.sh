#!/bin/bash
# round-2 GPU call B: decode pipeline v2 -- parity first, then A/B bench against the round-1 launch sequence, short kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --timeout 1200 -k "v2 or tiny or small or do_sample" ) > gpurun_out/test_v2.log 2>&1
echo "v2 tests rc=$?" >> gpurun_out/test_v2.log
( time python -m pytest tests -m gpu -q --timeout 1500 ) > gpurun_out/test_all.log 2>&1
echo "all tests rc=$?" >> gpurun_out/test_all.log
( time python bench.py --steps 2 --warmup 1 --cpu-baseline off --decode-path 1 ) > gpurun_out/bench_v2.log 2>&1
( time python bench.py --steps 2 --warmup 1 --cpu-baseline off --decode-path 0 ) > gpurun_out/bench_v1.log 2>&1
( time python bench.py --steps 1 --warmup 1 --cpu-baseline off --streams-per-gpu 8 --decode-path 1 ) > gpurun_out/bench_v2_8s.log 2>&1
( time python bench.py --steps 1 --warmup 1 --cpu-baseline off --streams-per-gpu 8 --decode-path 0 ) > gpurun_out/bench_v1_8s.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames_v2.csv \;
find /tmp/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_10frames_v2.csv \;
for f in test_v2 test_all bench_v2 bench_v1 bench_v2_8s bench_v1_8s; do echo "== $f"; tail -n 4 gpurun_out/$f.log; done
