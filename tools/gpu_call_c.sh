#!/bin/bash
# round-2 GPU call C: decode v2 after the burst-load fix: quick parity, A/B bench, kernel trace; plugin + sampling tests
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py tests/test_gpu_baseline_configs.py -m gpu -q -x --timeout 1200 -k "v2 or tiny or plugin or sampling or golden" ) > gpurun_out/test_c.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_c.log
( time python bench.py --steps 2 --warmup 1 --cpu-baseline off --decode-path 1 ) > gpurun_out/bench_v2.log 2>&1
( time python bench.py --steps 2 --warmup 1 --cpu-baseline off --decode-path 0 ) > gpurun_out/bench_v1.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames_v2.csv \;
for f in test_c bench_v2 bench_v1; do echo "== $f"; tail -n 6 gpurun_out/$f.log | cut -c1-600; done
