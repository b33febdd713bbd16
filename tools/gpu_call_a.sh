#!/bin/bash
# round-2 GPU call A: new parity tests first (fail fast), then the full suite, the default bench line, a short kernel trace
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x --timeout 1200 ) > gpurun_out/test_new.log 2>&1
echo "new tests rc=$?" >> gpurun_out/test_new.log
( time python -m pytest tests -m gpu -q --timeout 1200 --deselect tests/test_gpu_baseline_configs.py ) > gpurun_out/test_all.log 2>&1
echo "all tests rc=$?" >> gpurun_out/test_all.log
( time python bench.py --steps 3 --warmup 1 ) > gpurun_out/bench_default.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames.csv \;
tail -5 gpurun_out/test_new.log gpurun_out/test_all.log gpurun_out/bench_default.log
