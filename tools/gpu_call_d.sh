#!/bin/bash
# round-2 GPU call D: decode v2 with LDS-staged activations + fused attention combine: parity, A/B/C bench, kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py tests/test_gpu_golden.py -m gpu -q -x --timeout 1200 -k "v2 or tiny or small or plugin or golden or 2b" ) > gpurun_out/test_d.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_d.log
for p in 1 2 0; do ( time python bench.py --steps 2 --warmup 1 --cpu-baseline off --decode-path $p ) > gpurun_out/bench_path$p.log 2>&1; done
( time python bench.py --steps 1 --warmup 1 --cpu-baseline off --streams-per-gpu 2 --decode-path 1 ) > gpurun_out/bench_2s_path1.log 2>&1
( time python bench.py --steps 1 --warmup 1 --cpu-baseline off --streams-per-gpu 2 --decode-path 0 ) > gpurun_out/bench_2s_path0.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames_v2.csv \;
for f in test_d bench_path1 bench_path2 bench_path0 bench_2s_path1 bench_2s_path0; do echo "== $f"; tail -n 5 gpurun_out/$f.log | cut -c1-300; done
