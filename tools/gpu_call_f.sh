#!/bin/bash
# round-2 GPU call F: facade / batched / plugin parity tests, decode v2 with pinned load order: bench + trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_facade.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout 1200 -k "facade or batch or plugin or v2 or sampling or video_qa or once" ) > gpurun_out/test_f.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_f.log
B="python bench.py --steps 2 --warmup 1 --cpu-baseline off"
( $B --decode-path 1 ) > gpurun_out/f_path1.log 2>&1
( $B --decode-path 0 ) > gpurun_out/f_path0.log 2>&1
( $B --streams-per-gpu 2 ) > gpurun_out/f_2s.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames_v2.csv \;
tail -n 6 gpurun_out/test_f.log
for f in f_path1 f_path0 f_2s; do echo "== $f $(grep -o '"value": [0-9.]*' gpurun_out/$f.log) $(grep -o '"us_per_layer": [0-9.]*' gpurun_out/$f.log)"; done
