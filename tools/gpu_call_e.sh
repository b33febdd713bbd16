#!/bin/bash
# round-2 GPU call E: decode v2 (LDS-staged rows, separate combine) parity + bench; tuning sweeps: attention split, GEMM variants
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py tests/test_gpu_golden.py -m gpu -q -x --timeout 1200 -k "v2 or tiny or small or plugin or golden or 2b" ) > gpurun_out/test_e.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_e.log
B="python bench.py --steps 2 --warmup 1 --cpu-baseline off"
( $B --decode-path 1 ) > gpurun_out/e_path1.log 2>&1
( $B --decode-path 0 ) > gpurun_out/e_path0.log 2>&1
( LCC_ATTN_TPS=2 LCC_ATTN_MAXSPLIT=128 $B ) > gpurun_out/e_tps2.log 2>&1
( LCC_ATTN_TPS=8 $B ) > gpurun_out/e_tps8.log 2>&1
( $B --gemm-variant 3 ) > gpurun_out/e_gemm3.log 2>&1
( $B --gemm-variant 7 ) > gpurun_out/e_gemm7.log 2>&1
( $B --attn-variant 1 ) > gpurun_out/e_attn1.log 2>&1
( $B --gemv-variant 2 ) > gpurun_out/e_gemv2.log 2>&1
( $B --streams-per-gpu 2 ) > gpurun_out/e_2s.log 2>&1
( $B --streams-per-gpu 4 ) > gpurun_out/e_4s.log 2>&1
( $B --streams-per-gpu 4 --decode-path 0 ) > gpurun_out/e_4s_path0.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 10 --cpu-baseline off ) > gpurun_out/trace.log 2>&1
find /tmp/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/kernel_trace_10frames_v2.csv \;
tail -n 4 gpurun_out/test_e.log
for f in e_path1 e_path0 e_tps2 e_tps8 e_gemm3 e_gemm7 e_attn1 e_gemv2 e_2s e_4s e_4s_path0; do echo "== $f $(grep -o '"value": [0-9.]*' gpurun_out/$f.log) $(grep -o '"us_per_layer": [0-9.]*' gpurun_out/$f.log)"; done
