#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c5
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_decode_v2.py -m gpu -q -x --timeout 500 ) > $O/decode_v2_tests.log 2>&1
tail -n 5 $O/decode_v2_tests.log
( LCC_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_rccl.py tests/test_gpu_golden.py -m gpu -q -x --timeout 800 ) > $O/e2e_tests.log 2>&1
tail -n 5 $O/e2e_tests.log
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 2 --warmup 1 --no-prefetch --decode-chain 0 ) > $O/bench_1s_chain0_nopf.log 2>&1
( $B --steps 2 --warmup 1 --no-prefetch --decode-chain 1 ) > $O/bench_1s_chain1_nopf.log 2>&1
( $B --steps 3 --warmup 1 --decode-chain 0 ) > $O/bench_1s_chain0.log 2>&1
( $B --steps 3 --warmup 1 --decode-chain 1 ) > $O/bench_1s_chain1.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 2 --decode-chain 0 ) > $O/bench_2s_chain0.log 2>&1
( $B --steps 2 --warmup 1 --streams-per-gpu 2 --decode-chain 1 ) > $O/bench_2s_chain1.log 2>&1
for f in bench_1s_chain0_nopf bench_1s_chain1_nopf bench_1s_chain0 bench_1s_chain1 bench_2s_chain0 bench_2s_chain1; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"avg_step_us": [0-9.]*' $O/$f.log | head -2 | tr '\n' ' ')"; done
cd /tmp
D=$O/trace_1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --parity off --no-prefetch > $O/bench_trace_1stream.json 2> $O/trace_1.err
T=$(find $D -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_breakdown.py $T 28 > $O/step_breakdown_1stream_noprefetch_chain.json 2>> $O/trace_1.err
rm -rf $D
python - <<PY
import json
d=json.load(open("$O/step_breakdown_1stream_noprefetch_chain.json"))
print("step", d["avg_step_us"], "kern", d["avg_kernel_time_per_step_us"], "per layer", d["us_per_layer"])
for k,v in list(d["kernels"].items())[:9]: print("   ", k[:70], v)
PY
