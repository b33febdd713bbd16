#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c9
mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or attention or vit" --timeout 500 ) > $O/attn_tests.log 2>&1
tail -n 4 $O/attn_tests.log
( LCC_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_golden.py tests/test_gpu_layer_parity.py -m gpu -q -x --timeout 800 ) > $O/e2e_tests.log 2>&1
tail -n 4 $O/e2e_tests.log
cd /tmp
for S in 8 1; do
  D=$O/trace_$S
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --parity off --no-prefetch --streams-per-gpu $S > $O/bench_$S.json 2> $O/trace_$S.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_breakdown.py $T 28 > $O/step_breakdown_${S}streams_noprefetch_r03.json 2>> $O/trace_$S.err
  rm -rf $D
  python - <<PY
import json
d=json.load(open("$O/step_breakdown_${S}streams_noprefetch_r03.json"))
print("== $S streams: decode step", d["avg_step_us"], "prefill call", d["prefill"]["avg_call_us"], "vit call", d["vit"]["avg_call_us"], "calls", d["vit"]["vit_calls"])
for k,v in list(d["vit"]["kernels"].items())[:7]: print("   ", k[:80], v)
PY
done
cd $R
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_1s.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s.log 2>&1
( $B --steps 1 --warmup 1 --workload oneshot480 ) > $O/bench_oneshot480.log 2>&1
for f in bench_1s bench_8s bench_oneshot480; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"frames_per_s": [0-9.]*' $O/$f.log | head -1)"; done
