#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c4
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/pmc_attn.sh $O/pmc_attn32 1 32 3 > $O/pmc_attn32.log 2>&1
tail -n 25 $O/pmc_attn32.log
cd /tmp
D=$O/trace_1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --parity off --no-prefetch > $O/bench_trace_1stream.json 2> $O/trace_1.err
T=$(find $D -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_breakdown.py $T 28 > $O/step_breakdown_1stream_noprefetch_v3.json 2>> $O/trace_1.err
rm -rf $D
python - <<PY
import json
d=json.load(open("$O/step_breakdown_1stream_noprefetch_v3.json"))
p=d["prefill"]; print("prefill avg call us", p["avg_call_us"])
for k,v in list(p["kernels"].items())[:14]: print("   ", k[:90], v)
PY
cd $R
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 1 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8s_fa1.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 --fused-attn 3 ) > $O/bench_8s_fa3.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 --fused-attn 0 ) > $O/bench_8s_fa0.log 2>&1
( $B --steps 1 --warmup 0 --streams-per-gpu 32 ) > $O/bench_32s.log 2>&1
for f in bench_8s_fa1 bench_8s_fa3 bench_8s_fa0 bench_32s; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"decode_step": {[^}]*}' $O/$f.log | head -1 | cut -c1-200)"; done
( time LCC_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -q -x --timeout 800 -n 2 ) > $O/test_all.log 2>&1
tail -n 6 $O/test_all.log
