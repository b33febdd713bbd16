#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c8
mkdir -p $O
export TMPDIR=/tmp
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
for k,v in list(d["vit"]["kernels"].items())[:12]: print("   ", k[:80], v)
for k,v in list(d["prefill"]["kernels"].items())[:8]: print("   P ", k[:80], v)
PY
done
