#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c7
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for V in 0 1; do
  D=$O/trace_$V
  LCC_RESID_UNR3=$V timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --parity off --no-prefetch --decode-chain 0 > $O/bench_unr3_$V.json 2> $O/trace_$V.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_breakdown.py $T 28 > $O/step_unr3_$V.json 2>> $O/trace_$V.err
  rm -rf $D
  python - <<PY
import json
d=json.load(open("$O/step_unr3_$V.json"))
print("UNR3=$V step", d["avg_step_us"], "per layer", d["us_per_layer"])
for k,v in list(d["kernels"].items())[:5]: print("   ", k[:70], v)
PY
done
