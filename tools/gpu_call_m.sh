#!/bin/bash
# round-2 GPU call M: tall GEMM kernel (M = 386 prefill) -- parity, bench, MfmaUtil
set -x
mkdir -p gpurun_out/r02m
export TMPDIR=/tmp
O=gpurun_out/r02m
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 900 -k "tall or auto_choice or gemm_tiled_swiglu" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
( timeout 600 python -m pytest tests/test_gpu_facade.py tests/test_gpu_e2e.py -m gpu -q -x --timeout 600 ) > $O/tests2.log 2>&1
echo "tests2 rc=$?" >> $O/tests2.log
B="timeout 400 python bench.py --cpu-baseline off --parity off"
( $B --steps 3 --warmup 1 ) > $O/bench_default.log 2>&1
( $B --steps 3 --warmup 1 --no-prefetch ) > $O/bench_noprefetch.log 2>&1
( $B --steps 2 --warmup 1 --no-prefetch --gemm-variant 4 ) > $O/bench_noprefetch_big128.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off --no-prefetch > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/stats.err
find $GRAFT_REPO_ROOT/$O/stats -name '*kernel_trace.csv' -delete
timeout 200 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_MfmaUtil -o gemm -- python $GRAFT_REPO_ROOT/tools/pmc_target.py --gemm > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 6 $O/tests.log; tail -n 4 $O/tests2.log
for f in bench_default bench_noprefetch bench_noprefetch_big128; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"us_per_layer": [0-9.]*' $O/$f.log)"; tail -n 1 $O/$f.log | cut -c1-200; done
grep -i "gemm_tall\|gemm_big" $O/stats/bench_kernel_stats.csv | cut -c1-40,150-260
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r02m/pmc_MfmaUtil/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_' in r.get('Kernel_Name','') and r.get('Counter_Name')=='MfmaUtil':
            print(r['Kernel_Name'][:60], r['Counter_Value'])
PY
