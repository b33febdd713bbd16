set -x
mkdir -p gpurun_out/r02t
O=gpurun_out/r02t
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 1200 ) > $O/test_full.log 2>&1
echo "tests rc=$?" >> $O/test_full.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 400 python bench.py --cpu-baseline off --steps 3 --warmup 1 ) > $O/bench_default_nocpu.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-baseline off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/stats.err
find $GRAFT_REPO_ROOT/$O/stats -name '*kernel_trace.csv' -delete
cd $GRAFT_REPO_ROOT
tail -n 6 $O/test_full.log; tail -n 1 $O/smoke.log; tail -n 1 $O/bench_default_nocpu.log | cut -c1-300
