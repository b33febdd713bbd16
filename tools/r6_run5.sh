# round 6, GPU call 5: the driver's default bench line + the rocprofv3 kernel stats of a short run of the same command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6e; mkdir -p $O
( time python bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.err ) 2> $O/bench_time.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off --share8 off --live2fps off --more-configs off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
cp $O/prof/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
python tools/trace_breakdown.py $O/prof > $O/step_breakdown_1streams.json 2>$O/breakdown.err
rm -rf $O/prof
tail -c 1500 $O/bench_default_n1.json; cat $O/bench_time.txt; head -12 $O/bench_kernel_stats.csv
