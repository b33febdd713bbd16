#!/bin/bash
# Full validation on an MI355X box (through gpurun): every -m gpu test, smoke(), the benchmark lines kept under profiles/<round>, rocprofv3 profiles
#   tools/validate_on_gpu.sh [TAG]      (TAG default r05; output under gpurun_out/<TAG>, collected by tools/collect_profiles.sh <TAG>)
set -x
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
# serially, exactly as the driver runs the tier (its limit is 20 minutes; -n 2 makes the HF CPU runs of the slow tests contend for the
# host cores and triples their time: 22 min in round 3's first validation against ~12 min serial)
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/test_full.log 2>&1
echo "tests rc=$?" >> $O/test_full.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1500 python bench.py ) > $O/bench_default.log 2>&1
B="timeout 500 python bench.py --cpu-baseline off --parity off --live2fps off --more-configs off"
( $B --steps 2 --warmup 1 --no-prefetch --share8 off ) > $O/bench_noprefetch.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8streams.log 2>&1
( $B --steps 1 --warmup 0 --streams-per-gpu 32 ) > $O/bench_32streams.log 2>&1
( $B --steps 1 --warmup 0 --workload oneshot480 ) > $O/bench_oneshot480.log 2>&1
( $B --steps 2 --warmup 1 --config qwen2vl-2b ) > $O/bench_2b.log 2>&1
( $B --steps 1 --warmup 1 --weights fp8 --share8 off ) > $O/bench_7b_fp8.log 2>&1
bash tools/run_profiles.sh $TAG > $O/run_profiles.log 2>&1
# round 5: the 8-wave GEMMs at the engine's shapes (hipEvents) and the vision tower alone, on the final tree
timeout 300 python tools/r5_bench_gemm.py final > $O/gemm_shapes_final.jsonl 2>$O/gemm_shapes_final.err
timeout 300 python tools/r5_tower.py final > $O/tower_final.jsonl 2>$O/tower_final.err
bash tools/gpu_call.sh trace 8 > $O/trace8.log 2>&1; cp gpurun_out/trace/step_breakdown_8streams_noprefetch.json $O/ 2>/dev/null
bash tools/gpu_call.sh trace 1 > $O/trace1.log 2>&1; cp gpurun_out/trace/step_breakdown_1streams_noprefetch.json $O/ 2>/dev/null
tail -n 40 $O/test_full.log; tail -n 2 $O/smoke.log
for f in bench_default bench_noprefetch bench_8streams bench_32streams bench_oneshot480 bench_2b bench_7b_fp8; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"us_per_layer": [0-9.]*' $O/$f.log | tr '\n' ' ')"; done
grep -o '"parity": {.*' $O/bench_default.log | cut -c1-1500
tail -n 12 $O/run_profiles.log
