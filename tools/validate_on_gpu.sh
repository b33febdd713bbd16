#!/bin/bash
# Full validation on an MI355X box (through gpurun): every -m gpu test, smoke(), the benchmark lines kept under profiles/<round>, rocprofv3 profiles
set -x
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
O=gpurun_out/r02
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 1500 ) > $O/test_full.log 2>&1
echo "tests rc=$?" >> $O/test_full.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1200 python bench.py --steps 3 --warmup 1 --parity full --cpu-budget 300 ) > $O/bench_default_parity_full.log 2>&1
B="timeout 400 python bench.py --cpu-baseline off"
( $B --steps 2 --warmup 1 --no-prefetch ) > $O/bench_noprefetch.log 2>&1
( $B --steps 2 --warmup 1 --no-prefetch --decode-path 0 ) > $O/bench_round1_path.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 8 ) > $O/bench_8streams.log 2>&1
( $B --steps 1 --warmup 1 --streams-per-gpu 2 ) > $O/bench_2streams.log 2>&1
( $B --steps 1 --warmup 0 --streams-per-gpu 32 ) > $O/bench_32streams.log 2>&1
( $B --steps 2 --warmup 1 --no-prefetch --gemm-variant 4 ) > $O/bench_noprefetch_gemm128.log 2>&1
python tools/bench_tall.py 2>/dev/null | grep '^{' > $O/gemm_tall_microbench.jsonl
( $B --steps 1 --warmup 0 --workload long480 ) > $O/bench_long480.log 2>&1
( $B --steps 2 --warmup 1 --config qwen2vl-2b ) > $O/bench_2b.log 2>&1
( $B --steps 1 --warmup 1 --weights fp8 ) > $O/bench_7b_fp8.log 2>&1
( $B --steps 1 --warmup 1 --config qwen2vl-72b --weights fp8 --frames 16 ) > $O/bench_72b_fp8.log 2>&1
bash tools/run_profiles.sh r02 > $O/run_profiles.log 2>&1
tail -n 5 $O/test_full.log; tail -n 2 $O/smoke.log
for f in bench_default_parity_full bench_noprefetch bench_round1_path bench_noprefetch_gemm128 bench_8streams bench_2streams bench_32streams bench_long480 bench_2b bench_7b_fp8 bench_72b_fp8; do echo "== $f $(grep -o '"value": [0-9.]*' $O/$f.log | head -1) $(grep -o '"us_per_layer": [0-9.]*' $O/$f.log)"; done
grep -o '"parity": {[^}]*}' $O/bench_default_parity_full.log
