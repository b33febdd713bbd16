#!/bin/bash
# round-2 GPU call G: vision-tower prefetch on a side stream: bit-identity test, A/B bench (1, 2, 8 streams)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --timeout 600 -k "prefetch or tiny" ) > gpurun_out/test_g.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_g.log
B="timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline off"
( $B ) > gpurun_out/g_pf.log 2>&1
( $B --no-prefetch ) > gpurun_out/g_nopf.log 2>&1
( $B --streams-per-gpu 8 --steps 1 ) > gpurun_out/g_8s_pf.log 2>&1
( $B --streams-per-gpu 8 --steps 1 --no-prefetch ) > gpurun_out/g_8s_nopf.log 2>&1
( $B --streams-per-gpu 2 ) > gpurun_out/g_2s_pf.log 2>&1
tail -n 5 gpurun_out/test_g.log
for f in g_pf g_nopf g_8s_pf g_8s_nopf g_2s_pf; do echo "== $f $(grep -o '"value": [0-9.]*' gpurun_out/$f.log) $(grep -o '"us_per_layer": [0-9.]*' gpurun_out/$f.log) $(grep -o '"frames_per_s": [0-9.]*' gpurun_out/$f.log | head -1)"; done
