#!/bin/bash
# round-2 GPU call H: batched vision-tower prefetch A/B (1, 4, 8 streams)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --timeout 600 -k "prefetch" ) > gpurun_out/test_h.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_h.log
B="timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline off"
( $B ) > gpurun_out/h_pf.log 2>&1
( $B --streams-per-gpu 8 --steps 1 ) > gpurun_out/h_8s_pf.log 2>&1
( $B --streams-per-gpu 8 --steps 1 --no-prefetch ) > gpurun_out/h_8s_nopf.log 2>&1
( $B --streams-per-gpu 4 --steps 1 ) > gpurun_out/h_4s_pf.log 2>&1
( $B --streams-per-gpu 4 --steps 1 --no-prefetch ) > gpurun_out/h_4s_nopf.log 2>&1
tail -n 5 gpurun_out/test_h.log
for f in h_pf h_8s_pf h_8s_nopf h_4s_pf h_4s_nopf; do echo "== $f $(grep -o '"value": [0-9.]*' gpurun_out/$f.log) $(grep -o '"us_per_layer": [0-9.]*' gpurun_out/$f.log)"; done
