#!/bin/bash
# round-2 GPU call I: CU-masked prefetch stream sweep
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
B="timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline off"
for n in 48 64 96 128 176; do ( LCC_VIT_CUS=$n $B ) > gpurun_out/i_cu$n.log 2>&1; done
( LCC_VIT_CUS=96 $B --streams-per-gpu 8 --steps 1 ) > gpurun_out/i_8s_cu96.log 2>&1
( LCC_VIT_CUS=160 $B --streams-per-gpu 8 --steps 1 ) > gpurun_out/i_8s_cu160.log 2>&1
( LCC_VIT_CUS=96 timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k prefetch ) > gpurun_out/test_i.log 2>&1
tail -n 3 gpurun_out/test_i.log
for f in i_cu48 i_cu64 i_cu96 i_cu128 i_cu176 i_8s_cu96 i_8s_cu160; do echo "== $f $(grep -o '"value": [0-9.]*' gpurun_out/$f.log) $(grep -o '"us_per_layer": [0-9.]*' gpurun_out/$f.log) $(tail -c 300 gpurun_out/$f.log | grep -o 'Error.*' | head -1)"; done
