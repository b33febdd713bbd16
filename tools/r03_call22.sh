#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c22
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 700 python -m pytest tests/test_gpu_layer_parity.py::test_every_layer_at_livecc_7b_shapes_matches_hf_on_the_oracles_input "tests/test_gpu_baseline_configs.py::test_greedy_tokens_are_exact_on_decisive_weights" -m gpu -q -s --timeout 650 --durations=5 ) > $O/slow_block.log 2>&1
grep -E "first-token|passed|failed|skipped|s call|real" $O/slow_block.log | cut -c1-900
( LCC_GEMM_SCHED=6 LCC_TALL_SCHED=2 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 250 -k "gemm or linear or tall" ) > $O/gemm_tests.log 2>&1
tail -n 3 $O/gemm_tests.log
for S in 1 6; do LCC_GEMM_SCHED=$S timeout 100 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed "s/^/gemm_sched$S /" | tee -a $O/gemm_sched.txt; done
