#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c20
mkdir -p $O
cd $R
for D in 0 2 3 4 5; do LCC_GEMM_DIAG=$D timeout 200 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | tee -a $O/gemm_diag.jsonl; done
