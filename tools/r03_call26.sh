#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c26
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f gpurun_out/parity_report.json
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
( timeout 185 python -m pytest tests/test_gpu_layer_parity.py::test_every_layer_at_livecc_7b_shapes_matches_hf_on_the_oracles_input -m gpu -q -s --timeout 180 ) > $O/layer7b.log 2>&1
grep -E "passed|failed|first-token" $O/layer7b.log | cut -c1-300
cp gpurun_out/parity_report.json $O/parity_report_layer7b.json 2>/dev/null
