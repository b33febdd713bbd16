#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
( time timeout 800 python -m pytest tests/test_gpu_decode_v2.py tests/test_gpu_server.py -m gpu -q --timeout 600 ) > gpurun_out/test_j.log 2>&1
echo "tests rc=$?" >> gpurun_out/test_j.log
tail -n 30 gpurun_out/test_j.log | cut -c1-250
