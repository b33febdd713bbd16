#!/bin/bash
# Copies the small summaries of a GPU call (gpurun_out/ is scratch) into the tracked profiles/<TAG>/ directory.
#   tools/collect_profiles.sh r02 [bench-log-dir]      (bench-log-dir default gpurun_out/<TAG>)
set -u
TAG=${1:-r05}
SRC=gpurun_out/profiles_$TAG
LOGS=${2:-gpurun_out/$TAG}
DST=profiles/$TAG
mkdir -p $DST
cp $SRC/stats/bench_kernel_stats.csv $DST/bench_kernel_stats.csv
cp $SRC/bench_under_rocprof.json $DST/bench_under_rocprof.json
cp $SRC/pmc_FETCH_SIZE/gemv_counter_collection.csv $DST/pmc_FETCH_SIZE_gemv_gate_up.csv
cp $SRC/pmc_WRITE_SIZE/gemv_counter_collection.csv $DST/pmc_WRITE_SIZE_gemv_gate_up.csv
cp $SRC/pmc_gemm_MfmaUtil/gemm_counter_collection.csv $DST/pmc_MfmaUtil_gemm_gate_up.csv
cp $SRC/pmc_gemm_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES_GRBM_GUI_ACTIVE/gemm_counter_collection.csv $DST/pmc_SQ_MFMA_BUSY_gemm_gate_up.csv
cp $SRC/pmc_gemm_SQ_INSTS_VALU_MFMA_MOPS_BF16_SQ_WAVE_CYCLES_SQ_ACTIVE_INST_ANY/gemm_counter_collection.csv $DST/pmc_SQ_MFMA_MOPS_gemm_gate_up.csv
cp $SRC/roofline_traffic.json $DST/roofline_traffic.json
cp $SRC/roofline_traffic.json profiles/roofline_traffic.json
for f in $LOGS/bench_*.log; do
  n=$(basename $f .log)
  grep '^{' $f | tail -n 1 > $DST/${n}_n1.json
done
for f in $LOGS/*.jsonl; do [ -f "$f" ] && cp $f $DST/; done
[ -f gpurun_out/parity_report.json ] && cp gpurun_out/parity_report.json $DST/parity_report.json
[ -f $LOGS/test_full.log ] && tail -n 12 $LOGS/test_full.log > $DST/pytest_gpu_tail.txt
[ -f $LOGS/smoke.log ] && tail -n 2 $LOGS/smoke.log > $DST/smoke_tail.txt
ls -la $DST
