#!/bin/bash
# Runs on the MI355X box (via gpurun): rocprofv3 kernel stats of the bench + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the
# dominant kernel.  Only the small summary CSVs are kept under gpurun_out/ (the kernel traces are tens of MB).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r05}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off --live2fps off --more-configs off > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name '*kernel_trace.csv' -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o gemv -- python $R/tools/pmc_target.py > $OUT/pmc_$C.log 2>&1
  find $OUT/pmc_$C -name '*kernel_trace.csv' -delete
done
# MFMA utilisation of the 8-wave GEMM (LLM prefill gate/up, M = 3088 and 386): raw SQ counters + the derived metric (gfx94x formula)
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "MfmaUtil"; do
  T=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_gemm_$T -o gemm -- python $R/tools/pmc_target.py --gemm > $OUT/pmc_gemm_$T.log 2>&1 || true
  find $OUT/pmc_gemm_$T -name '*kernel_trace.csv' -size +2M -delete
done
grep -c . $OUT/counters_available.txt > /dev/null 2>&1 && grep -i "mfma" $OUT/counters_available.txt | head -40 > $OUT/counters_mfma.txt; rm -f $OUT/counters_available.txt
python $R/tools/summarize_pmc.py $OUT > $OUT/roofline_traffic.json 2> $OUT/summarize.err
cat $OUT/roofline_traffic.json
ls -R $OUT | head -40
