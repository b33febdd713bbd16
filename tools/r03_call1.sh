#!/bin/bash
# Round 3, GPU call 1: evidence first -- kernel traces at 8 / 4 / 1 streams, barrier-xcd microbench, SQ counters of the prefill / ViT
# attention -- then the new parity tests (per-layer at tiny / small / LiveCC-7B, fp32 error ratio, decisive-weights token identity).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
nproc >> $O/device.txt; free -g | head -2 >> $O/device.txt
# 1. barrier microbench
timeout 120 python $R/tools/bench_barrier.py 2>/dev/null | grep '^{' > $O/grid_barrier_microbench.jsonl
# 2. kernel traces -> per-step / per-prefill breakdown (the CSV traces are tens of MB: reduced on the box)
for S in 8 4 1; do
  D=$O/trace_$S
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --parity off --no-prefetch --streams-per-gpu $S > $O/bench_trace_${S}streams.json 2> $O/trace_$S.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_breakdown.py $T 28 > $O/step_breakdown_${S}streams_noprefetch.json 2>> $O/trace_$S.err
  rm -rf $D
done
# 3. SQ counters of the LLM prefill attention (8 streams x 386 rows vs 6k keys) -- one pass per counter group
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_attn_$i -o attn -- python $R/tools/pmc_attn.py 4 16 > $O/pmc_attn_$i.log 2>&1
  find $O/pmc_attn_$i -name '*kernel_trace.csv' -delete
done
python - <<PY > $O/pmc_attn_summary.json 2>$O/pmc_attn_summary.err
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_attn_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "attn_" in kn:
            res[kn.split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
PY
# 4. parity tests (host-heavy: four worker processes x 32 threads)
cd $R
export OMP_NUM_THREADS=32
( time timeout 1500 python -m pytest tests/test_gpu_layer_parity.py tests/test_gpu_golden.py "tests/test_gpu_baseline_configs.py::test_livecc_7b_turns_match_hf_cpu_path_on_identical_weights" "tests/test_gpu_baseline_configs.py::test_greedy_tokens_are_exact_on_decisive_weights" -m gpu -q -n 4 --timeout 1400 ) > $O/parity_tests.log 2>&1
python - <<PY
import glob, json
m = {}
for f in sorted(glob.glob("gpurun_out/parity_report*.json")):
    m.update(json.load(open(f)))
json.dump(m, open("$O/parity_report.json", "w"), indent=1, sort_keys=True)
PY
ls gpurun_out > $O/ls.txt
tail -n 15 $O/parity_tests.log
cat $O/grid_barrier_microbench.jsonl
