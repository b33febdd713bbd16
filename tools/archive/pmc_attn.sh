#!/bin/bash
# SQ counters of the LLM prefill attention (8 streams x 386 rows vs 6k keys), one rocprofv3 pass per counter group.
#   tools/pmc_attn.sh <out_dir> <nsplit> <tile_rows> <variant>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; NS=${2:-1}; TR=${3:-32}; V=${4:-3}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pass$i -o attn -- python $R/tools/pmc_attn.py $NS $TR $V > $O/pass$i.log 2>&1
  find $O/pass$i -name '*kernel_trace.csv' -delete
done
python - <<PY > $O/summary.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "attn_" in kn:
            res[kn.split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
PY
cat $O/summary.json
