#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03c16
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pass$i -o gemm -- python $R/tools/pmc_target.py --gemm > $O/pass$i.log 2>&1
  find $O/pass$i -name '*kernel_trace.csv' -delete
  tail -n 2 $O/pass$i.log
done
python - <<PY > $O/gemm_lds_counters.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "gemm_big_kernel" in kn or "gemm_tall_kernel" in kn:
            key = kn.split("(")[0].replace("void lcc::", "")[:60] + " grid=" + r["Grid_Size"]
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
PY
cat $O/gemm_lds_counters.json
rm -rf $O/pass*/
