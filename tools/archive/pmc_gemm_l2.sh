#!/bin/bash
# Where do the bytes of the 8-wave GEMM's
# LDS-DMA ring come from?  The round-3 timing split says the ring is THROUGHPUT-bound at ~12 TB/s chip-wide (profiles/r03/gemm_diag.jsonl);
# these passes ask whether that is the L2 (hit bandwidth, same-line contention), the fabric behind it (MALL / HBM misses), or the CU side
# (TCP / TA).  One rocprofv3 pass per counter group (never combined with trace domains), kernel = gemm_big_kernel<256,...> at M = 3088
# and the M = 386 kernels of tools/pmc_target.py --gemm.  (Counter names checked against /opt/rocm/share/rocprofiler-sdk/counter_defs.yaml
# for gfx950; a pass whose group cannot be scheduled together is reported and skipped.)
#   tools/pmc_gemm_l2.sh <out_dir>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-$R/gpurun_out/pmc_gemm_l2}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "TCC_|TCP_|TA_" | cut -c1-160 > $O/counters_tcc_tcp.txt
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_NORMAL_WRITEBACK_sum" \
         "TA_BUSY_avr TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
         "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pass$i -o gemm -- python $R/tools/pmc_target.py --gemm > $O/pass$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -n 1 $O/pass$i.log)"
  find $O/pass$i -name '*kernel_trace.csv' -delete
done
python - <<PY > $O/gemm_l2_counters.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "gemm_big_kernel" in kn or "gemm_tall_kernel" in kn:
            key = kn.split("(")[0].replace("void lcc::", "")[:60] + " grid=" + r["Grid_Size"]
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[key]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
for k, c in out.items():      # derived: L2 hit rate, bytes from beyond the L2 per launch (32-byte and 64-byte requests), L2 request rate
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    if c.get("TCC_EA0_RDREQ_sum") is not None:
        r32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        c["bytes_from_beyond_l2"] = r32 * 32 + (c["TCC_EA0_RDREQ_sum"] - r32) * 64
print(json.dumps(out, indent=1))
PY
cat $O/gemm_l2_counters.json
rm -rf $O/pass*/
