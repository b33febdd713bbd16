#!/usr/bin/env python
"""Reduce the MfmaUtil pass of tools/r6_pmc_llm_gemms.py: per LLM GEMM (q|k|v, o, gate/up, down) at M = 3088 and M = 386 the mean MfmaUtil of its
dispatches (the first two of six dropped as warm-up), the kernel that served it, its duration under the counters, and the FLOP-weighted
utilisation of a decoder layer's four GEMMs per M.  Merged into profiles/roofline_traffic.json as `llm_gemm_mfma_util` together with the
sha256 of csrc/gemm.hip: bench.py reports the block in `configs2_share.roofline` and nulls it when gemm.hip has changed since the pass."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
manifest = json.load(open(os.path.join(d, "manifest.json")))
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "MfmaUtil" and "gemm_" in r.get("Kernel_Name", "") and "dequant" not in r.get("Kernel_Name", ""):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
rows.sort()
assert len(rows) == len(manifest), f"{len(rows)} GEMM dispatches in the counter CSV, {len(manifest)} in the manifest"
per = {}
for (did, kn, util, us), m in zip(rows, manifest):
    if m["rep"] < 2:
        continue
    e = per.setdefault(m["label"], dict(M=m["M"], N=m["N"], K=m["K"], flops=m["flops"], splits=m["splits"], tile_rows=m["tile_rows"],
                                        kernel=kn.split("(")[0][:80], util=[], us=[]))
    e["util"].append(util)
    e["us"].append(us)
out = {}
for label, e in per.items():
    out[label] = dict(mfma_util_pct=round(sum(e["util"]) / len(e["util"]), 2), duration_us_under_pmc=round(sum(e["us"]) / len(e["us"]), 1),
                      pflops_algorithmic=round(e["flops"] / (sum(e["us"]) / len(e["us"])) / 1e9, 3), kernel=e["kernel"], splits=e["splits"],
                      tile_rows=e["tile_rows"], flops=e["flops"], n=len(e["util"]))
agg = {}
for M in (3088, 386):
    ks = [k for k in out if k.startswith(f"M{M}_")]
    fl = sum(out[k]["flops"] for k in ks)
    agg[f"M{M}"] = dict(flop_weighted_mfma_util_pct=round(sum(out[k]["mfma_util_pct"] * out[k]["flops"] for k in ks) / fl, 2),
                        time_weighted_mfma_util_pct=round(sum(out[k]["mfma_util_pct"] * out[k]["duration_us_under_pmc"] for k in ks) /
                                                          sum(out[k]["duration_us_under_pmc"] for k in ks), 2),
                        gemm_us_per_layer_under_pmc=round(sum(out[k]["duration_us_under_pmc"] for k in ks), 1))
src = os.path.join(ROOT, "livecc_amd", "csrc", "gemm.hip")
blk = dict(per_gemm=out, per_layer=agg, gemm_source="livecc_amd/csrc/gemm.hip", gemm_source_sha16=hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           how="rocprofv3 --pmc MfmaUtil (gfx94x formula: sum SQ_VALU_MFMA_BUSY_CYCLES / (max GRBM_GUI_ACTIVE x SIMDs)) over tools/r6_pmc_llm_gemms.py: the four "
               "GEMMs of a LiveCC-7B decoder layer as the engine's prefill launches them, N(0,1) activations, 4 timed dispatches each; T2 of the "
               "north_star = flop_weighted_mfma_util_pct at M3088 (8 co-scheduled streams)")
tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
t = json.load(open(tp))
t["llm_gemm_mfma_util"] = blk
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps(blk, indent=1))
