#!/usr/bin/env python
"""Reduce the rocprofv3 --pmc CSVs of tools/r6_gpu_calls.sh pmc_gemv; earlier rounds: tools/archive/run_profiles.sh to HBM bytes per launch of the gate/up GEMV.
Corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte
read requests as 64 bytes for wide (16 B/lane) coalesced streams -> doubled.  WRITE_SIZE is uncalibrated (reported raw)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(os.path.join(out, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dgemv_kernel<2" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
                vals.append(float(r["Counter_Value"]))
    vals = vals[2:] if len(vals) > 4 else vals          # drop warm-up launches
    res[c] = dict(n=len(vals), mean_raw=sum(vals) / len(vals) if vals else None)
fetch = res["FETCH_SIZE"]["mean_raw"]
write = res["WRITE_SIZE"]["mean_raw"]
alg = 2 * 18944 * 3584 * 2 + 3584 * 2 + 18944 * 2
import hashlib  # noqa: E402
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "livecc_amd", "csrc", "decode_v2.hip")
d = dict(kernel="dgemv_kernel<2,NORM,SWIGLU,4,1> ([RMSNorm] gate/up GEMV [SwiGLU]) M=1 N=37888 K=3584",
         kernel_source="livecc_amd/csrc/decode_v2.hip",      # bench.py nulls `roofline.traffic` when this file has changed since the PMC pass
         kernel_source_sha16=hashlib.sha256(open(_src, "rb").read()).hexdigest()[:16], algorithmic_bytes_per_launch=alg,
         fetch_size_kib_raw=fetch, write_size_kib_raw=write,
         gemv_gate_up_hbm_bytes_per_launch=(fetch * 1024 * 2 + (write or 0) * 1024) if fetch else None,
         note="FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024 (uncalibrated)", passes=res)
# MFMA-utilisation passes of the 8-wave GEMM (best effort: raw per-dispatch counters and the kernel duration of the same dispatch)
gemm = {}
for f in glob.glob(os.path.join(out, "pmc_gemm_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "gemm_big_kernel" not in kn and "gemm_tall_kernel" not in kn and "gemm_vh_kernel" not in kn:
            continue
        # round 5: the gate/up GEMM at M = 3088 runs on the variable-height tiles (gemm_vh_kernel: 12 x 148 tiles of 256 / 272 rows)
        key = ("M386_gemm_tall_kernel" if "gemm_tall_kernel" in kn else "M3088_gemm_vh_kernel" if "gemm_vh_kernel" in kn else
               "M3088_gemm_big_kernel_256" if "<256" in kn else "M386_gemm_big_kernel_128")
        if "gemm_vh_kernel<5, 1>" in kn:      # the small variable-height class (split-K slabs of one chunk): 54 tiles x 4 splits = q/k/v, 42 x 6 = o / down
            grid = int(r.get("Grid_Size", 0)) // 512
            dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
            key = "M386_qkv_splitk4_vh_small" if grid == 216 else ("M386_down_splitk6_vh_small" if dur > 35.0 else "M386_o_splitk6_vh_small")
            gemm.setdefault(key, {}).setdefault("duration_us_under_pmc", []).append(dur)
        if "<128, 5" in kn:      # split-K slabs at M = 386: grid (tiles, splits) x 512 threads tells q/k/v (72 x 3) from o / down (56 x 4);
            grid = int(r.get("Grid_Size", 0)) // 512      # o and down share a grid: told apart by the duration of the dispatch
            dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
            key = "M386_qkv_splitk3" if grid == 216 else ("M386_down_splitk4" if dur > 45.0 else "M386_o_splitk4")
            gemm.setdefault(key, {}).setdefault("duration_us_under_pmc", []).append(dur)
        gemm.setdefault(key, {}).setdefault(r.get("Counter_Name"), []).append(float(r["Counter_Value"]))
d["gemm_counters_mean_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in gemm.items()}
d["gemm_note"] = ("MfmaUtil = rocprofv3 derived metric reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum)/(reduce(GRBM_GUI_ACTIVE,max)*SIMD_NUM)*100 (gfx94x formula); "
                  "the raw SQ_* / GRBM_* values are per-dispatch means of the CSV rows as rocprofv3 writes them")
print(json.dumps(d, indent=1))
