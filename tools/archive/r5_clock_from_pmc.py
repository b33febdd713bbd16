#!/usr/bin/env python
"""Effective shader clock of each gemm_vh_kernel dispatch = GRBM_GUI_ACTIVE cycles / dispatch duration, from a rocprofv3 --pmc pass over
tools/r5_power_probe.py (dispatch order: per M, 3 repetitions of [random, zeros, random x, zero W] x 24 launches)."""
import csv
import glob
import json
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_vh_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "GRBM_GUI_ACTIVE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
rows.sort()
# fold the XCD instances of one dispatch (rocprofv3 writes one row per dispatch with the reduced value, or one per instance)
by = {}
for d, v, us in rows:
    by.setdefault(d, []).append((v, us))
disp = [(d, max(v for v, _ in xs), sum(v for v, _ in xs), xs[0][1]) for d, xs in sorted(by.items())]
names = ("random", "zeros", "random_x_zero_w")
per_phase = 24          # 4 warm-up + 20 timed launches per (rep, operand kind)
out = {}
for i, (d, vmax, vsum, us) in enumerate(disp):
    ph = i // per_phase
    M = 3088 if ph < 9 else 9048
    kind = names[ph % 3]
    if (i % per_phase) < 4:
        continue
    out.setdefault((M, kind), []).append((vmax, vsum, us))
for (M, kind), xs in sorted(out.items()):
    n = len(xs)
    us = sum(x[2] for x in xs) / n
    cyc_max = sum(x[0] for x in xs) / n
    cyc_sum = sum(x[1] for x in xs) / n
    print(json.dumps(dict(M=M, operands=kind, dispatches=n, us_under_pmc=round(us, 1), grbm_gui_active_mean=round(cyc_max), rows_per_dispatch=len(by[disp[0][0]]),
                          mhz_if_counter_is_per_xcd=round(cyc_max / us, 1), mhz_if_counter_is_sum_of_8=round(cyc_sum / 8 / us, 1))))
