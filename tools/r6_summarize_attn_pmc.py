#!/usr/bin/env python
"""Reduce a `rocprofv3 --pmc MfmaUtil --kernel-trace` pass over tools/bench_attn.py --only32 and tools/r5_tower.py: mean MfmaUtil and duration
of the attention kernels' dispatches, grouped by kernel and grid size (the grid tells the cases apart).  argv[1] = output directory of the pass."""
import collections
import csv
import glob
import json
import os
import sys

rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if r.get("Counter_Name") != "MfmaUtil" or not ("attn_gqa32" in kn or "attn_vit32" in kn):
            continue
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0)
        rows[(kn.split("(")[0].replace("void lcc::", ""), grid // max(wg, 1), wg)].append(
            (float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
for (kn, blocks, wg), v in sorted(rows.items()):
    v = v[len(v) // 4:]          # the first quarter of a group's dispatches = warm-up
    print(json.dumps(dict(kernel=kn, workgroups=blocks, threads=wg, dispatches=len(v), mfma_util_pct=round(sum(x[0] for x in v) / len(v), 2),
                          us_under_pmc=round(sum(x[1] for x in v) / len(v), 1))))
