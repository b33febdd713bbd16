#!/usr/bin/env python
"""Per-phase breakdown of a rocprofv3 kernel trace of bench.py (`rocprofv3 --kernel-trace --output-format csv`): what one decode step
and one decoder layer cost, kernel by kernel, and how much of the wall time between the first and the last kernel of a step is not
covered by any kernel (launch gaps).  A decode step starts at every `decode_step_begin_kernel` (pipeline v2) or `seen_set_kernel`
(round-1 launch sequence) and ends at the next `sample_*` kernel on the same queue.

    python tools/trace_breakdown.py <..._kernel_trace.csv> [n_layers]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)               # drop the argument list
    return name.replace("lcc::", "")


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                continue
            name = short(r["Kernel_Name"])
            if name.startswith("gemm_") and "Grid_Size_X" in r:      # the same GEMM template serves several shapes: tell them apart by grid
                wx, wy = max(1, int(r.get("Workgroup_Size_X", 1))), max(1, int(r.get("Workgroup_Size_Y", 1)))
                name += f" grid=({int(r['Grid_Size_X']) // wx},{int(r.get('Grid_Size_Y', 1)) // wy},{r.get('Grid_Size_Z', 1)})"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0")))
    rows.sort()
    return rows


def breakdown(rows, n_layers=28):
    steps, cur = [], None
    for s, e, name, q in rows:
        if name.startswith("decode_step_begin_kernel") or name.startswith("seen_set_kernel"):
            cur = dict(queue=q, kernels=[])
        if cur is not None and q == cur["queue"]:
            cur["kernels"].append((s, e, name))
            if name.startswith("sample_") and not name.startswith("sample_partial"):
                # a prefill call also marks ids (seen_set) and samples: it is recognised by its GEMM / prefill-attention kernels
                if not any(k[2].startswith(("gemm_", "attn_shared", "attn_prefill", "attn_vit")) for k in cur["kernels"]):
                    steps.append(cur)
                cur = None
    prefill = prefill_breakdown(rows)
    per_kernel = defaultdict(lambda: [0, 0.0])
    wall = busy = 0.0
    for st in steps:
        ks = st["kernels"]
        wall += (ks[-1][1] - ks[0][0]) / 1e3
        for s, e, name in ks:
            per_kernel[name][0] += 1
            per_kernel[name][1] += (e - s) / 1e3
            busy += (e - s) / 1e3
    n = max(1, len(steps))
    out = dict(decode_steps=len(steps), avg_step_us=round(wall / n, 2), avg_kernel_time_per_step_us=round(busy / n, 2),
               avg_gap_per_step_us=round((wall - busy) / n, 2), us_per_layer=round(wall / n / max(1, n_layers), 2), kernels={})
    for name, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        out["kernels"][name] = dict(calls_per_step=round(c / n, 2), avg_us=round(t / c, 2), us_per_step=round(t / n, 2))
    out["prefill"] = prefill
    out["vit"] = vit_breakdown(rows)
    return out


def vit_breakdown(rows):
    """Vision-tower calls on the main stream (bench --no-prefetch): from `patchify_norm_kernel` (or the fp32 -> bf16 cast of
    pixel_values) to the last kernel before the next `seen_set_kernel` (= the start of the LLM prefill that consumes the embeddings)."""
    calls, cur = [], None
    for s, e, name, q in rows:
        if cur is None and name.startswith(("patchify_norm_kernel", "cast_f32_bf16")):
            cur = dict(queue=q, kernels=[])
        if cur is not None and q == cur["queue"]:
            if name.startswith("seen_set_kernel"):
                calls.append(cur)
                cur = None
            else:
                cur["kernels"].append((s, e, name))
    per_kernel = defaultdict(lambda: [0, 0.0])
    wall = busy = 0.0
    for c in calls:
        ks = c["kernels"]
        wall += (ks[-1][1] - ks[0][0]) / 1e3
        for s, e, name in ks:
            per_kernel[name][0] += 1
            per_kernel[name][1] += (e - s) / 1e3
            busy += (e - s) / 1e3
    n = max(1, len(calls))
    out = dict(vit_calls=len(calls), avg_call_us=round(wall / n, 1), avg_kernel_time_per_call_us=round(busy / n, 1), kernels={})
    for name, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:16]:
        out["kernels"][name] = dict(calls_per_vit=round(c / n, 2), avg_us=round(t / c, 2), us_per_vit=round(t / n, 1))
    return out


def prefill_breakdown(rows):
    """LLM prefill calls (embed_gather_kernel ... first sample_* on the same queue, recognised by their GEMM / prefill-attention
    kernels) and vision-tower calls (patchify / cast ... the last GEMM before the next non-ViT kernel are not delimited by a marker, so
    the ViT is reported as the per-kernel totals of everything that is neither inside a decode step nor inside an LLM prefill)."""
    calls, cur = [], None
    for s, e, name, q in rows:
        if name.startswith("embed_gather_kernel") and cur is None:
            cur = dict(queue=q, kernels=[])
        if cur is not None and q == cur["queue"]:
            cur["kernels"].append((s, e, name))
            if name.startswith("sample_") and not name.startswith("sample_partial"):
                if any(k[2].startswith(("gemm_", "attn_shared", "attn_prefill")) for k in cur["kernels"]):
                    calls.append(cur)
                cur = None
    per_kernel = defaultdict(lambda: [0, 0.0])
    wall = busy = 0.0
    for c in calls:
        ks = c["kernels"]
        wall += (ks[-1][1] - ks[0][0]) / 1e3
        for s, e, name in ks:
            per_kernel[name][0] += 1
            per_kernel[name][1] += (e - s) / 1e3
            busy += (e - s) / 1e3
    n = max(1, len(calls))
    out = dict(prefill_calls=len(calls), avg_call_us=round(wall / n, 1), avg_kernel_time_per_call_us=round(busy / n, 1), kernels={})
    for name, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:24]:
        out["kernels"][name] = dict(calls_per_prefill=round(c / n, 2), avg_us=round(t / c, 2), us_per_prefill=round(t / n, 1))
    return out


if __name__ == "__main__":
    print(json.dumps(breakdown(load(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 28), indent=1))
