#!/usr/bin/env python
"""What does a concurrent MFMA-bound GEMM stream cost an HBM-bound weight-streaming GEMV stream (and vice versa)?  The measurement behind
DESIGN section 9 "overlapping the tower with decode": stream A = 300 decode gate/up GEMVs (M = 1, 271.6 MB each, rotating weights),
stream B = back-to-back vision-tower fc1 GEMMs (11648 x 5120 x 1280, the 8-wave kernel), each alone and both together (B on a
lowest-priority stream as the prefetched tower is), optionally with B under a grid cap.  One JSON line per arrangement."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, I = 3584, 18944
wv = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(3)]
xv = torch.randn(1, H, device=dev).to(torch.bfloat16)
wm = ops.pack_weight((torch.randn(5120, 1280, device=dev) * 0.02).to(torch.bfloat16))
bm = (torch.randn(5120, device=dev) * 0.1).to(torch.bfloat16)
xm = torch.randn(11648, 1280, device=dev).to(torch.bfloat16)
lo = max(torch.cuda.Stream.priority_range())
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=lo)
NV, NM = 300, 60


def gemvs():
    for i in range(NV):
        ops.linear(xv, wv[i % 3], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))


def gemms():
    for _ in range(NM):
        ops.linear(xm, wm, bm, ops.EPI_QUICK_GELU, packed_shape=(5120, 1280))


def timed(run_a, run_b):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if run_b:
        with torch.cuda.stream(sb):
            eb0.record()
            gemms()
            eb1.record()
    if run_a:
        with torch.cuda.stream(sa):
            ea0.record()
            gemvs()
            ea1.record()
    torch.cuda.synchronize()
    return (ea0.elapsed_time(ea1) * 1e3 / NV if run_a else None, eb0.elapsed_time(eb1) * 1e3 / NM if run_b else None)


with torch.cuda.stream(sa):
    gemvs()
with torch.cuda.stream(sb):
    gemms()
torch.cuda.synchronize()
for name, a, b in (("gemv alone", True, False), ("gemm alone", False, True), ("both", True, True), ("both again", True, True)):
    ua, ub = timed(a, b)
    print(json.dumps(dict(arrangement=name, gemv_us=None if ua is None else round(ua, 2), gemm_us=None if ub is None else round(ub, 1),
                          gemv_TBps=None if ua is None else round(271.6e6 / ua / 1e6, 2), gemm_pflops=None if ub is None else round(2 * 11648 * 5120 * 1280 / ub / 1e9, 3))), flush=True)
