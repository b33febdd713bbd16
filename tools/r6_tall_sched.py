#!/usr/bin/env python
"""Round-6 A/B target: the one-chunk (M = 386) LLM GEMMs timed back to back with hipEvents over rotating weights -- gate/up on the tall kernel under
LCC_TALL_SCHED = 0 / 1 / 2 (the pinned schedules were last measured BEFORE the LDS-DMA went into inline asm: round 5 changed what hipcc's waitcnt
pass sees), and the three split-K projections.  argv[1] = tag; one JSON line per GEMM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
H, I, QKV, M = 3584, 18944, 4608, 386
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
xi = torch.randn(M, I, device=dev).to(torch.bfloat16)


def timeit(fn, ws, iters=60):
    for i in range(4):
        fn(ws[i % len(ws)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(ws[i % len(ws)])
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def packed(n, k, c):
    return [ops.pack_weight((torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(c)]


for name, N, K, a, c in (("gate_up", 2 * I, H, x, 3), ("down", H, I, xi, 4), ("qkv", QKV, H, x, 8), ("o", H, H, x, 8)):
    ws = packed(N, K, c)
    if name == "gate_up":
        us = timeit(lambda w: ops.linear(a, w, None, ops.EPI_SWIGLU, packed_shape=(N, K)), ws)
        splits = 1
    else:
        splits = ops.gemm_plan(M, N, K, ops.EPI_NONE)[1]
        us = timeit(lambda w: ops.linear_partial(a, w, splits, packed_shape=(N, K)), ws)
    print(json.dumps(dict(tag=tag, gemm=name, M=M, N=N, K=K, splits=splits, us=round(us, 1), pflops=round(2.0 * M * N * K / us / 1e9, 3))), flush=True)
    del ws
