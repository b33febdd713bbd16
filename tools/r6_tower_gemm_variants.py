#!/usr/bin/env python
"""Round-6 probe: the four GEMM shapes of ONE 2-frame tower chunk (M = 1456: q|k|v 1280 -> 3840, proj 1280 -> 1280 + residual, fc1 1280 -> 5120 +
QuickGELU, fc2 5120 -> 1280 + residual) under the automatic tile choice and with the 8-wave tile height forced (lcc_debug_set_gemm_variant 4 =
128 rows, 13 = 192, 3 = 256, 7 = never the 8-wave kernel), split-K as the engine picks it (nsplit 0 = the plan's own count).  One JSON line per
(shape, variant): microseconds per launch, PFLOP/s, the plan (tile rows, splits)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1456
shapes = (("qkv", 3840, 1280, ops.EPI_NONE), ("proj", 1280, 1280, ops.EPI_RESIDUAL), ("fc1", 5120, 1280, ops.EPI_QUICK_GELU), ("fc2", 1280, 5120, ops.EPI_RESIDUAL))
for name, N, K, epi in shapes:
    torch.manual_seed(7)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(4)]
    bias = (torch.randn(N, device=dev) * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_RESIDUAL else None
    for variant in (2, 4, 13, 3, 7, 2):
        ops.set_gemm_variant(variant)
        try:
            plan = ops.gemm_plan(M, N, K, epi, 0)

            def call(i):
                return ops.linear(x, ws[i % 4], bias, epi, residual=res, packed_shape=(N, K))
            for i in range(4):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40):
                call(i)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 40
            print(json.dumps(dict(shape=name, M=M, N=N, K=K, gemm_variant=variant, plan_tile_rows=plan[0], plan_splits=plan[1], us=round(us, 1),
                                  pflops=round(2.0 * M * N * K / (us * 1e-6) / 1e15, 3))), flush=True)
        finally:
            ops.set_gemm_variant(2)
