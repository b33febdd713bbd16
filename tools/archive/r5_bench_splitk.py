#!/usr/bin/env python
"""The split-K projections of ONE streaming chunk (7B, M = 386): 128-row tiles with the round-4 split counts vs the small variable-height
class with the splits the engine now asks for.  20 back-to-back launches per case (+ the slab reduction is NOT included: the consumer
kernels read S slabs either way), A/B/A/B.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, H, I = 386, 3584, 18944
cases = [("down", H, I, 4, 6), ("o", H, H, 4, 6), ("qkv", 4608, H, 3, 4)]
for name, N, K, s_old, s_new in cases:
    ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(3)]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for rep in range(3):
        for tag, variant, S in (("tiles128", 4, s_old), ("vh_small", 15, s_new), ("vh_small_same_splits", 15, s_old)):
            ops.set_gemm_variant(variant)
            for i in range(4):
                ops.linear_partial(x, ws[i % 3], S, packed_shape=(N, K))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for i in range(n):
                ops.linear_partial(x, ws[i % 3], S, packed_shape=(N, K))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n      # includes torch.empty of the slab tensor per call (same for every case)
            if rep:
                print(json.dumps(dict(shape=name, M=M, N=N, K=K, tiles=tag, splits=S, us=round(us, 1), pflops=round(2.0 * M * N * K / us / 1e9, 3))), flush=True)
ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
