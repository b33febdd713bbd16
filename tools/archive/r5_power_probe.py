#!/usr/bin/env python
"""Is the 8-wave GEMM clock / power bound?  The same launch sequence (gate/up, M = 3088 and 9048, 20 back-to-back launches, A/B/A/B)
on random operands and on zero-filled operands: the instruction stream, the addresses and the DMA traffic are identical, only the bit
toggling in the MFMA / LDS / register datapaths differs (cdna_hip_programming.md section 5.4 rule 25: zero-filled operands run the
guide's GEMM templates 15-18 % faster)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, I = 3584, 18944
N, K = 2 * I, H
w_rand = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
w_zero = [torch.zeros_like(w_rand[0]) for _ in range(2)]
for M in (3088, 9048):
    x_rand = torch.randn(M, K, device=dev).to(torch.bfloat16)
    x_zero = torch.zeros_like(x_rand)
    for rep in range(3):
        for name, x, ws in (("random", x_rand, w_rand), ("zeros", x_zero, w_zero), ("random_x_zero_w", x_rand, w_zero)):
            for i in range(4):
                ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(N, K))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for i in range(n):
                ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(N, K))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            if rep:      # the first pass of a process carries the first-run penalty
                print(json.dumps(dict(shape="gate_up", M=M, operands=name, us=round(us, 1), pflops=round(2.0 * M * N * K / us / 1e9, 3))), flush=True)
