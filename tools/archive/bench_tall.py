#!/usr/bin/env python
"""Times the LLM prefill gate/up GEMM of ONE streaming chunk (M = 386, N = 2 x 18944, K = 3584, SwiGLU epilogue) with the tall kernel
(default choice) and the 128 x 256 tiles (variant 4), hipEvents over rotating weight buffers.  LCC_TALL_SCHED=1: pinned schedule."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, I = 3584, 18944
ws = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
for M in (386, 300, 448):
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    for variant, name in ((2, "auto(tall)"), (8, "tall"), (4, "big128"), (3, "big256")):
        ops.set_gemm_variant(variant)
        for i in range(4):
            ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for i in range(n):
            ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(json.dumps(dict(M=M, kernel=name, sched=os.environ.get("LCC_TALL_SCHED", "0"), us=round(us, 2),
                              pflops=round(2.0 * M * 2 * I * H / us / 1e9, 4))))
ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
