#!/usr/bin/env python
"""Which side bounds gemm_big_kernel<256> (LLM prefill gate/up, M = 3088 x N = 37888 x K = 3584, SwiGLU)?  Run three times:
LCC_GEMM_DIAG=0 (the kernel), =2 (no LDS-DMA after the prologue: MFMAs + fragment reads + barriers only), =3 (no MFMAs: DMA ring +
barriers only).  hipEvents, rotating weights.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
if len(sys.argv) > 1:            # python tools/bench_gemm_diag.py <gemm variant>
    ops.set_gemm_variant(int(sys.argv[1]))
H, I = 3584, 18944
if len(sys.argv) > 2 and sys.argv[2] == "down":      # the down projection (N = 3584, K = 18944, residual epilogue): 13 x 14 = 182 blocks of 256 rows at M = 3088
    ws = [ops.pack_weight((torch.randn(H, I, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
    for M in (3088, 1544, 9048):
        x, res = torch.randn(M, I, device=dev).to(torch.bfloat16), torch.randn(M, H, device=dev).to(torch.bfloat16)
        for i in range(4):
            ops.linear(x, ws[i % 2], None, ops.EPI_RESIDUAL, res, packed_shape=(H, I))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for i in range(n):
            ops.linear(x, ws[i % 2], None, ops.EPI_RESIDUAL, res, packed_shape=(H, I))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(json.dumps(dict(shape="down_proj", variant=sys.argv[1], M=M, us=round(us, 1), pflops=round(2.0 * M * H * I / us / 1e9, 3))))
    sys.exit(0)
ws = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
for M in (1131, 3088, 1131, 3088, 9048):      # the first shape of a process carries a ~10 % first-run penalty: read the repeats
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    for i in range(4):
        ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for i in range(n):
        ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(json.dumps(dict(diag=os.environ.get("LCC_GEMM_DIAG", "0"), variant=sys.argv[1] if len(sys.argv) > 1 else "default", M=M, us=round(us, 1), pflops_if_real=round(2.0 * M * 2 * I * H / us / 1e9, 3))))
