#!/usr/bin/env python
"""Does the row stride of the activation matrix matter to the 8-wave GEMM's LDS-DMA (L2 channel spread of a row tile's 128-byte lines)?
The same GEMM with the activations in a [M, K + pad] buffer (lda = K + pad), 20 back-to-back launches, weights rotated.  One JSON line per
(shape, pad)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
H, I = 3584, 18944


def run(name, M, N, K, epi, pads):
    ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_RESIDUAL else None
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for pad in pads:
        buf = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)

        def go(w):
            _lib.check(lib.lcc_gemm_bf16(buf.data_ptr(), K + pad, w.data_ptr(), K, 1, None, res.data_ptr() if res is not None else None, N,
                                         out.data_ptr(), out.shape[1], M, N, K, epi, None, 0, st), "lcc_gemm_bf16")
        for i in range(4):
            go(ws[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for i in range(n):
            go(ws[i % 2])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(json.dumps(dict(shape=name, M=M, N=N, K=K, lda=K + pad, row_stride_lines=(K + pad) * 2 / 128, us=round(us, 1),
                              pflops=round(2.0 * M * N * K / us / 1e9, 3))), flush=True)


pads = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (0, 64, 192, 512, 8)
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, M, N, K, epi in (("gate_up_M3088", 3088, 2 * I, H, ops.EPI_SWIGLU), ("down_M3088", 3088, H, I, ops.EPI_RESIDUAL),
                          ("qkv_M3088", 3088, 4608, H, ops.EPI_NONE), ("o_M3088", 3088, H, H, ops.EPI_RESIDUAL),
                          ("gate_up_M386", 386, 2 * I, H, ops.EPI_SWIGLU), ("gate_up_M1131", 1131, 2 * I, H, ops.EPI_SWIGLU),
                          ("gate_up_M9048", 9048, 2 * I, H, ops.EPI_SWIGLU),
                          ("vit_fc2_P11648", 11648, 1280, 5120, ops.EPI_RESIDUAL), ("vit_fc1_P11648", 11648, 5120, 1280, ops.EPI_QUICK_GELU)):
    if only is None or name in only:
        run(name, M, N, K, epi, pads)
