#!/usr/bin/env python
"""Round-5 A/B target: the 8-wave GEMM at the engine's prefill / vision-tower shapes, one process per (library build, LCC_GEMM_* knobs).
hipEvents around every launch, weights rotated over buffers larger than the Infinity Cache where that matters, median + min over the
launches.  One JSON line per shape; `tag` (argv[1]) names the configuration."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
H, I = 3584, 18944
g = torch.Generator(device="cpu").manual_seed(3)


def rnd(*shape, s=1.0):
    return (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).to(dev)


def bench(name, M, N, K, epi, iters=12, nbuf=2):
    if only and name not in only:
        return
    ws = [ops.pack_weight(rnd(N, K, s=0.02)) for _ in range(nbuf)]
    x = rnd(M, K)
    res = rnd(M, N) if epi == ops.EPI_RESIDUAL else None

    def run(w):
        return ops.linear(x, w, None, epi, res, packed_shape=(N, K))
    for w in ws:
        run(w)
    torch.cuda.synchronize()
    ts = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(ws[i % nbuf])
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print(json.dumps(dict(tag=tag, shape=name, M=M, N=N, K=K, us_median=round(med, 1), us_min=round(ts[0], 1),
                          pflops_median=round(2.0 * M * N * K / med / 1e9, 3))), flush=True)
    del ws


bench("gate_up_M3088", 3088, 2 * I, H, ops.EPI_SWIGLU)
bench("down_M3088", 3088, H, I, ops.EPI_RESIDUAL)
bench("qkv_M3088", 3088, 4608, H, ops.EPI_NONE, nbuf=3)
bench("o_M3088", 3088, H, H, ops.EPI_RESIDUAL, nbuf=3)
bench("gate_up_M1131", 1131, 2 * I, H, ops.EPI_SWIGLU)
bench("gate_up_M386_tall", 386, 2 * I, H, ops.EPI_SWIGLU)
bench("gate_up_M24058", 24058, 2 * I, H, ops.EPI_SWIGLU, iters=4)
bench("vit_fc1_P11648", 11648, 5120, 1280, ops.EPI_QUICK_GELU, nbuf=3)
bench("vit_fc2_P11648", 11648, 1280, 5120, ops.EPI_RESIDUAL, nbuf=3)
bench("vit_qkv_P11648", 11648, 3840, 1280, ops.EPI_NONE, nbuf=3)
bench("vit_proj_P11648", 11648, 1280, 1280, ops.EPI_RESIDUAL, nbuf=3)
bench("vit_fc1_P1456", 1456, 5120, 1280, ops.EPI_QUICK_GELU, nbuf=3)
bench("vit_fc2_P1456", 1456, 1280, 5120, ops.EPI_RESIDUAL, nbuf=3)
