#!/usr/bin/env python
"""Fixed cost vs streaming rate of the decode GEMVs (pipeline v2) on MI355X: the [RMSNorm] gate/up GEMV [SwiGLU] and the down GEMV
[residual] at LiveCC-7B's K, with N scaled to 1/4, 1/2, 1 and 2 times the model's -- bf16 and fp8 (e4m3) weights.  hipEvents over
rotating weight buffers (every launch streams bytes that are not in the 256-MiB Infinity Cache).  One JSON line per case, then the
least-squares fit  t = fixed_us + bytes / rate  per kernel: `fixed_us` is what a launch costs before / after its weight stream
(dispatch ramp, prologue, cross-wave reduce, epilogue, drain), `rate` the marginal HBM rate."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, I = 3584, 18944
ITERS = 30


def timed(fn, nbuf):
    for i in range(nbuf + 2):
        fn(i % nbuf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(ITERS):
        fn(i % nbuf)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


def make_w(N, K, fp8, nbuf):
    ws = []
    for _ in range(nbuf):
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        if fp8:
            ws.append(ops.quantize_fp8(w))
        else:
            ws.append((ops.pack_weight(w), None))
        del w
    return ws


results = {}
KINDS = [k for k in ("gate_up", "down", "o") if k in sys.argv[1:]] or ["gate_up", "down"]
for fp8 in ((False,) if "--bf16" in sys.argv else (False, True)):
    bpe = 1 if fp8 else 2
    for kind in KINDS:
        pts = []
        for scale in (0.25, 0.5, 1.0, 2.0):
            if kind == "gate_up":
                N, K = int(2 * I * scale) // 64 * 64, H
            elif kind == "down":
                N, K = int(H * scale) // 64 * 64, I
            else:
                N, K = int(H * scale) // 64 * 64, H
            nbytes = N * K * bpe
            nbuf = max(2, int(np.ceil(600e6 / nbytes)))          # > 2 x the Infinity Cache between two uses of a buffer
            ws = make_w(N, K, fp8, nbuf)
            M = 1
            if kind == "gate_up":
                h = torch.randn(M, K, device=dev).to(torch.bfloat16)
                nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
                st = ops.tile_stats(h)
                us = timed(lambda i: ops.dgemv_norm_linear(ws[i][0], h, st, nw, 1e-6, (N, K), swiglu=True, wscale=ws[i][1]), nbuf)
            else:
                x = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
                hres = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
                us = timed(lambda i: ops.dgemv_resid_(ws[i][0], x, hres, (N, K), wscale=ws[i][1]), nbuf)
            pts.append((nbytes, us))
            print(json.dumps(dict(kernel=kind, weights="fp8" if fp8 else "bf16", N=N, K=K, weight_mb=round(nbytes / 1e6, 1), us=round(us, 2),
                                  tb_per_s=round(nbytes / us / 1e6, 3))), flush=True)
            del ws
            torch.cuda.empty_cache()
        b = np.array([p[0] for p in pts], dtype=np.float64)
        t = np.array([p[1] for p in pts], dtype=np.float64)
        A = np.stack([np.ones_like(b), b], 1)
        (fixed, slope), *_ = np.linalg.lstsq(A, t, rcond=None)
        results[f"{kind}_{'fp8' if fp8 else 'bf16'}"] = dict(fixed_us=round(float(fixed), 2), marginal_tb_per_s=round(1e-6 / float(slope), 3))
print(json.dumps(dict(fit=results)))
