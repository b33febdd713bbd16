#!/usr/bin/env python
"""Kernel micro-benchmarks on the MI355X (not part of the product): the weight-streaming GEMV at the LiveCC-7B decode
shapes, per weight layout and kernel variant, timed with HIP events on the launch stream.  Weight buffers are rotated so
that nothing is served from L2 / Infinity Cache (256 MiB).  Prints one JSON object per measurement."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402


def time_gemv(M, N, K, packed, variant, mode, iters=40):
    dev = torch.device("cuda:0")
    nbuf = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ops.set_gemv_variant(variant)
    S = ops.gemv_num_splits(N, K)

    def run(w):
        if mode == "swiglu":
            return ops.linear(x, w, None, ops.EPI_SWIGLU, packed_shape=(N, K) if packed else None)
        if mode == "partial":
            return ops.linear_partial(x, w, S, packed_shape=(N, K) if packed else None)
        return ops.linear(x, w, None, packed_shape=(N, K) if packed else None)

    for w in ws[:2]:
        run(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(iters):
        w = ws[i % nbuf]
        e0.record()
        run(w)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ops.set_gemv_variant(ops.GEMV_DEFAULT_VARIANT)
    ts.sort()
    med = ts[len(ts) // 2]
    gb = N * K * 2 / 1e9
    return dict(kernel="gemv", mode=mode, M=M, N=N, K=K, packed=packed, variant=variant, splits=S if mode == "partial" else 1,
                us_median=round(med, 2), us_min=round(ts[0], 2), TBps_median=round(gb / med * 1e3, 3), TBps_best=round(gb / ts[0] * 1e3, 3))


def main():
    H, I, V, QKV = 3584, 18944, 152064, 4608
    cases = [("swiglu", 2 * I, H), ("partial", QKV, H), ("partial", H, H), ("partial", H, I), ("plain", V, H)]
    Ms = [1] if "--quick" in sys.argv else [1, 8]
    for mode, N, K in cases:
        for M in Ms:
            for packed in (False, True):
                for variant in (0, 1, 2):
                    print(json.dumps(time_gemv(M, N, K, packed, variant, mode)), flush=True)


if __name__ == "__main__":
    main()
