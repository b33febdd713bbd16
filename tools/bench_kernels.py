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


def time_gemm(M, N, K, epi, variant, iters=20):
    dev = torch.device("cuda:0")
    nbuf = 3
    ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_RESIDUAL else None
    ops.set_gemm_variant(variant)
    S = 0
    if epi == "partial":
        S = 4

    def run(w):
        if epi == "partial":
            return ops.linear_partial(x, w, S, packed_shape=(N, K))
        return ops.linear(x, w, None, epi, res, packed_shape=(N, K))

    for w in ws:
        run(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(ws[i % nbuf])
    e1.record()
    e1.synchronize()
    ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    us = e0.elapsed_time(e1) * 1e3 / iters
    return dict(kernel="gemm_tiled", M=M, N=N, K=K, epi=str(epi), variant=variant, us=round(us, 1),
                TFLOPs=round(2.0 * M * N * K / us / 1e6, 1))


def main_gemm():
    H, I = 3584, 18944
    cases = [(386, 2 * I, H, ops.EPI_SWIGLU), (386, 4608, H, ops.EPI_NONE), (386, H, H, "partial"), (386, H, I, "partial"),
             (3088, 2 * I, H, ops.EPI_SWIGLU), (3088, H, I, ops.EPI_RESIDUAL), (3088, 4608, H, ops.EPI_NONE),
             (1456, 3840, 1280, ops.EPI_NONE), (1456, 5120, 1280, ops.EPI_QUICK_GELU), (1456, 1280, 5120, ops.EPI_RESIDUAL),
             (11648, 5120, 1280, ops.EPI_QUICK_GELU), (11648, 1280, 5120, ops.EPI_RESIDUAL)]
    for M, N, K, epi in cases:
        for variant in (2, 3, 4, 5, 6):
            print(json.dumps(time_gemm(M, N, K, epi, variant)), flush=True)


def main_mall():
    """Does a decode GEMV run faster when its weights were just read by ANOTHER kernel (Infinity Cache / MALL warm)?
    cold = rotating > 600 MB of weights; warm = each buffer is touched by a plain read kernel right before its GEMV (the GEMV
    alone is timed).  Decides whether it pays to prefetch the next GEMV's weights under the latency-bound small kernels."""
    dev = torch.device("cuda:0")
    H, I, QKV = 3584, 18944, 4608
    for name, mode, N, K in [("qkv", "partial", QKV, H), ("o", "partial", H, H), ("down", "partial", H, I), ("gate_up", "swiglu", 2 * I, H)]:
        nbuf = max(2, int(700e6 // (N * K * 2)) + 1)
        ws = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nbuf)]
        x = torch.randn(1, K, device=dev).to(torch.bfloat16)
        S = ops.gemv_num_splits(N, K)

        def run(w):
            if mode == "swiglu":
                return ops.linear(x, w, None, ops.EPI_SWIGLU, packed_shape=(N, K))
            return ops.linear_partial(x, w, S, packed_shape=(N, K))

        for frac in (0.0, 1.0, 0.5, 0.25):
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(30):
                w = ws[i % nbuf]
                if frac > 0:
                    flat = w.view(-1).view(torch.int32)
                    flat[: int(flat.numel() * frac)].sum()      # a plain (temporal) read of the leading part of the weights
                e0.record()
                run(w)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(json.dumps(dict(kernel="gemv_mall", shape=name, N=N, K=K, MB=round(N * K * 2 / 1e6, 1), warm_fraction=frac,
                                  us_median=round(ts[len(ts) // 2], 2), us_min=round(ts[0], 2))), flush=True)


def main_attn():
    """One layer's decode attention chain at 7B head counts (28 q / 4 kv heads), B streams with L cached keys each."""
    import ctypes
    from livecc_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.load()
    Hq, Hkv, D, NS = 28, 4, 128, 4
    for B, L in [(1, 1200), (1, 6000), (1, 12000), (8, 6000), (8, 12000), (1, 32000)]:
        Lmax = (L + 64 + 31) // 32 * 32
        kv = ops.KvArena(B, 1, Hkv, Lmax, dev)
        kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
        qd = (Hq + 2 * Hkv) * D
        part = torch.randn(NS, B, qd, device=dev) * 0.5
        bias = (torch.randn(qd, device=dev) * 0.1).to(torch.bfloat16)
        cos = torch.rand(B, 64, device=dev).to(torch.bfloat16)
        sin = torch.rand(B, 64, device=dev).to(torch.bfloat16)
        slots = torch.arange(B, dtype=torch.int32, device=dev)
        kv_len = torch.full((B,), L, dtype=torch.int32, device=dev)
        ntile = (L + 1 + 31) // 32
        ns_sep = max(1, min(64, (ntile + 3) // 4))
        q = torch.empty(B, Hq * D, dtype=torch.bfloat16, device=dev)
        out = torch.empty_like(q)
        ws_o = torch.empty(B * Hkv * 64 * 16 * 128, dtype=torch.float32, device=dev)
        ws_ml = torch.empty(B * Hkv * 64 * 16 * 2, dtype=torch.float32, device=dev)
        cnt = torch.zeros(256, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for variant, ns_f in [(0, 0), (1, 8), (1, 16), (1, 32), (2, 8), (2, 16)]:
            ns_fused = max(1, min(ns_f, (ntile + 3) // 4)) if ns_f else 1
            us = ctypes.c_float(0)
            rc = lib.lcc_debug_bench_attn_decode(variant, 300, part.data_ptr(), NS, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                 slots.data_ptr(), kv_len.data_ptr(), kv.ptrs.data_ptr(), kv.lay, 0, q.data_ptr(),
                                                 out.data_ptr(), B, Hq, ns_sep, ns_fused, ws_o.data_ptr(), ws_ml.data_ptr(),
                                                 cnt.data_ptr(), ctypes.byref(us), st)
            _lib.check(rc, "lcc_debug_bench_attn_decode")
            print(json.dumps(dict(kernel="attn_decode_chain", B=B, L=L, variant=["3 kernels", "fused+combine", "fused in-launch merge"][variant],
                                  nsplit=ns_sep if variant == 0 else ns_fused, us_per_layer=round(us.value, 2))), flush=True)


def main_resize():
    """Antialias bicubic uint8 resize (8f-1): decoder-order frames -> model resolution, algorithmic bytes = src + dst."""
    from livecc_amd import resize as R
    dev = torch.device("cuda:0")
    for T, hi, wi, ho, wo in [(2, 1080, 1920, 392, 728), (6, 1080, 1920, 392, 728), (32, 1080, 1920, 392, 728), (2, 720, 1280, 392, 700), (480, 360, 640, 280, 504)]:
        x = torch.randint(0, 256, (T, hi, wi, 3), dtype=torch.uint8, device=dev)
        R.resize_bicubic_aa(x, ho, wo, "THWC")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            R.resize_bicubic_aa(x, ho, wo, "THWC")
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        alg = T * 3 * (hi * wi + ho * wo)
        print(json.dumps(dict(kernel="resize_bicubic_aa_u8", T=T, src=[hi, wi], dst=[ho, wo], us=round(us, 1),
                              algorithmic_GBps=round(alg / us / 1e3, 1), frames_per_s=round(T / us * 1e6))), flush=True)


def main():
    if "--gemm" in sys.argv:
        return main_gemm()
    if "--resize" in sys.argv:
        return main_resize()
    if "--attn" in sys.argv:
        return main_attn()
    if "--mall" in sys.argv:
        return main_mall()
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
