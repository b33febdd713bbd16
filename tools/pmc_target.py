#!/usr/bin/env python
"""Target of the rocprofv3 PMC passes (tools/r6_gpu_calls.sh pmc_gemv; earlier rounds: tools/archive/run_profiles.sh): launches ONLY the dominant kernel of the bench -- the decode
gate/up weight-streaming GEMV at LiveCC-7B shapes (packed weights, M = 1; the v2 kernel with the RMSNorm prologue, then the round-1
kernel) -- a few times over rotating weight buffers
(> 256 MiB Infinity Cache in total), so that the per-dispatch FETCH_SIZE / WRITE_SIZE counters can be read per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

H, I = 3584, 18944
dev = torch.device("cuda:0")
if "--gemm" in sys.argv:
    # the LLM prefill gate/up GEMM (SwiGLU epilogue) with 8 streams (M = 3088) and one stream (M = 386): MFMA-utilisation passes
    ws = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
    for M in (3088, 386):
        x = torch.randn(M, H, device=dev).to(torch.bfloat16)
        for i in range(6):
            ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
    # the other LLM prefill GEMMs of ONE streaming chunk (M = 386) as the engine launches them: split-K slabs -- until round 5's small
    # variable-height class on the 128 x 256 tiles (q/k/v 3 splits, o / down 4), since then on 3 row tiles of 128 / 144 rows with
    # q/k/v (N = 4608) 4 splits, o_proj (N = K = 3584) 6, down_proj (N = 3584, K = 18944) 6 (LCC_GEMM_VH_SMALL=0: the old counts)
    small = os.environ.get("LCC_GEMM_VH_SMALL", "1") != "0"
    sq, so = (4, 6) if small else (3, 4)
    x = torch.randn(386, H, device=dev).to(torch.bfloat16)
    xi = torch.randn(386, I, device=dev).to(torch.bfloat16)
    wq = ops.pack_weight((torch.randn(4608, H, device=dev) * 0.02).to(torch.bfloat16))
    wo = ops.pack_weight((torch.randn(H, H, device=dev) * 0.02).to(torch.bfloat16))
    wd = ops.pack_weight((torch.randn(H, I, device=dev) * 0.02).to(torch.bfloat16))
    for i in range(6):
        ops.linear_partial(x, wq, sq, packed_shape=(4608, H))
        ops.linear_partial(x, wo, so, packed_shape=(H, H))
        ops.linear_partial(xi, wd, so, packed_shape=(H, I))
    torch.cuda.synchronize()
    print("ok")
    sys.exit(0)
if "--gemv-rows" in sys.argv:
    # the weight-streaming gate/up GEMV (+ SwiGLU) at M = 8 (one activation fragment per weight fragment) and M = 32 / 64 (MG = 2 / 4):
    # what makes the multi-fragment kernel slower per byte (tools/archive/gpu_call.sh pmc_gemv)
    ws = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(3)]
    for M in (8, 32, 64):
        x = torch.randn(M, H, device=dev).to(torch.bfloat16)
        for i in range(9):
            ops.linear(x, ws[i % 3], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
    torch.cuda.synchronize()
    print("ok")
    sys.exit(0)
if "--attn-decode" in sys.argv:
    # the per-wave decode attention + its split merge at the benchmark's mean cache length (one stream, L = 6,552 keys, 4 KV heads x 7 query
    # heads, 52 key splits as the engine picks them): 13.4 MB of K + V per launch, rotated over 20 caches (268 MB > the Infinity Cache) so
    # that the reads come from HBM as in a decode step, where 14.5 GB of weights pass between two uses of a layer's cache
    NS, L, HQ, HKV = 20, 6552, 28, 4
    kv = ops.KvArena(NS, 1, HKV, 8192, dev)
    kv.buf.copy_((torch.randn(NS, kv.per_slot, device=dev) * 0.5).to(torch.bfloat16))
    kv_len = torch.full((NS,), L, dtype=torch.int32, device=dev)
    q = torch.randn(1, HQ * 128, device=dev).to(torch.bfloat16)
    nsplit = min(64, ((L + 31) // 32 + 3) // 4)
    for i in range(60):
        ops.attn_decode(q, kv, 0, torch.tensor([i % NS], dtype=torch.int32, device=dev), kv_len, HQ, nsplit)
    torch.cuda.synchronize()
    print("ok", nsplit)
    sys.exit(0)
ws = [(torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
x = torch.randn(1, H, device=dev).to(torch.bfloat16)
nw = torch.ones(H, device=dev, dtype=torch.bfloat16)
stats = ops.tile_stats(x)
for i in range(12):   # the shipped decode path (v2): [RMSNorm] gate/up GEMV [SwiGLU] = dgemv_kernel<2,1,1,4,1>
    ops.dgemv_norm_linear(ws[i % 3], x, stats, nw, 1e-6, (2 * I, H), swiglu=True)
for i in range(12):   # and the round-1 kernel for comparison (gemv_skinny_kernel<2,2,...>)
    ops.linear(x, ws[i % 3], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
torch.cuda.synchronize()
print("ok")
