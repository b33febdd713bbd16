#!/usr/bin/env python
"""Target of the round-6 MfmaUtil pass (T2 = ">= 50 % MFMA utilisation on the LLM GEMMs" must be judged from ONE pass over
ALL FOUR LLM GEMMs, not from gate/up alone): q|k|v, o, gate/up, down of a LiveCC-7B decoder layer at M = 3088 (8 co-scheduled chunks) and at
M = 386 (one chunk), each launched exactly as the engine's prefill launches it (same epilogue, same split-K count: ops.gemm_plan), 6 times in
a fixed order over rotating operands.  Writes the launch manifest (label of every GEMM dispatch, in order) next to the counters:

    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d gpurun_out/r6_pmc -o g -- python tools/r6_pmc_llm_gemms.py gpurun_out/r6_pmc/manifest.json
    python tools/r6_summarize_llm_gemm_pmc.py gpurun_out/r6_pmc        # -> profiles/roofline_traffic.json: llm_gemm_mfma_util
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

H, I, QKV = 3584, 18944, 4608
dev = torch.device("cuda:0")
REPS = 6


def w(n, k, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return ops.pack_weight((torch.randn(n, k, device=dev, generator=g) * 0.02).to(torch.bfloat16))


W = dict(qkv=[w(QKV, H, 1), w(QKV, H, 2)], o=[w(H, H, 3), w(H, H, 4)], gate_up=[w(2 * I, H, 5), w(2 * I, H, 6)], down=[w(H, I, 7), w(H, I, 8)])
bias = torch.randn(QKV, device=dev).to(torch.bfloat16)
manifest = []
for M in (3088, 386):
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    xi = torch.randn(M, I, device=dev).to(torch.bfloat16)
    res = torch.randn(M, H, device=dev).to(torch.bfloat16)
    shapes = dict(qkv=(QKV, H, x), o=(H, H, x), gate_up=(2 * I, H, x), down=(H, I, xi))
    for name, (N, K, a) in shapes.items():
        epi = ops.EPI_SWIGLU if name == "gate_up" else (ops.EPI_NONE if name == "qkv" else ops.EPI_RESIDUAL)
        tile, splits = ops.gemm_plan(M, N, K, ops.EPI_NONE if name != "gate_up" else epi)
        use_split = name != "gate_up" and splits > 1          # engine_llm.hip run_layers: tp_qkv / tp_o / tp_dn
        for i in range(REPS):
            if use_split:
                ops.linear_partial(a, W[name][i % 2], splits, packed_shape=(N, K))
            elif name == "qkv":
                ops.linear(a, W[name][i % 2], bias, epi, packed_shape=(N, K))
            elif name == "gate_up":
                ops.linear(a, W[name][i % 2], None, epi, packed_shape=(N, K))
            else:
                ops.linear(a, W[name][i % 2], None, epi, res, packed_shape=(N, K))
            manifest.append(dict(label=f"M{M}_{name}", M=M, N=N, K=K, flops=2 * M * N * K, splits=splits if use_split else 1,
                                 tile_rows=ops.gemm_plan(M, N, K, ops.EPI_NONE if name != "gate_up" else epi, nsplit=splits if use_split else 0)[0], rep=i))
torch.cuda.synchronize()
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6_pmc_manifest.json"
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(manifest, open(out, "w"))
print("ok", len(manifest))
