#!/usr/bin/env python
"""Is `weights.quantize_fp8_rows` bit-identical on the CPU (the oracle's side) and on the GPU (WeightArena's side)?  Seeded synthetic 72B
parameters, both devices, byte-for-byte comparison of the e4m3 patterns and the row scales."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.weights import hf_param_shapes, quantize_fp8_rows, synthetic_param  # noqa: E402

cfg = get_config("qwen2vl-72b")
shapes = hf_param_shapes(cfg)
dev = torch.device("cuda:0")
for name in ("language_model.layers.0.self_attn.q_proj.weight", "language_model.layers.0.mlp.gate_proj.weight",
             "language_model.layers.7.mlp.down_proj.weight", "language_model.layers.41.self_attn.o_proj.weight"):
    wc = synthetic_param(name, shapes, 0, "cpu")
    wg = synthetic_param(name, shapes, 0, dev)
    same_w = bool(torch.equal(wc, wg.cpu()))
    qc, sc = quantize_fp8_rows(wc.float())
    qg, sg = quantize_fp8_rows(wg.float())
    dq = (qc != qg.cpu())
    x = (wc.float() / sc[:, None])
    print(json.dumps(dict(param=name, shape=list(wc.shape), weights_equal=same_w, scales_equal=bool(torch.equal(sc, sg.cpu())),
                          q_mismatch=int(dq.sum()), q_total=dq.numel(),
                          max_abs_scaled_at_mismatch=float(x[dq].abs().max()) if dq.any() else 0.0)), flush=True)
