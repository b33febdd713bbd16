"""Operator plugins in the reference's own style: `apply_livecc_amd_kernel_to_qwen2_vl()` is the MI355X counterpart of
liger's `apply_liger_kernel_to_qwen2_vl()` (called at ref demo/infer.py:2-3): it rebinds names inside
`transformers.models.qwen2_vl.modeling_qwen2_vl` BEFORE the HF model is constructed, so that an unmodified HF
`Qwen2VLForConditionalGeneration` on a ROCm device executes the HIP kernels for the patched operators.  It exists so that
each kernel can be swapped in one at a time under the HF module graph; the full native engine
(`livecc_amd.modeling`) is the production path.

Patched slots: RMSNorm (Q2VL:96-110), LayerNorm (Q2VL:428-429, 281), SwiGLU MLP activation (Q2VL:453-466).
All wrappers require bf16 GPU tensors and raise otherwise (no CPU path).
"""
from __future__ import annotations

import torch

from . import ops


def apply_livecc_amd_kernel_to_qwen2_vl(rms_norm: bool = True, layer_norm: bool = True, swiglu: bool = True) -> None:
    import transformers.models.qwen2_vl.modeling_qwen2_vl as m

    if rms_norm:
        class LccRMSNorm(m.Qwen2VLRMSNorm):
            def forward(self, hidden_states):
                x = hidden_states.contiguous()
                return ops.rmsnorm(x.view(-1, x.shape[-1]), self.weight, self.variance_epsilon).view_as(x)
        m.Qwen2VLRMSNorm = LccRMSNorm

    if layer_norm:
        class LccLayerNorm(torch.nn.LayerNorm):
            def forward(self, x):
                x = x.contiguous()
                return ops.layernorm(x.view(-1, x.shape[-1]), self.weight, self.bias, self.eps).view_as(x)
        m.LayerNorm = LccLayerNorm

    if swiglu:
        class LccMLP(m.Qwen2MLP):
            def forward(self, x):
                g = self.gate_proj(x).contiguous()
                u = self.up_proj(x).contiguous()
                return self.down_proj(ops.swiglu(g.view(-1, g.shape[-1]), u.view(-1, u.shape[-1])).view_as(g))
        m.Qwen2MLP = LccMLP
