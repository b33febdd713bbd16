"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by `livecc_amd/`).

Per-stage tensors of the reference arithmetic: ONE forward of HF
`Qwen2VLForConditionalGeneration` on CPU with hooks that record -- and optionally REPLACE -- the input of every vision block
(Q2VL:425-449) and of every decoder layer (Q2VL:559-624):

    vit_in[l]  / vit_out[l]   residual stream before / after vision block l   (vit_in[0] = PatchEmbed output, Q2VL:251-274)
    patch_embed / embeds      what this run computed itself before any replacement: PatchEmbed output, embeddings + scattered rows
    merger_in / vit_merged    PatchMerger input / output (Q2VL:277-290) = the rows scattered into the LLM embeddings
    llm_in[l] / llm_mid[l] / llm_out[l]   decoder layer l: input, after the attention residual (the input of
                              post_attention_layernorm, Q2VL:594-603), output (Q2VL:605-612)
    final_in                  input of the final norm (= llm_out[-1] unless overridden), logits = lm_head of the LAST position

Teacher forcing per layer: `overrides` (the `*_in` tensors of another run, e.g. the bf16 run) replace the inputs, so that the fp32
run computes every layer's TRUTH on exactly the input the bf16 reference -- and the native engine, through
`lcc_debug_set_llm_taps` / `lcc_debug_set_vit_taps` -- saw.  The per-layer error ratio

        rms(native_out - fp32_out) / rms(hf_bf16_out - fp32_out)

then says layer by layer whether the HIP path is as close to the truth as the reference's own bf16 arithmetic is.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from livecc_amd import protocol
from livecc_amd.config import LiveCCConfig


def _text_model(model):
    return model.model.language_model


def _visual(model):
    return model.model.visual


@torch.inference_mode()
def probe(model, cfg: LiveCCConfig, input_ids: np.ndarray, pixel_values: Optional[torch.Tensor], grid_thw,
          overrides: Optional[Dict[str, torch.Tensor]] = None, light: bool = False) -> Dict[str, torch.Tensor]:
    """One prefill forward (no cache) with recording / replacing hooks.  Tensors come back in the model's dtype, on CPU:
    vit_* [depth, P, E], llm_* [L, S, H], final_in [S, H], logits [V] (fp32), vit_merged [P/4, H].
    light=True: no hooks at all -- only `logits` of the last position (a plain free-running forward)."""
    dtype = next(model.parameters()).dtype
    ov = overrides or {}
    rec: Dict[str, list] = {k: [] for k in ("vit_in", "vit_out", "llm_in", "llm_mid", "llm_out")}
    raw: Dict[str, torch.Tensor] = {}
    handles = []

    def pre_replace(key, idx):
        def hook(mod, args, kwargs):
            x = args[0]
            if idx == 0:     # what THIS run computed before any replacement: PatchEmbed output / embeddings + scattered video rows
                raw[key] = x.detach().reshape(-1, x.shape[-1]).clone()
            if key in ov:
                x = ov[key][idx].to(dtype).view_as(x)
            rec[key].append(x.detach().reshape(-1, x.shape[-1]).clone())
            return (x,) + tuple(args[1:]), kwargs
        return hook

    def post_record(key):
        def hook(mod, args, out):
            o = out[0] if isinstance(out, tuple) else out
            rec[key].append(o.detach().reshape(-1, o.shape[-1]).clone())
        return hook

    def pre_record(key):
        def hook(mod, args):
            rec[key].append(args[0].detach().reshape(-1, args[0].shape[-1]).clone())
        return hook

    vis, txt = _visual(model), _text_model(model)
    if light:
        ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).view(1, -1)
        kwargs = dict(input_ids=ids, use_cache=False, logits_to_keep=1,
                      mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(ids.numpy(), cfg)))
        if pixel_values is not None:
            kwargs["pixel_values_videos"] = pixel_values.to(dtype)
            kwargs["video_grid_thw"] = torch.as_tensor([list(grid_thw)], dtype=torch.long)
        model.model.rope_deltas = None
        return dict(logits=model(**kwargs).logits[0, -1].float().clone())
    for l, blk in enumerate(vis.blocks):
        handles.append(blk.register_forward_pre_hook(pre_replace("vit_in", l), with_kwargs=True))
        handles.append(blk.register_forward_hook(post_record("vit_out")))
    merged, merger_in = [], []

    def merger_pre(mod, args, kwargs):
        x = args[0]
        if "merger_in" in ov:
            x = ov["merger_in"].to(dtype).view_as(x)
        merger_in.append(x.detach().clone())
        return (x,) + tuple(args[1:]), kwargs
    handles.append(vis.merger.register_forward_pre_hook(merger_pre, with_kwargs=True))
    handles.append(vis.merger.register_forward_hook(lambda m, a, o: merged.append(o.detach().clone())))
    for l, layer in enumerate(txt.layers):
        handles.append(layer.register_forward_pre_hook(pre_replace("llm_in", l), with_kwargs=True))
        handles.append(layer.post_attention_layernorm.register_forward_pre_hook(pre_record("llm_mid")))
        handles.append(layer.register_forward_hook(post_record("llm_out")))
    final_in = []

    def final_hook(mod, args, kwargs):
        x = args[0]
        if "final_in" in ov:
            x = ov["final_in"].to(dtype).view_as(x)
        final_in.append(x.detach().reshape(-1, x.shape[-1]).clone())
        return (x,) + tuple(args[1:]), kwargs
    handles.append(txt.norm.register_forward_pre_hook(final_hook, with_kwargs=True))

    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long).view(1, -1)
    kwargs = dict(input_ids=ids, use_cache=False, logits_to_keep=1,
                  mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(ids.numpy(), cfg)))
    if pixel_values is not None:
        kwargs["pixel_values_videos"] = pixel_values.to(dtype)
        kwargs["video_grid_thw"] = torch.as_tensor([list(grid_thw)], dtype=torch.long)
    model.model.rope_deltas = None
    try:
        out = model(**kwargs)
    finally:
        for h in handles:
            h.remove()
    res = {k: torch.stack(v) for k, v in rec.items() if v}
    res["final_in"] = final_in[0]
    res["logits"] = out.logits[0, -1].float().clone()
    if merged:
        res["vit_merged"], res["merger_in"] = merged[0], merger_in[0]
    if "vit_in" in raw:
        res["patch_embed"] = raw["vit_in"]
    res["embeds"] = raw["llm_in"]
    return res


def inputs_of(res: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The override set that feeds another run (or the native engine) the inputs this run saw."""
    return {k: res[k] for k in ("vit_in", "merger_in", "llm_in", "final_in") if k in res}


def rms(x: torch.Tensor) -> float:
    return float(x.double().pow(2).mean().sqrt())


def layer_error_table(native: Dict[str, torch.Tensor], ref16: Dict[str, torch.Tensor], ref32: Dict[str, torch.Tensor], keys) -> list:
    """Per stage and layer: rms / max error of the native and of the bf16 reference tensors against the fp32 truth and their ratios."""
    rows = []
    for key in keys:
        n, a, t = native[key].double(), ref16[key].double(), ref32[key].double()
        per_layer = n.dim() == 3
        L = n.shape[0] if per_layer else 1
        for l in range(L):
            nl, al, tl = (n[l], a[l], t[l]) if per_layer else (n, a, t)
            en, ea = (nl - tl), (al - tl)
            scale = float(tl.pow(2).mean().sqrt())
            rn, ra = float(en.pow(2).mean().sqrt()), float(ea.pow(2).mean().sqrt())
            rows.append(dict(stage=key, layer=l, scale_rms=scale, err_native_rms=rn, err_ref16_rms=ra,
                             ratio_rms=rn / max(ra, 1e-30), err_native_max=float(en.abs().max()), err_ref16_max=float(ea.abs().max()),
                             ratio_max=float(en.abs().max()) / max(float(ea.abs().max()), 1e-30),
                             native_vs_ref16_rms=float((nl - al).pow(2).mean().sqrt())))
    return rows
