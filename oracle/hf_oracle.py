"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (`livecc_amd/`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

What it is
----------
The reference (showlab/livecc) has no arithmetic of its own on the hot path: `demo/infer.py:43-47`
instantiates HuggingFace `Qwen2VLForConditionalGeneration` and `demo/infer.py:165-172` calls its
`generate`.  HF `transformers` is a third-party dependency that is NOT vendored under /root/reference
(README pins `transformers>=4.52.4`; installed here and on the GPU box: 5.15.0), so the oracle is
that very implementation run on CPU (SDPA attention, no liger -- liger's Triton kernels cannot run on
CPU, so this is what the reference's `device='cpu'` branch, infer.py:36-47, can execute), driven with
the reference's own turn protocol:

  * first chunk 6 frames, later chunks 2 frames            ref demo/infer.py:121-129
  * input_ids = cat(past_ids, new_ids), KV carried over     ref demo/infer.py:159-172
  * past_ids = sequences[:, :-1]                            ref demo/infer.py:173-174
  * greedy / repetition_penalty / ThresholdLogitsProcessor  ref demo/infer.py:10-23,165-172

plus a numpy restatement of the CPU preprocessing that HF cannot run here (its video processor needs
torchvision, which is not installed): rescale+normalise+patchify,
HF models/qwen2_vl/video_processing_qwen2_vl.py:236-274 and image_processing_backends.py:298-337.

PARITY PINNING: the reference holds no tests, golden vectors or fixtures for this path
(SURVEY.md section 4/8c), so the oracle cannot be pinned against reference-held vectors; it IS the reference's
arithmetic (HF itself, imported, not restated).  The numpy patchify restatement is pinned in
tests/test_oracle.py against HF's own `Qwen2VLVideoProcessor.patchify` tensor code path re-executed
without torchvision, and the golden fixtures under tests/golden/ were produced by
oracle/make_golden.py from this module.

Version drift encoded: transformers 5.15 needs `mm_token_type_ids`; it forwards
`pixel_values_videos` on the first forward of every generate call natively, so the reference's
generation patch (livecc_utils/generation_patch.py:35-39, which raises KeyError on 5.15) is not
installed; `rope_deltas` is module-global state in HF (modeling_qwen2_vl.py:857) and is saved /
restored per stream here.
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence

import numpy as np
import torch

from livecc_amd.config import LiveCCConfig
from livecc_amd import protocol

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


# --------------------------------------------------------------------------------------
# numpy/torch-CPU restatement of the HF video preprocessing (K1 in SURVEY 2.2)
# --------------------------------------------------------------------------------------
def fused_mean_std():
    """image_processing_backends.py:307-311: mean*(1/rescale), std*(1/rescale), rescale=1/255, in fp32."""
    rescale_factor = 1 / 255
    mean = torch.tensor(OPENAI_CLIP_MEAN) * (1.0 / rescale_factor)
    std = torch.tensor(OPENAI_CLIP_STD) * (1.0 / rescale_factor)
    return mean, std


def patchify_normalize_ref(frames_tchw_u8: torch.Tensor, cfg: LiveCCConfig):
    """uint8 [T,3,H,W] -> (pixel_values fp32 [P,1176], grid (t,h,w)).

    normalise : image_processing_backends.py:331-333 -> torchvision normalize = (x.float() - mean) / std
    patchify  : video_processing_qwen2_vl.py:236-274
    """
    assert frames_tchw_u8.dtype == torch.uint8 and frames_tchw_u8.dim() == 4
    mean, std = fused_mean_std()
    x = frames_tchw_u8.to(torch.float32)
    x = (x - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    T, C, H, W = x.shape
    tps, ps, ms = cfg.temporal_patch_size, cfg.patch_size, cfg.spatial_merge_size
    if T % tps:
        pad = tps - T % tps
        x = torch.cat([x, x[-1:].expand(pad, -1, -1, -1)], dim=0)
        T += pad
    gt, gh, gw = T // tps, H // ps, W // ps
    p = x.view(1, gt, tps, C, gh // ms, ms, ps, gw // ms, ms, ps)
    p = p.permute(0, 1, 4, 7, 5, 8, 3, 2, 6, 9)
    flat = p.reshape(gt * gh * gw, C * tps * ps * ps)
    return flat.contiguous(), (gt, gh, gw)


# --------------------------------------------------------------------------------------
# HF model construction
# --------------------------------------------------------------------------------------
def build_hf_model(cfg: LiveCCConfig, dtype=torch.bfloat16, seed: int = 0, init_scale: float = 1.0,
                   attn_implementation: str = "sdpa"):
    """HF Qwen2VLForConditionalGeneration with HF init under torch.manual_seed(seed), built in fp32
    and cast (so that bf16 and fp32 oracles share the same bf16-representable weights when
    `round_weights_to_bf16` is applied)."""
    from transformers import Qwen2VLForConditionalGeneration
    torch.manual_seed(seed)
    hf_cfg = cfg.to_hf(attn_implementation=attn_implementation)
    model = Qwen2VLForConditionalGeneration(hf_cfg)
    model.eval()
    if init_scale != 1.0:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() >= 2:
                    p.mul_(init_scale)
    # deterministic non-trivial norm weights / biases so that affine terms are exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                if n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
    round_weights_to_bf16(model)
    model.to(dtype)
    model.generation_config.do_sample = False
    model.generation_config.top_k = None
    model.generation_config.top_p = None
    model.generation_config.temperature = None
    return model


def parse_weight_spec(spec: str):
    """'tiled:<seed>' | 'decisive:<seed>' -> (variant, seed): the seeded synthetic models of livecc_amd.weights.synthetic_param."""
    variant, _, seed = spec.partition(":")
    if variant not in ("tiled", "decisive"):
        raise ValueError(f"only the seeded synthetic weights (tiled:<seed> / decisive:<seed>) exist offline, got {spec!r}")
    return variant, int(seed or 0)


def build_hf_model_synthetic(cfg: LiveCCConfig, dtype=torch.bfloat16, spec: str = "tiled:0", attn_implementation: str = "sdpa"):
    """HF model at `cfg`'s shapes filled with the seeded synthetic weights of `WeightArena.fill_tiled(seed, variant)` -- bit-identical
    to the MI355X arena, so this model's outputs ARE the oracle for the native path at full shapes.  Built on the meta device (no
    random init of 8 B parameters) and filled window by window."""
    from transformers import Qwen2VLForConditionalGeneration
    from livecc_amd.weights import fill_hf_model_tiled
    variant, seed = parse_weight_spec(spec)
    with torch.device("meta"):
        m = Qwen2VLForConditionalGeneration._from_config(cfg.to_hf(attn_implementation=attn_implementation), dtype=dtype)
    m = m.to_empty(device="cpu")
    fill_hf_model_tiled(m, cfg, seed, variant)
    if cfg.tie_word_embeddings:
        m.tie_weights()
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if "inv_freq" in name:
                dim = buf.numel() * 2
                theta = 10000.0 if "visual" in name else cfg.rope_theta
                buf.copy_(1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim)))
    m.eval()
    m.generation_config.do_sample = False
    m.generation_config.top_k = m.generation_config.top_p = m.generation_config.temperature = None
    return m


def fake_quantize_llm_fp8(model) -> None:
    """Replace every LLM Linear weight (q/k/v/o, gate/up/down, lm_head) by its OCP-e4m3 per-output-row quantisation
    q * scale (the values the native fp8 weight path computes with), rounded to the model's dtype.  On an fp32 model the
    products q*scale are exact up to one fp32 rounding = the truth for the fp8 path; on a bf16 model they carry bf16's
    2^-9 weight rounding = what the reference's dtype would do with the dequantised checkpoint.  TEST INFRASTRUCTURE."""
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() != 2 or "visual" in n or "embed_tokens" in n:
                continue
            if not any(k in n for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj", "lm_head")):
                continue
            q, sc = quantize_fp8_rows(p.float())
            p.copy_(dequantize_fp8_rows(q, sc).to(p.dtype))


def round_weights_to_bf16(model) -> None:
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).to(p.dtype))


class ThresholdLogitsProcessor:
    """ref demo/infer.py:10-23, restated verbatim in behaviour (stateful `count`)."""

    def __init__(self, token_id: int, base_threshold: float, step: float):
        self.token_id, self.base_threshold, self.step, self.count = token_id, base_threshold, step, 0

    def __call__(self, input_ids, scores):
        threshold = self.base_threshold + self.step * self.count
        low_confidence = torch.softmax(scores, dim=-1)[:, self.token_id] <= threshold
        if low_confidence.any():
            scores[low_confidence, self.token_id] = -float("inf")
        self.count += 1
        return scores


class OracleStream:
    """One video stream replayed through HF `generate` with the reference's protocol."""

    def __init__(self, model, cfg: LiveCCConfig, device=None, past_key_values=None):
        """`device`: where the inputs go (the model's device; default CPU).  `past_key_values`: an HF `Cache` object to start
        from (the plugin tests pass a `NativeKVCache`)."""
        self.model, self.cfg = model, cfg
        self.device = torch.device(device) if device is not None else None
        self.past_key_values = past_key_values
        self.past_ids: Optional[torch.Tensor] = None
        self.rope_deltas = None
        self.last_prefill_seconds = 0.0

    @torch.inference_mode()
    def turn(self, new_ids: Sequence[int], pixel_values: Optional[torch.Tensor], grid_thw,
             max_new_tokens: int = 16, repetition_penalty: float = 1.05, force_length: bool = True,
             streaming_eos: Optional[tuple] = None, teacher_tokens: Optional[Sequence[int]] = None, tick=None):
        """Run one turn.  Returns dict(sequences, new_tokens, logits[list of fp32 [V]], scores)."""
        cfg, model = self.cfg, self.model
        new_ids_t = torch.as_tensor(np.asarray(new_ids), dtype=torch.long).view(1, -1)
        input_ids = new_ids_t if self.past_ids is None else torch.cat([self.past_ids, new_ids_t], dim=1)
        kwargs = {}
        if pixel_values is not None:
            kwargs["pixel_values_videos"] = pixel_values
            kwargs["video_grid_thw"] = torch.as_tensor([list(grid_thw)], dtype=torch.long)
        kwargs["mm_token_type_ids"] = torch.as_tensor(protocol.mm_token_type_ids(input_ids.numpy(), cfg))
        if self.device is not None:
            input_ids = input_ids.to(self.device)
            kwargs = {k: v.to(self.device) for k, v in kwargs.items()}
        procs = None
        if streaming_eos is not None:
            from transformers import LogitsProcessorList
            procs = LogitsProcessorList([ThresholdLogitsProcessor(*streaming_eos)])
        force = None
        if teacher_tokens is not None:
            from transformers import LogitsProcessorList
            force = _ForceTokens(list(teacher_tokens), input_ids.shape[1])
            procs = LogitsProcessorList(([procs[0]] if procs is not None else []) + [force])
        if tick is not None:      # progress callback (input_ids, scores) -> None, called once per generated token (CPU baseline timing)
            from transformers import LogitsProcessorList

            class _Tick:
                def __call__(self, input_ids, scores):
                    tick(input_ids, scores)
                    return scores
            procs = LogitsProcessorList(list(procs or []) + [_Tick()])
        # restore this stream's rope_deltas (HF keeps it on the module: modeling_qwen2_vl.py:857)
        model.model.rope_deltas = self.rope_deltas
        t0 = time.perf_counter()
        out = model.generate(
            input_ids=input_ids, past_key_values=self.past_key_values, return_dict_in_generate=True,
            do_sample=False, repetition_penalty=repetition_penalty, logits_processor=procs,
            max_new_tokens=max_new_tokens, min_new_tokens=max_new_tokens if force_length else None,
            pad_token_id=cfg.eos_token_id, eos_token_id=cfg.eos_token_id,
            output_logits=True, output_scores=True, **kwargs)
        self.seconds = time.perf_counter() - t0
        self.rope_deltas = model.model.rope_deltas
        self.past_key_values = out.past_key_values
        self.past_ids = out.sequences[:, :-1].cpu()
        n_in = input_ids.shape[1]
        return dict(sequences=out.sequences[0].cpu().clone(), new_tokens=out.sequences[0, n_in:].tolist(),
                    logits=[l[0].float().cpu().clone() for l in out.logits],
                    scores=(force.pre if force is not None else [s[0].float().cpu().clone() for s in out.scores]), n_input=n_in)


class _ForceTokens:
    """Teacher forcing: after all other processors, force the argmax to a prescribed token so that the
    oracle can be stepped along a given token sequence while still reporting its own raw logits."""

    def __init__(self, tokens: List[int], n_prompt: int):
        self.tokens, self.n_prompt = tokens, n_prompt
        self.pre: List[torch.Tensor] = []      # processed scores before forcing (the oracle's own preference)

    def __call__(self, input_ids, scores):
        self.pre.append(scores[0].float().cpu().clone())
        i = input_ids.shape[1] - self.n_prompt
        if i < len(self.tokens):
            scores = scores.clone()
            scores[:, self.tokens[i]] = float("inf")
        return scores


@torch.inference_mode()
def vit_forward_ref(model, pixel_values: torch.Tensor, grid_thw) -> torch.Tensor:
    """Merged ViT embeddings [P/4, hidden] exactly as Qwen2VLModel.get_video_features computes them."""
    g = torch.as_tensor([list(grid_thw)], dtype=torch.long)
    out = model.model.get_video_features(pixel_values, g)
    return torch.cat(list(out.pooler_output), dim=0)


def replay_stream(model, cfg: LiveCCConfig, frames_tchw_u8: torch.Tensor, builder: protocol.TurnBuilder,
                  max_new_tokens: int = 16, repetition_penalty: float = 1.05, max_turns: Optional[int] = None,
                  streaming_eos: Optional[tuple] = None):
    """Whole-clip replay = ref `live_cc_once_for_evaluation` (demo/infer.py:244-310) on synthetic ids."""
    stream = OracleStream(model, cfg)
    results = []
    chunks = protocol.split_clip(frames_tchw_u8.shape[0])
    for ti, (a, b) in enumerate(chunks):
        if max_turns is not None and ti >= max_turns:
            break
        pv, grid = patchify_normalize_ref(frames_tchw_u8[a:b], cfg)
        ids = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        r = stream.turn(ids, pv, grid, max_new_tokens=max_new_tokens, repetition_penalty=repetition_penalty,
                        streaming_eos=streaming_eos)
        r["turn_ids"], r["grid"], r["seconds"] = ids, grid, stream.seconds
        results.append(r)
    return results
