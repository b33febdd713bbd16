"""Operator plugins in the reference's own style: every slot through which the reference plugs kernels into HF
`Qwen2VLForConditionalGeneration`, filled with the MI355X kernels so that an UNMODIFIED HF model on a ROCm device runs them --
one kernel at a time under the HF module graph.  (The production path is the full native engine, `livecc_amd.modeling`; this
module is the drop-in for callers that keep the HF model object.)

  (1) liger pattern          `apply_livecc_amd_kernel_to_qwen2_vl()` = the MI355X counterpart of `apply_liger_kernel_to_qwen2_vl()`
                             (ref demo/infer.py:2-3): rebinds names in `transformers.models.qwen2_vl.modeling_qwen2_vl` BEFORE the
                             model is constructed -- RMSNorm (Q2VL:96-110), LayerNorm (Q2VL:428-429, 281), the SwiGLU MLP (Q2VL:453-466)
                             and `apply_multimodal_rotary_pos_emb` (Q2VL:180-222);
  (2) attention registry     `AttentionInterface.register("livecc_amd", ...)`, selected by `attn_implementation="livecc_amd"` exactly
                             where the reference passes 'flash_attention_2' (ref demo/infer.py:46): ViT slices (non-causal, d = 80)
                             and the LLM (causal GQA over the native cache, d = 128);
  (3) KV cache               `NativeKVCache`, an HF `Cache` whose `update()` appends in place into the pre-allocated arena the
                             attention kernels read (HF:cache_utils.py:127-146 does `torch.cat` of the whole K and V), passed as
                             `past_key_values=`;
  (4) LogitsProcessor        `ThresholdLogitsProcessor` lives in infer.py (executed inside the native sampler);
  (6) video reader backend   `register_video_reader_backend()`: 'decord+' (ref video_process_patch.py:85-86).

All wrappers require bf16 tensors on the GPU and raise otherwise (no CPU path).
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Tuple

import torch

from . import ops

ATTN_NAME = "livecc_amd"
_SAVED = {}


# ------------------------------------------------------------------------------------------------------------------------------
# (1) liger-style rebinding
# ------------------------------------------------------------------------------------------------------------------------------
def lcc_apply_multimodal_rotary_pos_emb(q, k, cos, sin, mrope_section, unsqueeze_dim=1):
    """Same signature as HF's function (Q2VL:180-222): q [B,Hq,S,128], k [B,Hkv,S,128], cos/sin [3,B,S,128] in the model dtype.
    The per-axis channel selection is index bookkeeping (host-side views); the rotation itself -- HF's bf16 op sequence
    bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)) -- runs in the M-RoPE kernel (bit-exact vs HF, tests/test_gpu_ops.py)."""
    if not q.is_cuda:
        # a module-level function rebinding is process-wide (as liger's is): tensors of models that do not live on the GPU are not
        # this library's business and go to HF's own function
        return _SAVED["apply_multimodal_rotary_pos_emb"](q, k, cos, sin, mrope_section, unsqueeze_dim)
    if unsqueeze_dim != 1 or q.dtype != torch.bfloat16 or q.shape[-1] != 128:
        raise ValueError("livecc_amd M-RoPE: bf16 tensors [B, heads, S, 128] expected")
    sec = list(mrope_section) * 2
    cs = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)      # [B,S,128], halves identical
    sn = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    B, Hq, S, D = q.shape
    Hkv = k.shape[1]
    q_out, k_out = torch.empty_like(q), torch.empty_like(k)
    for b in range(B):
        # q and k rows side by side as the fused q|k|v Linear would deliver them; the scratch arena receives the rotated K rows
        qkv = torch.cat([q[b].transpose(0, 1).reshape(S, Hq * D), k[b].transpose(0, 1).reshape(S, Hkv * D),
                         torch.zeros(S, Hkv * D, dtype=q.dtype, device=q.device)], dim=1).contiguous()
        kv = ops.KvArena(1, 1, Hkv, (S + 31) // 32 * 32, q.device)
        pos = torch.arange(S, dtype=torch.int32, device=q.device)
        qr = ops.rope_kv_append(qkv, cs[b, :, :64].contiguous(), sn[b, :, :64].contiguous(), torch.zeros(S, dtype=torch.int32, device=q.device),
                                pos, kv, 0, Hq)
        q_out[b] = qr.view(S, Hq, D).transpose(0, 1)
        k_out[b] = kv.k_view(0, 0)[:, :S]
    return q_out, k_out


def apply_livecc_amd_kernel_to_qwen2_vl(rms_norm: bool = True, layer_norm: bool = True, swiglu: bool = True, rope: bool = True,
                                        attention: bool = True) -> None:
    """Call before constructing the HF model, like `apply_liger_kernel_to_qwen2_vl()`.  `attention=True` also registers the
    native attention function under `attn_implementation="livecc_amd"`."""
    import transformers.models.qwen2_vl.modeling_qwen2_vl as m
    for name in ("Qwen2VLRMSNorm", "LayerNorm", "Qwen2MLP", "apply_multimodal_rotary_pos_emb"):
        _SAVED.setdefault(name, getattr(m, name))

    if rms_norm:
        class LccRMSNorm(_SAVED["Qwen2VLRMSNorm"]):
            def forward(self, hidden_states):
                # inference plugins: the native ops carry no autograd formula (torch.ops.livecc_amd.* refuse a tensor that would record a
                # graph), so parameters and activations are detached -- under generate()'s no_grad() that is a no-op
                x = hidden_states.detach().contiguous()
                return ops.rmsnorm(x.view(-1, x.shape[-1]), self.weight.detach(), self.variance_epsilon).view_as(x)
        m.Qwen2VLRMSNorm = LccRMSNorm

    if layer_norm:
        class LccLayerNorm(torch.nn.LayerNorm):
            def forward(self, x):
                x = x.detach().contiguous()
                return ops.layernorm(x.view(-1, x.shape[-1]), self.weight.detach(), self.bias.detach(), self.eps).view_as(x)
        m.LayerNorm = LccLayerNorm

    if swiglu:
        class LccMLP(_SAVED["Qwen2MLP"]):
            def forward(self, x):
                g = self.gate_proj(x).detach().contiguous()
                u = self.up_proj(x).detach().contiguous()
                return self.down_proj(ops.swiglu(g.view(-1, g.shape[-1]), u.view(-1, u.shape[-1])).view_as(g))
        m.Qwen2MLP = LccMLP

    if rope:
        m.apply_multimodal_rotary_pos_emb = lcc_apply_multimodal_rotary_pos_emb

    if attention:
        register_attention()


def revert_livecc_amd_kernel_to_qwen2_vl() -> None:
    """Undo the rebinding (tests)."""
    import transformers.models.qwen2_vl.modeling_qwen2_vl as m
    for name, obj in _SAVED.items():
        setattr(m, name, obj)


# ------------------------------------------------------------------------------------------------------------------------------
# (3) KV cache: in-place append into the arena the attention kernels read
# ------------------------------------------------------------------------------------------------------------------------------
def _cache_classes():
    from transformers.cache_utils import Cache, CacheLayerMixin

    class NativeKVLayer(CacheLayerMixin):
        """One decoder layer of `NativeKVCache`: K row-major [Hkv][Lmax][128], V blocked-transposed [Lmax/32][128][32] inside the
        shared arena (the layout of csrc/attention.hip).  `update` writes the new rows only (57,344 B per token over 28 layers at
        7B instead of re-copying the whole cache)."""
        is_sliding = False
        is_compileable = False

        def __init__(self, owner, layer_idx: int):
            super().__init__()
            self.owner, self.layer_idx = owner, layer_idx
            self.length = 0
            self.is_initialized = True

        def lazy_initialization(self, key_states, value_states) -> None:
            self.is_initialized = True

        def update(self, key_states: torch.Tensor, value_states: torch.Tensor, *args, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
            kv = self.owner.arena
            B, Hkv, s, D = key_states.shape
            if B != 1 or Hkv != kv.n_kv_heads or D != 128 or key_states.dtype != torch.bfloat16 or not key_states.is_cuda:
                raise ValueError("NativeKVCache: bf16 GPU key/value states [1, Hkv, s, 128] expected")
            if self.length + s > kv.lmax:
                raise RuntimeError(f"NativeKVCache capacity {kv.lmax} exceeded")
            dev = key_states.device
            # rows k | v as the fused Linear would deliver them; identity rotation (x*1 + rot(x)*0 is exact): the kernel only appends
            rows = torch.cat([key_states[0].transpose(0, 1).reshape(s, Hkv * D), value_states[0].transpose(0, 1).reshape(s, Hkv * D)], dim=1).contiguous()
            ones, zeros = self.owner.identity_tables(s)
            pos = torch.arange(self.length, self.length + s, dtype=torch.int32, device=dev)
            ops.rope_kv_append(rows, ones, zeros, torch.zeros(s, dtype=torch.int32, device=dev), pos, kv, self.layer_idx, 0)
            self.length += s
            self.owner.last_update[self.layer_idx] = (s, self.length)
            return kv.k_view(0, self.layer_idx)[None, :, :self.length], kv.v_view(0, self.layer_idx)[None, :, :self.length]

        def get_seq_length(self) -> int:
            return self.length

        def get_mask_sizes(self, query_length: int) -> Tuple[int, int]:
            return self.length + query_length, 0

        def get_max_length(self) -> int:
            return self.owner.arena.lmax

        def get_max_cache_shape(self) -> int:
            return self.owner.arena.lmax

        def reset(self) -> None:
            self.length = 0

        def crop(self, max_length: int) -> None:
            """HF semantics (cache_utils.DynamicLayer.crop): keep `max_length` tokens; a negative value removes that many.  The
            arena rows past the new length are simply overwritten by the next `update`."""
            max_length = int(max_length)
            if max_length < 0:
                max_length = self.length + max_length
            if max_length < self.length:
                self.length = max(0, max_length)

        def reorder_cache(self, beam_idx) -> None:
            raise NotImplementedError("beam search is not part of the reference's path")

    class NativeKVCache(Cache):
        """`past_key_values=NativeKVCache(config, max_len, device)`: an HF `Cache` over ONE pre-allocated arena; the registered
        attention function finds it through the layer index of the calling module."""

        def __init__(self, config, max_cache_len: int, device):
            tc = config.get_text_config() if hasattr(config, "get_text_config") else config
            self.arena = ops.KvArena(1, tc.num_hidden_layers, tc.num_key_value_heads, (int(max_cache_len) + 31) // 32 * 32, device)
            self.last_update = {}
            self._ident = {}
            super().__init__(layers=[NativeKVLayer(self, i) for i in range(tc.num_hidden_layers)])
            _LIVE_CACHES[:] = [r for r in _LIVE_CACHES if r() is not None]      # weak references: a dropped cache frees its arena
            _LIVE_CACHES.append(weakref.ref(self))

        def identity_tables(self, s: int):
            if s not in self._ident:
                dev = self.arena.buf.device
                self._ident = {s: (torch.ones(s, 64, dtype=torch.bfloat16, device=dev), torch.zeros(s, 64, dtype=torch.bfloat16, device=dev))}
            return self._ident[s]

    return NativeKVLayer, NativeKVCache


_LIVE_CACHES: List["weakref.ref"] = []      # caches the attention function may be called under (newest last)


def NativeKVCache(config, max_cache_len: int, device):
    """Factory (the classes derive from HF's, imported lazily)."""
    return _cache_classes()[1](config, max_cache_len, device)


# ------------------------------------------------------------------------------------------------------------------------------
# (2) attention registry
# ------------------------------------------------------------------------------------------------------------------------------
def lcc_attention_forward(module, query, key, value, attention_mask=None, dropout: float = 0.0, scaling: Optional[float] = None,
                          is_causal: Optional[bool] = None, **kwargs):
    """`AttentionInterface` signature (HF:modeling_utils.py:5093-5131; call sites Q2VL:375-417 ViT, 537-552 LLM):
    query [B,H,S,D], key/value [B,Hkv,L,D] -> (out [B,S,H,D], None).  ViT (module.is_causal False, d = 80): full attention inside
    the slice it is called with.  LLM (d = 128): bottom-right causal GQA; with a `NativeKVCache` the kernels read the arena that
    `update()` has just appended to (the passed key/value tensors are views of it), otherwise key/value are copied into a scratch
    arena first.  The softmax scale is the modules' head_dim**-0.5 (what HF passes)."""
    if dropout:
        raise NotImplementedError("attention dropout is a training feature")
    B, H, S, D = query.shape
    if B != 1 or query.dtype != torch.bfloat16 or not query.is_cuda:
        raise ValueError("livecc_amd attention: bf16 GPU tensors with batch 1 expected (the reference is batch 1)")
    causal = getattr(module, "is_causal", True) if is_causal is None else is_causal
    dev = query.device
    if D == 80 and not causal:                                   # ---- ViT slice ----
        n = S
        E = H * D
        qkv = torch.cat([query[0].transpose(0, 1).reshape(n, E), key[0].transpose(0, 1).reshape(n, E),
                         value[0].transpose(0, 1).reshape(n, E)], dim=1).contiguous()
        ones = torch.ones(n, D // 2, dtype=torch.float32, device=dev)      # HF already rotated q and k: identity tables
        out = ops.vit_attention(qkv, ones, torch.zeros_like(ones), [(1, n, 1)], H)
        return out.view(1, n, H, D), None
    if D != 128 or not causal:
        raise NotImplementedError(f"livecc_amd attention: head_dim {D}, causal={causal}")
    Hkv, L = key.shape[1], key.shape[2]
    layer = getattr(module, "layer_idx", 0)
    live = [c for c in (r() for r in reversed(_LIVE_CACHES)) if c is not None]
    cache = next((c for c in live if c.last_update.get(layer, (None, None))[1] == L
                  and key.data_ptr() == c.arena.k_view(0, layer).data_ptr()), None)
    if cache is not None:
        kv, past = cache.arena, L - S
    else:                                                         # no native cache in use: stage K/V once
        kv = ops.KvArena(1, 1, Hkv, (L + 31) // 32 * 32, dev)
        rows = torch.cat([key[0].transpose(0, 1).reshape(L, Hkv * D), value[0].transpose(0, 1).reshape(L, Hkv * D)], dim=1).contiguous()
        ops.rope_kv_append(rows, torch.ones(L, 64, dtype=torch.bfloat16, device=dev), torch.zeros(L, 64, dtype=torch.bfloat16, device=dev),
                           torch.zeros(L, dtype=torch.int32, device=dev), torch.arange(L, dtype=torch.int32, device=dev), kv, 0, 0)
        layer, past = 0, L - S
    q2 = query[0].transpose(0, 1).reshape(S, H * D).contiguous()
    if S == 1:
        kv_len = torch.tensor([past], dtype=torch.int32, device=dev)
        ntile = (L + 31) // 32
        out = ops.attn_decode(q2, kv, layer, torch.zeros(1, dtype=torch.int32, device=dev), kv_len, H, max(1, min(64, (ntile + 3) // 4)))
    else:
        out = ops.attn_prefill(q2, kv, layer, [(0, S, past)], H)
    return out.view(1, S, H, D), None


def register_attention() -> str:
    from transformers import AttentionInterface
    # No AttentionMaskInterface entry on purpose: for an attention name without a registered mask function HF builds no mask
    # tensor at all (masking_utils._preprocess_mask_arguments) -- the kernels implement the causal structure themselves.
    AttentionInterface.register(ATTN_NAME, lcc_attention_forward)
    return ATTN_NAME


# ------------------------------------------------------------------------------------------------------------------------------
# (6) video reader backend
# ------------------------------------------------------------------------------------------------------------------------------
VIDEO_READER_BACKENDS = {}


def register_video_reader_backend() -> None:
    """ref video_process_patch.py:85-86: VIDEO_READER_BACKENDS['decord+'] = _read_video_decord_plus (on decoded videos here)."""
    from . import video as V
    VIDEO_READER_BACKENDS["decord+"] = V.read_video_decord_plus
    try:
        from qwen_vl_utils.vision_process import VIDEO_READER_BACKENDS as theirs      # absent offline; registered when present
        theirs["decord+"] = V.read_video_decord_plus
    except Exception:
        pass
