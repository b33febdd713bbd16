"""Python owner of one native engine per GPU: allocates HBM through PyTorch, binds it to the C engine, and exposes
the three model-level calls (ViT encode, LLM prefill, LLM decode) on torch tensors.  No arithmetic here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .config import LiveCCConfig
from .weights import WeightArena

# HF image_processing_backends.py:307-311: mean*(1/rescale_factor), std*(1/rescale_factor) as fp32 tensors
_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def fused_mean_std() -> Tuple[np.ndarray, np.ndarray]:
    rescale_factor = 1 / 255
    mean = (torch.tensor(_CLIP_MEAN) * (1.0 / rescale_factor)).numpy().astype(np.float32)
    std = (torch.tensor(_CLIP_STD) * (1.0 / rescale_factor)).numpy().astype(np.float32)
    return mean, std


def vision_rope_tables(grids: Sequence[Sequence[int]], cfg: LiveCCConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 cos/sin [P, 40] of the ViT 2-D RoPE, computed on the host exactly as HF does
    (vision_utils.py:81-127 position ids block-major over 2x2 merge blocks; modeling_qwen2_vl.py:238-248
    VisionRotaryEmbedding with dim = head_dim//2, theta 1e4; 711-713 cos/sin of cat(freqs, freqs))."""
    m = cfg.spatial_merge_size
    dim = cfg.vit_head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    pos = []
    for t, h, w in grids:
        hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        shape = (h // m, m, w // m, m)
        hp = hp.reshape(shape).transpose(1, 2).flatten()
        wp = wp.reshape(shape).transpose(1, 2).flatten()
        pos.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    pos = torch.cat(pos, dim=0)
    freqs = (pos.unsqueeze(-1) * inv_freq).flatten(1)          # [P, 40]: 20 h-freqs | 20 w-freqs
    return freqs.cos().contiguous(), freqs.sin().contiguous()


class Sampling:
    def __init__(self, repetition_penalty: float = 1.0, eos_token: int = -1, suppress_eos: bool = False,
                 thr_token: int = -1, thr_base: Optional[float] = None, thr_step: float = 0.0, eos_token2: int = -1,
                 do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, seed: int = 0):
        self.repetition_penalty, self.eos_token, self.suppress_eos = repetition_penalty, eos_token, suppress_eos
        self.thr_token, self.thr_base, self.thr_step = thr_token, thr_base, thr_step
        self.eos_token2, self.do_sample, self.temperature, self.top_k, self.top_p, self.seed = eos_token2, do_sample, temperature, top_k, top_p, seed

    def to_c(self, scores_out=None, logits_out=None) -> _lib.Sampling:
        s = _lib.Sampling()
        s.repetition_penalty = float(self.repetition_penalty)
        s.thr_token = int(self.thr_token) if self.thr_base is not None else -1
        s.use_thr = 1 if self.thr_base is not None else 0
        s.thr_base = float(self.thr_base or 0.0)
        s.thr_step = float(self.thr_step or 0.0)
        s.eos_token = int(self.eos_token)
        s.suppress_eos = 1 if self.suppress_eos else 0
        s.scores_out = scores_out.data_ptr() if scores_out is not None else None
        s.logits_out = logits_out.data_ptr() if logits_out is not None else None
        s.eos_token2 = int(self.eos_token2)
        s.do_sample = 1 if self.do_sample else 0
        s.temperature = float(self.temperature)
        s.top_k = int(self.top_k or 0)
        s.top_p = float(self.top_p if self.top_p is not None else 1.0)
        s.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        return s


class Engine:
    """One engine = one GPU.  `max_slots` concurrent video streams, each with a contiguous KV arena of
    `max_kv_len` tokens (7B: 57,344 B/token -> 1.88 GB at 32k; 288 GB of HBM hold >100 such streams)."""

    def __init__(self, cfg: LiveCCConfig, weights: WeightArena, device=None, max_slots: int = 1, max_kv_len: int = 32768,
                 max_new_rows: int = 2048, max_patches: int = 8192, max_history: int = 64):
        cfg.validate()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LccError("livecc_amd.Engine needs a GPU (MI355X / gfx950): torch.cuda.is_available() is False; "
                                "there is no CPU fallback")
        self.cfg, self.weights = cfg, weights
        self.device = torch.device(device if device is not None else weights.device)
        assert self.device.type == "cuda"
        torch.cuda.set_device(self.device)
        max_kv_len = (max_kv_len + 31) // 32 * 32
        self.max_slots, self.max_kv_len, self.max_history = max_slots, max_kv_len, max_history
        self.max_new_rows, self.max_patches = max_new_rows, max_patches
        mc = _lib.ModelConfig(cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers,
                              cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.rms_norm_eps,
                              cfg.mrope_section[0], cfg.mrope_section[1], cfg.mrope_section[2], cfg.vit_depth,
                              cfg.vit_embed_dim, cfg.vit_num_heads, cfg.vit_mlp_dim, cfg.patch_dim, cfg.spatial_merge_size,
                              1 if getattr(weights, "llm_fp8", False) else 0)
        lim = _lib.EngineLimits(max_slots, max_kv_len, max_new_rows, max_patches, max_history)
        self.h = self.lib.lcc_engine_create(C.byref(mc), C.byref(lim))
        if not self.h:
            raise _lib.LccError("lcc_engine_create: " + self.lib.lcc_last_error().decode())
        lib, h = self.lib, self.h
        self._ws = torch.empty(lib.lcc_engine_workspace_bytes(h), dtype=torch.uint8, device=self.device)
        self._state = torch.zeros(lib.lcc_engine_state_bytes(h), dtype=torch.uint8, device=self.device)
        mb = lib.lcc_engine_meta_bytes(h)
        self._meta_dev = torch.empty(mb, dtype=torch.uint8, device=self.device)
        self._meta_host = torch.empty(mb, dtype=torch.uint8).pin_memory()
        _lib.check(lib.lcc_engine_bind_buffers(h, self._ws.data_ptr(), self._ws.numel(), self._state.data_ptr(),
                                               self._state.numel(), self._meta_dev.data_ptr(), self._meta_host.data_ptr(), mb),
                   "lcc_engine_bind_buffers")
        # private ViT workspace + meta ring: the vision tower of the NEXT turn's frames may then run on a second stream under the
        # current turn's decode steps (modeling.generate_batch(prefetch=...))
        self._ws_vit = torch.empty(lib.lcc_engine_vit_workspace_bytes(h), dtype=torch.uint8, device=self.device)
        vmb = lib.lcc_engine_vit_meta_bytes(h)
        self._vmeta_dev = torch.empty(vmb, dtype=torch.uint8, device=self.device)
        self._vmeta_host = torch.empty(vmb, dtype=torch.uint8).pin_memory()
        _lib.check(lib.lcc_engine_bind_vit_buffers(h, self._ws_vit.data_ptr(), self._ws_vit.numel(), self._vmeta_dev.data_ptr(),
                                                   self._vmeta_host.data_ptr(), vmb), "lcc_engine_bind_vit_buffers")
        kvb = lib.lcc_engine_kv_bytes_per_slot(h)
        self.kv_bytes_per_slot = kvb
        self._kv = [torch.zeros(kvb, dtype=torch.uint8, device=self.device) for _ in range(max_slots)]
        for s, t in enumerate(self._kv):
            _lib.check(lib.lcc_engine_bind_kv(h, s, t.data_ptr(), t.numel()), "lcc_engine_bind_kv")
        for name in weights.names():
            v = weights.view(name)
            _lib.check(lib.lcc_engine_set_weight(h, name.encode(), v.data_ptr(), v.numel()), f"set_weight {name}")
        _lib.check(lib.lcc_engine_set_weight(h, b"inv_freq", weights.inv_freq.data_ptr(), weights.inv_freq.numel()), "inv_freq")
        miss = C.create_string_buffer(512)
        if not lib.lcc_engine_weights_ready(h, miss, 512):
            raise _lib.LccError("missing weights: " + miss.value.decode())
        m, s = fused_mean_std()
        self._mean = (C.c_float * 3)(*m.tolist())
        self._std = (C.c_float * 3)(*s.tolist())
        self._rope_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.lcc_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset_slot(self, slot: int) -> None:
        _lib.check(self.lib.lcc_slot_reset(self.h, slot, self._stream()), "lcc_slot_reset")

    def slot_length(self, slot: int) -> Tuple[int, int]:
        a, b = C.c_int(), C.c_int()
        _lib.check(self.lib.lcc_slot_get_length(self.h, slot, C.byref(a), C.byref(b)), "lcc_slot_get_length")
        return a.value, b.value

    def set_slot_length(self, slot: int, kv_len: int, next_pos: int) -> None:
        _lib.check(self.lib.lcc_slot_set_length(self.h, slot, kv_len, next_pos, self._stream()), "lcc_slot_set_length")

    def _rope(self, grids) -> Tuple[torch.Tensor, torch.Tensor]:
        key = tuple(tuple(int(x) for x in g) for g in grids)
        if key not in self._rope_cache:
            c, s = vision_rope_tables(key, self.cfg)
            self._rope_cache[key] = (c.to(self.device), s.to(self.device))
        return self._rope_cache[key]

    def set_vit_grid_cap(self, cap: int) -> None:
        """Workgroup budget of the next vit_encode calls (lcc_engine_set_vit_grid_cap): > 0 = the tower occupies `cap` of the 256 CUs, 0 = all."""
        _lib.check(self.lib.lcc_engine_set_vit_grid_cap(self.h, int(cap)), "lcc_engine_set_vit_grid_cap")

    def vit_encode(self, clips: Sequence[dict], stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """clips: dicts with either {'frames': uint8 device tensor, 'layout': 'THWC'|'TCHW'} or
        {'pixel_values': fp32 device [P,1176], 'grid': (t,h,w)}.  Returns bf16 [sum P/4, hidden].  `stream`: launch on that stream
        (the ViT has its own workspace; the caller orders consumers and later ViT calls with events)."""
        cfg = self.cfg
        arr = (_lib.Clip * len(clips))()
        grids, keep = [], []
        for i, c in enumerate(clips):
            if "frames" in c:
                f = c["frames"]
                assert f.dtype == torch.uint8 and f.is_cuda and f.is_contiguous() and f.dim() == 4
                layout = c.get("layout", "THWC")
                if layout == "THWC":
                    T, H, W, ch = f.shape
                    lay = 0
                else:
                    T, ch, H, W = f.shape
                    lay = 1
                assert ch == 3
                arr[i] = _lib.Clip(f.data_ptr(), None, lay, T, H, W)
                grids.append(((T + 1) // 2, H // cfg.patch_size, W // cfg.patch_size))
                keep.append(f)
            else:
                pv = c["pixel_values"]
                t, h, w = (int(x) for x in c["grid"])
                assert pv.dtype == torch.float32 and pv.is_cuda and pv.is_contiguous()
                assert pv.shape == (t * h * w, cfg.patch_dim), f"pixel_values {tuple(pv.shape)} vs grid {(t, h, w)}"
                arr[i] = _lib.Clip(None, pv.data_ptr(), 0, t * 2, h * cfg.patch_size, w * cfg.patch_size)
                grids.append((t, h, w))
                keep.append(pv)
        P = sum(t * h * w for t, h, w in grids)
        cos, sin = self._rope(grids)
        if stream is None:
            out = torch.empty(P // 4, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
            st = self._stream()
        else:
            with torch.cuda.stream(stream):
                out = torch.empty(P // 4, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
            st = stream.cuda_stream
        _lib.check(self.lib.lcc_vit_encode(self.h, len(clips), arr, self._mean, self._std, cos.data_ptr(), sin.data_ptr(),
                                           out.data_ptr(), st), "lcc_vit_encode")
        return out

    def prefill(self, slots: Sequence[int], ids: Sequence[np.ndarray], pos3: Sequence[np.ndarray],
                vit_embeds: Optional[torch.Tensor], sampling: Sampling, scores_out=None, logits_out=None) -> None:
        """ids[b]: int array of the new tokens of stream b; pos3[b]: [3, n_new] M-RoPE positions.  <|video_pad|> rows take
        consecutive rows of `vit_embeds` (stream order)."""
        n = len(slots)
        ids_cat = np.concatenate([np.asarray(x, dtype=np.int32).reshape(-1) for x in ids])
        n_new = np.asarray([len(x) for x in ids], dtype=np.int32)
        S = int(n_new.sum())
        pos_cat = np.concatenate([np.asarray(p, dtype=np.int32).reshape(3, -1) for p in pos3], axis=1)
        assert pos_cat.shape == (3, S)
        pos_cat = np.ascontiguousarray(pos_cat)
        is_vid = (ids_cat == self.cfg.video_token_id) | (ids_cat == self.cfg.image_token_id)
        vit_index = np.full(S, -1, dtype=np.int32)
        nv = int(is_vid.sum())
        vit_index[is_vid] = np.arange(nv, dtype=np.int32)
        if nv:
            # HF get_placeholder_mask check (modeling_qwen2_vl.py:1069-1090)
            if vit_embeds is None or vit_embeds.shape[0] != nv:
                raise ValueError(f"Video features and video tokens do not match, tokens: {nv}, features: "
                                 f"{0 if vit_embeds is None else vit_embeds.shape[0]}")
        slots_a = np.asarray(slots, dtype=np.int32)
        sp = sampling.to_c(scores_out, logits_out)
        _lib.check(self.lib.lcc_llm_prefill(
            self.h, n, slots_a.ctypes.data, n_new.ctypes.data, ids_cat.ctypes.data, vit_index.ctypes.data,
            vit_embeds.data_ptr() if vit_embeds is not None else None, pos_cat.ctypes.data, C.byref(sp), self._stream()),
            "lcc_llm_prefill")

    def decode(self, slots: Sequence[int], n_steps: int, first_step_index: int, sampling: Sampling, scores_out=None,
               logits_out=None) -> None:
        slots_a = np.asarray(slots, dtype=np.int32)
        sp = sampling.to_c(scores_out, logits_out)
        _lib.check(self.lib.lcc_llm_decode(self.h, len(slots), slots_a.ctypes.data, n_steps, first_step_index, C.byref(sp),
                                           self._stream()), "lcc_llm_decode")

    # ---- parity instrumentation (tests only): residual-stream taps / per-layer input overrides ----
    def set_llm_taps(self, max_rows: int, overrides: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Bind (max_rows > 0) or unbind (0) the LLM taps; returns the tap tensor bf16 [2L+1, max_rows, H] that the following
        prefill / decode calls fill.  `overrides`: bf16 [L + 1, max_rows, H] = the input of every layer and of the final norm (prefill only)."""
        cfg = self.cfg
        if max_rows <= 0:
            _lib.check(self.lib.lcc_debug_set_llm_taps(self.h, None, None, 0), "lcc_debug_set_llm_taps")
            self._llm_taps = self._llm_over = None
            return None
        taps = torch.zeros(2 * cfg.num_hidden_layers + 1, max_rows, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        if overrides is not None:
            assert overrides.shape == (cfg.num_hidden_layers + 1, max_rows, cfg.hidden_size) and overrides.dtype == torch.bfloat16
            overrides = overrides.to(self.device).contiguous()
        self._llm_taps, self._llm_over = taps, overrides
        _lib.check(self.lib.lcc_debug_set_llm_taps(self.h, taps.data_ptr(), overrides.data_ptr() if overrides is not None else None,
                                                   max_rows), "lcc_debug_set_llm_taps")
        return taps

    def set_vit_taps(self, max_rows: int, overrides: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """The same for the vision tower: taps bf16 [depth+1, max_rows, E] (0 = PatchEmbed output), overrides [depth+1, max_rows, E] (last = PatchMerger input)."""
        cfg = self.cfg
        if max_rows <= 0:
            _lib.check(self.lib.lcc_debug_set_vit_taps(self.h, None, None, 0), "lcc_debug_set_vit_taps")
            self._vit_taps = self._vit_over = None
            return None
        taps = torch.zeros(cfg.vit_depth + 1, max_rows, cfg.vit_embed_dim, dtype=torch.bfloat16, device=self.device)
        if overrides is not None:
            assert overrides.shape == (cfg.vit_depth + 1, max_rows, cfg.vit_embed_dim) and overrides.dtype == torch.bfloat16
            overrides = overrides.to(self.device).contiguous()
        self._vit_taps, self._vit_over = taps, overrides
        _lib.check(self.lib.lcc_debug_set_vit_taps(self.h, taps.data_ptr(), overrides.data_ptr() if overrides is not None else None,
                                                   max_rows), "lcc_debug_set_vit_taps")
        return taps

    def set_forced_tokens(self, tokens: Optional[np.ndarray]) -> None:
        """Teacher forcing (tests): tokens int [n_steps, n_streams] replace what the sampler chooses at generate-step k of the following
        calls with n_streams streams (lcc_debug_set_forced_tokens); None unbinds."""
        if tokens is None:
            _lib.check(self.lib.lcc_debug_set_forced_tokens(self.h, None, 0, 0), "lcc_debug_set_forced_tokens")
            self._forced = None
            return
        t = torch.as_tensor(np.ascontiguousarray(tokens, dtype=np.int32)).to(self.device)
        assert t.dim() == 2
        self._forced = t                      # keeps the device buffer alive while bound
        _lib.check(self.lib.lcc_debug_set_forced_tokens(self.h, t.data_ptr(), int(t.shape[0]), int(t.shape[1])), "lcc_debug_set_forced_tokens")

    def profile(self, enable: bool, max_samples: int = 4096) -> None:
        _lib.check(self.lib.lcc_engine_profile(self.h, 1 if enable else 0, max_samples), "lcc_engine_profile")

    def profile_read(self, max_n: int = 4096) -> np.ndarray:
        buf = np.zeros(max_n, dtype=np.float32)
        n = C.c_int()
        _lib.check(self.lib.lcc_engine_profile_read(self.h, buf.ctypes.data, max_n, C.byref(n)), "lcc_engine_profile_read")
        return buf[:n.value]

    def profile_read_steps(self, max_n: int = 4096) -> np.ndarray:
        """Milliseconds of every whole decode step recorded while profiling was on."""
        buf = np.zeros(max_n, dtype=np.float32)
        n = C.c_int()
        _lib.check(self.lib.lcc_engine_profile_read_steps(self.h, buf.ctypes.data, max_n, C.byref(n)), "lcc_engine_profile_read_steps")
        return buf[:n.value]

    def profile_read_step_index(self, max_n: int = 4096) -> np.ndarray:
        """For the samples of `profile_read_steps`: the index of each step inside its decode call (0 = right after the prefill)."""
        buf = np.zeros(max_n, dtype=np.int32)
        n = C.c_int()
        _lib.check(self.lib.lcc_engine_profile_read_step_index(self.h, buf.ctypes.data, max_n, C.byref(n)), "lcc_engine_profile_read_step_index")
        return buf[:n.value]

    def generated_count(self, slot: int) -> int:
        """Blocking: number of tokens the slot has generated in the current generate call (stops growing after EOS)."""
        buf = np.zeros(1, dtype=np.int32)
        n = C.c_int()
        _lib.check(self.lib.lcc_slot_read_tokens(self.h, slot, buf.ctypes.data, 0, C.byref(n), self._stream()), "lcc_slot_read_tokens")
        return n.value

    def read_tokens(self, slot: int, max_n: int) -> List[int]:
        buf = np.zeros(max(max_n, 1), dtype=np.int32)
        n = C.c_int()
        _lib.check(self.lib.lcc_slot_read_tokens(self.h, slot, buf.ctypes.data, max_n, C.byref(n), self._stream()),
                   "lcc_slot_read_tokens")
        return buf[:min(n.value, max_n)].tolist()
