"""HF-shaped model surface over the native engine: what `LiveCCDemoInfer` consumes.

    model = LiveCCForConditionalGeneration.from_pretrained(path, torch_dtype="auto", device_map="cuda:0")
    out = model.generate(input_ids=..., pixel_values_videos=..., video_grid_thw=..., past_key_values=state_or_None,
                         return_dict_in_generate=True, do_sample=False, repetition_penalty=1.05,
                         logits_processor=[ThresholdLogitsProcessor(...)], max_new_tokens=16, pad_token_id=eos)
    out.sequences, out.past_key_values

mirrors ref demo/infer.py:43-50,165-175 / HF Qwen2VLForConditionalGeneration.generate.  Differences that are
design decisions, not omissions:
  * `past_key_values` is an opaque `StreamState` (device-resident KV arena slot + per-stream rope_delta) instead of
    a DynamicCache of torch tensors; it is fed back exactly like the reference does (infer.py:166,173).
  * per-stream `rope_deltas` (HF keeps it as module-global state, modeling_qwen2_vl.py:857, which corrupts
    interleaved streams).
  * the reference's `prepare_inputs_for_generation` patch (livecc_utils/generation_patch.py) is honoured by
    construction: video features are consumed whenever the un-cached suffix contains <|video_pad|> (line 37-39).
  * frames may be handed over as uint8 (`frames=`) so that normalise+patchify run on the GPU; the HF
    `pixel_values_videos` fp32 tensor is accepted too.
No CPU path: everything below needs the HIP library and a GPU, and raises otherwise.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import protocol
from .config import LiveCCConfig, get_config
from .engine import Engine, Sampling
from .weights import WeightArena, from_hf_model as _arena_from_hf, from_pretrained as _arena_from_pretrained


class StreamState:
    """Opaque per-stream generation state (the `past_key_values` handle)."""

    def __init__(self, model: "LiveCCForConditionalGeneration", slot: int):
        self.model, self.slot = model, slot
        self.rope_delta: Optional[int] = None
        self.released = False

    def get_seq_length(self) -> int:
        return self.model.engine.slot_length(self.slot)[0]

    def release(self) -> None:
        if not self.released:
            self.released = True
            self.model._free_slots.append(self.slot)

    def __bool__(self):  # `if past_key_values:` in ref demo/infer.py:281,286
        return True

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


@dataclasses.dataclass
class GenerateOutput:
    sequences: torch.Tensor
    past_key_values: StreamState
    logits: Optional[torch.Tensor] = None     # [steps, V] raw bf16 lm_head outputs (parity tests)
    scores: Optional[torch.Tensor] = None     # [V] processed scores of the last step


class _Cfg:
    """`.config` attribute subset the reference reads (infer.py:171, generation_patch.py:37)."""

    def __init__(self, c: LiveCCConfig):
        self.eos_token_id = c.eos_token_id
        self.video_token_id = c.video_token_id
        self.image_token_id = c.image_token_id
        self.vision_start_token_id = c.vision_start_token_id
        self.vision_end_token_id = c.vision_end_token_id
        self.vocab_size = c.vocab_size


class LiveCCForConditionalGeneration:
    main_input_name = "input_ids"

    def __init__(self, cfg: LiveCCConfig, weights: WeightArena, device, max_streams: int = 1, max_kv_len: int = 32768,
                 max_new_rows: int = 2048, max_patches: int = 8192, max_history: int = 64,
                 text_offset_rule: str = "hf5"):
        self.cfg, self.weights = cfg, weights
        self.config = _Cfg(cfg)
        self.device = torch.device(device)
        self.engine = Engine(cfg, weights, self.device, max_slots=max_streams, max_kv_len=max_kv_len,
                             max_new_rows=max_new_rows, max_patches=max_patches, max_history=max_history)
        self._free_slots = list(range(max_streams - 1, -1, -1))
        self.text_offset_rule = text_offset_rule
        self.prepare_inputs_for_generation = None   # assignable, as ref demo/infer.py:50 does
        self.generation_config: dict = {}            # from_pretrained fills it from generation_config.json (do_sample / top_k ...)

    # ---- constructors ----
    @classmethod
    def from_pretrained(cls, model_path: str, torch_dtype="auto", device_map=None, attn_implementation=None, **kw):
        """Same call as ref demo/infer.py:43-47.  bf16 only ('auto' resolves to bf16 for these checkpoints)."""
        if torch_dtype not in ("auto", torch.bfloat16, "bfloat16"):
            raise ValueError("the MI355X engine computes in bf16 (torch_dtype='auto' of the released checkpoints)")
        device = device_map if isinstance(device_map, (str, torch.device)) and device_map not in ("auto",) else "cuda"
        if "cuda" not in str(device):
            raise RuntimeError("livecc_amd has no CPU path; use device_map='cuda[:i]' (the oracle under oracle/ is the CPU path)")
        cfg = get_config(model_path)
        arena = _arena_from_pretrained(model_path, cfg, device, llm_fp8=bool(kw.pop("llm_fp8", False)))
        model = cls(cfg, arena, device, **kw)
        import json
        import os
        gc = os.path.join(model_path, "generation_config.json")
        if os.path.exists(gc):
            model.generation_config = json.load(open(gc))
        return model

    @classmethod
    def from_config(cls, cfg: LiveCCConfig, device="cuda", seed: int = 0, llm_fp8: bool = False, **kw):
        """llm_fp8: quantise the LLM Linear weights to OCP e4m3 with per-output-row scales (the 72B single-GPU weight path)."""
        return cls(cfg, WeightArena(cfg, device, llm_fp8=llm_fp8).fill_random(seed), device, **kw)

    @classmethod
    def from_hf_model(cls, hf_model, cfg: LiveCCConfig, device="cuda", llm_fp8: bool = False, **kw):
        return cls(cfg, _arena_from_hf(hf_model, cfg, device, llm_fp8=llm_fp8), device, **kw)

    def eval(self):
        return self

    # ---- stream slots ----
    def new_stream(self) -> StreamState:
        if not self._free_slots:
            raise RuntimeError(f"all {self.engine.max_slots} stream slots are in use (raise max_streams)")
        st = StreamState(self, self._free_slots.pop())
        self.engine.reset_slot(st.slot)
        return st

    # ---- generate ----
    @staticmethod
    def _threshold_params(logits_processor):
        if not logits_processor:
            return None
        if len(logits_processor) != 1:
            raise NotImplementedError("only the reference's ThresholdLogitsProcessor is supported as a custom processor")
        p = logits_processor[0]
        if not all(hasattr(p, a) for a in ("token_id", "base_threshold", "step")):
            raise NotImplementedError(f"unsupported logits processor {type(p).__name__}")
        return int(p.token_id), float(p.base_threshold), float(p.step)

    def _positions(self, st: StreamState, ids_new: np.ndarray, grids, past_len: int) -> np.ndarray:
        if past_len == 0:
            pos, delta = protocol.rope_index_first_turn(ids_new, grids, self.cfg, self.text_offset_rule)
            st.rope_delta = delta
            return pos
        if st.rope_delta is None:
            st.rope_delta = 0
        return protocol.positions_with_cache(past_len, len(ids_new), st.rope_delta)

    @torch.inference_mode()
    def generate(self, input_ids: torch.Tensor = None, pixel_values_videos: Optional[torch.Tensor] = None,
                 video_grid_thw: Optional[torch.Tensor] = None, mm_token_type_ids=None,
                 past_key_values: Optional[StreamState] = None, return_dict_in_generate: bool = True,
                 do_sample: bool = False, repetition_penalty: float = 1.0, logits_processor=None,
                 max_new_tokens: int = 16, min_new_tokens: Optional[int] = None, pad_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, frames: Optional[torch.Tensor] = None, frames_layout: str = "TCHW",
                 output_logits: bool = False, output_scores: bool = False, attention_mask=None, **unused):
        if do_sample:
            # The reference's live_cc defaults to do_sample=True (ref demo/infer.py:68) with the checkpoints' generation_config
            # (top_k = 1): the top-k warper leaves ONE finite score, so multinomial sampling is the argmax -- served by the
            # greedy sampler.  Any other sampling configuration is not implemented.
            top_k = unused.get("top_k", self.generation_config.get("top_k"))
            if top_k != 1:
                raise NotImplementedError("sampling is only supported in its degenerate top_k=1 form (the released "
                                          "generation_config); pass do_sample=False or top_k=1")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).all()):
            raise NotImplementedError("padding masks are not supported (the reference passes none, infer.py:156)")
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("generate() is single-stream like the reference (batch 1); use generate_batch for many streams")
        if min_new_tokens is not None and min_new_tokens not in (0, max_new_tokens):
            raise NotImplementedError("min_new_tokens must be None/0 or equal to max_new_tokens")
        if max_new_tokens < 1 or max_new_tokens > self.engine.max_history:
            raise ValueError(f"max_new_tokens must be in [1, {self.engine.max_history}] (engine max_history)")
        r = self.generate_batch([dict(
            input_ids=input_ids[0], pixel_values_videos=pixel_values_videos,
            video_grid_thw=video_grid_thw, frames=frames, frames_layout=frames_layout, state=past_key_values)],
            repetition_penalty=repetition_penalty, logits_processor=logits_processor, max_new_tokens=max_new_tokens,
            force_length=bool(min_new_tokens), eos_token_id=eos_token_id, output_logits=output_logits,
            output_scores=output_scores)[0]
        return r if return_dict_in_generate else r.sequences

    @torch.inference_mode()
    def generate_batch(self, requests: Sequence[dict], repetition_penalty: float = 1.0, logits_processor=None,
                       max_new_tokens: int = 16, force_length: bool = False, eos_token_id: Optional[int] = None,
                       output_logits: bool = False, output_scores: bool = False) -> List[GenerateOutput]:
        """Many streams, one call: the ViTs of all clips run as one batch, all prefills as one packed batch, and the
        decode steps advance every stream together (weights are streamed from HBM once per step for the whole batch).
        Each request: input_ids (1-D, full history like the reference's cat(past_ids, new_ids)), optional
        pixel_values_videos+video_grid_thw or uint8 frames, and `state` (StreamState or None)."""
        cfg, eng = self.cfg, self.engine
        eos = cfg.eos_token_id if eos_token_id is None else int(eos_token_id)
        thr = self._threshold_params(logits_processor)
        n = len(requests)
        states, ids_new, pos3, clips, slots = [], [], [], [], []
        full_ids = []
        for rq in requests:
            st = rq.get("state") or self.new_stream()
            ids_full = torch.as_tensor(rq["input_ids"]).reshape(-1).cpu().numpy().astype(np.int64)
            past_len = st.get_seq_length()
            if len(ids_full) <= past_len:
                raise ValueError(f"input_ids ({len(ids_full)}) must extend the cached sequence ({past_len})")
            new = ids_full[past_len:]
            grids = []
            n_vid = int((new == cfg.video_token_id).sum())
            if n_vid:
                if rq.get("frames") is not None:
                    f = rq["frames"]
                    lay = rq.get("frames_layout", "TCHW")
                    if not f.is_cuda:
                        f = f.to(self.device, non_blocking=True)
                    f = f.contiguous()
                    T, H, W = (f.shape[0], f.shape[2], f.shape[3]) if lay == "TCHW" else (f.shape[0], f.shape[1], f.shape[2])
                    grids = [protocol.grid_of(T, H, W, cfg)]
                    clips.append(dict(frames=f, layout=lay))
                elif rq.get("pixel_values_videos") is not None:
                    g = torch.as_tensor(rq["video_grid_thw"]).reshape(-1, 3).tolist()
                    pv = rq["pixel_values_videos"].to(self.device, dtype=torch.float32).contiguous()
                    off = 0
                    for t, h, w in g:
                        clips.append(dict(pixel_values=pv[off:off + t * h * w], grid=(t, h, w)))
                        off += t * h * w
                    grids = [tuple(x) for x in g]
                else:
                    raise ValueError("input_ids contain <|video_pad|> but no frames / pixel_values_videos were given")
                if sum(protocol.num_video_tokens(g, cfg) for g in grids) != n_vid:
                    raise ValueError(f"Video features and video tokens do not match, tokens: {n_vid}, features: "
                                     f"{sum(protocol.num_video_tokens(g, cfg) for g in grids)}")
            pos3.append(self._positions(st, new, grids, past_len))
            states.append(st); ids_new.append(new); slots.append(st.slot); full_ids.append(ids_full)
        vit = eng.vit_encode(clips) if clips else None
        V = cfg.vocab_size
        logits_buf = torch.empty(max_new_tokens, n, V, dtype=torch.bfloat16, device=self.device) if output_logits else None
        scores_buf = torch.empty(n, V, dtype=torch.float32, device=self.device) if output_scores else None
        sp = Sampling(repetition_penalty=repetition_penalty, eos_token=eos, suppress_eos=force_length,
                      thr_token=thr[0] if thr else -1, thr_base=thr[1] if thr else None, thr_step=thr[2] if thr else 0.0)
        eng.prefill(slots, ids_new, pos3, vit, sp, scores_out=scores_buf, logits_out=logits_buf)
        # decode: the device loop needs no host round trip per token (EOS freezes a slot on the device); long generations
        # (video_qa: max_new_tokens=512, ref demo/infer.py:236) are cut into chunks of 32 steps so that the host can stop
        # early once every stream has emitted EOS.
        chunk = max_new_tokens - 1 if force_length else 32
        for b0 in range(0, n, 16):   # decode batches of <= 16 streams (one MFMA column tile)
            grp = slots[b0:b0 + 16]
            step = 1
            while step < max_new_tokens:
                k = min(chunk, max_new_tokens - step)
                eng.decode(grp, k, step, sp, scores_out=scores_buf[b0:b0 + 16] if scores_buf is not None else None,
                           logits_out=logits_buf if (logits_buf is not None and n <= 16) else None)
                step += k
                if step < max_new_tokens and all(eng.generated_count(s) < step for s in grp):
                    break
        outs = []
        for b, st in enumerate(states):
            toks = eng.read_tokens(st.slot, max_new_tokens)
            if thr and logits_processor:
                logits_processor[0].count += len(toks)        # keep the stateful processor's counter as HF would
            seq = torch.from_numpy(np.concatenate([full_ids[b], np.asarray(toks, dtype=np.int64)])).view(1, -1).to(self.device)
            lg = logits_buf[:len(toks), b] if logits_buf is not None and n <= 16 else None
            outs.append(GenerateOutput(sequences=seq, past_key_values=st, logits=lg,
                                       scores=scores_buf[b] if scores_buf is not None else None))
        return outs

    # ---- ViT only (frames/s benchmarks, parity tests) ----
    @torch.inference_mode()
    def get_video_features(self, pixel_values_videos: torch.Tensor = None, video_grid_thw=None, frames=None,
                           frames_layout: str = "TCHW") -> torch.Tensor:
        if frames is not None:
            return self.engine.vit_encode([dict(frames=frames.to(self.device).contiguous(), layout=frames_layout)])
        g = torch.as_tensor(video_grid_thw).reshape(-1, 3).tolist()
        pv = pixel_values_videos.to(self.device, dtype=torch.float32).contiguous()
        clips, off = [], 0
        for t, h, w in g:
            clips.append(dict(pixel_values=pv[off:off + t * h * w], grid=(t, h, w)))
            off += t * h * w
        return self.engine.vit_encode(clips)
