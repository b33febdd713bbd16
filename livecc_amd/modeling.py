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
  * ONE engine lock per model (round 6): `generate` / `generate_batch` / `get_video_features` / `new_stream` /
    `StreamState.release` serialise on `model._lock` (re-entrant).  The engine has one activation workspace, one meta ring
    and one ViT workspace; the reference is called with `default_concurrency_limit=5` over one model object
    (ref demo/app.py:178) and races on `rope_deltas` -- here concurrent callers with their own stream states are safe and
    produce the tokens of the serial run (tests/test_gpu_facade.py).
No CPU path: everything below needs the HIP library and a GPU, and raises otherwise.
"""
from __future__ import annotations

import contextlib
import os
import threading

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
        lock = getattr(self.model, "_lock", None)          # (stand-in models of the CPU tests have no engine lock)
        if lock is None:
            lock = contextlib.nullcontext()
        with lock:
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


def read_generation_config(model_path: str):
    """(generation_config.json as a dict, its EOS id list or None): what HF's `from_pretrained` attaches to the model and what
    `generate` falls back to (do_sample / top_k / top_p / temperature / eos_token_id)."""
    import json
    import os
    gen_cfg = {}
    gc = os.path.join(model_path, "generation_config.json")
    if os.path.exists(gc):
        with open(gc) as f:
            gen_cfg = json.load(f)
    eos = gen_cfg.get("eos_token_id")
    eos_ids = None if eos is None else (list(eos) if isinstance(eos, (list, tuple)) else [eos])
    return gen_cfg, eos_ids


class LiveCCForConditionalGeneration:
    main_input_name = "input_ids"
    DECODE_GROUP = 64          # LCC_MAX_DECODE_BATCH of include/livecc_amd.h: streams decoded per weight pass

    def __init__(self, cfg: LiveCCConfig, weights: WeightArena, device, max_streams: int = 1, max_kv_len: Optional[int] = None,
                 max_new_rows: int = 4096, max_patches: int = 16384, max_history: int = 512,
                 text_offset_rule: str = "hf4", eos_token_ids: Optional[Sequence[int]] = None):
        """Capacity knobs (all served out of the box with the defaults, like the reference's flows):
        max_history     = the largest max_new_tokens of one generate call (video_qa uses 512, ref demo/infer.py:236);
        max_new_rows    = rows of ONE prefill launch sequence -- longer prompts (the one-shot 24k-token video_qa / MCQ prefill)
                          are cut into pieces of at most this many rows over the carried KV (bit-identical to one pass);
        max_patches     = ViT patches of ONE encode launch sequence -- longer clips are encoded in groups of temporal slices
                          (slices never attend to each other, Q2VL vision_utils.py:60-65);
        max_kv_len      = KV capacity per stream; default = the trained 32k window + max_history of generation headroom.
        text_offset_rule: "hf4" = transformers-4.5x M-RoPE text offset after a vision block (what the released checkpoints
        were trained with, ref README.md:30); "hf5" = the installed 5.15 oracle's rule.  Identical for every streaming chunk;
        they differ for one-shot long clips (grid_t > max(h,w)/2).
        PINNING STATUS: "hf5" is pinned to the EXECUTED transformers 5.15 (`get_rope_index`, tests/test_protocol.py + every one-shot GPU
        fixture).  "hf4" is UNPINNED: it restates the 4.5x algorithm (`st_idx = llm_pos_ids_list[-1].max() + 1`) from memory of that
        source -- no 4.5x wheel / sdist exists offline (searched the image and the pip cache, round 6) -- and is checked only against
        that restatement (tests/test_protocol.py:137,177).  bench.py says so in `config.workload` whenever the hf4 rule is what ran."""
        self.cfg, self.weights = cfg, weights
        self.config = _Cfg(cfg)
        self.device = torch.device(device)
        if max_kv_len is None:
            max_kv_len = cfg.max_position_embeddings + max_history
        self.engine = Engine(cfg, weights, self.device, max_slots=max_streams, max_kv_len=max_kv_len,
                             max_new_rows=max_new_rows, max_patches=max_patches, max_history=max_history)
        self._free_slots = list(range(max_streams - 1, -1, -1))
        self._lock = threading.RLock()          # every engine entry point below holds it for the whole call (see the module docstring)
        self.text_offset_rule = text_offset_rule
        self.prepare_inputs_for_generation = None   # assignable, as ref demo/infer.py:50 does
        self.generation_config: dict = {}            # from_pretrained fills it from generation_config.json (do_sample / top_k ...)
        # HF stops on ANY id of generation_config.eos_token_id ([<|im_end|>, <|endoftext|>] in the released checkpoints)
        ids = [int(cfg.eos_token_id)] + [int(t) for t in (eos_token_ids or []) if int(t) != int(cfg.eos_token_id)]
        if len(ids) > 2:
            raise NotImplementedError(f"at most two EOS ids are supported, got {ids}")
        self.eos_token_ids = ids
        self.config.eos_token_id = ids[0] if len(ids) == 1 else list(ids)
        # vision-tower prefetch (generate_batch(prefetch=...)): side stream, results keyed by the clip's storage
        self._side: Optional[torch.cuda.Stream] = None
        self._vit_cache: dict = {}
        self._vit_last_event: Optional[torch.cuda.Event] = None
        # CUs a prefetched vision tower may occupy while it runs under another turn's decode steps: -1 (default) = 128 for >= 4 clips (measured
        # +1.5 % tokens/s at 8 streams, profiles/r05/prefetch_grid_cap_ab.txt; one stream's tower is too small to share: 276 -> 256-269 tokens/s
        # under any cap), 0 = whole chip, n = n workgroups.  LCC_VIT_PREFETCH_CAP overrides.
        self.prefetch_grid_cap = int(os.environ.get("LCC_VIT_PREFETCH_CAP", "-1"))

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
        gen_cfg, eos_ids = read_generation_config(model_path)
        if eos_ids is not None and "eos_token_ids" not in kw:
            kw["eos_token_ids"] = eos_ids
        arena = _arena_from_pretrained(model_path, cfg, device, llm_fp8=bool(kw.pop("llm_fp8", False)))
        model = cls(cfg, arena, device, **kw)
        model.generation_config = gen_cfg
        return model

    @classmethod
    def from_config(cls, cfg: LiveCCConfig, device="cuda", seed: int = 0, llm_fp8: bool = False, **kw):
        """llm_fp8: quantise the LLM Linear weights to OCP e4m3 with per-output-row scales (the 72B single-GPU weight path)."""
        return cls(cfg, WeightArena(cfg, device, llm_fp8=llm_fp8).fill_random(seed), device, **kw)

    @classmethod
    def from_hf_model(cls, hf_model, cfg: LiveCCConfig, device="cuda", llm_fp8: bool = False, **kw):
        """Weights of an instantiated HF model (the oracle-parity constructor): positions follow the INSTALLED transformers
        (5.x text-offset rule) unless told otherwise, so that one-shot long clips compare equal to that oracle."""
        kw.setdefault("text_offset_rule", "hf5")
        return cls(cfg, _arena_from_hf(hf_model, cfg, device, llm_fp8=llm_fp8), device, **kw)

    def eval(self):
        return self

    # ---- stream slots ----
    def new_stream(self) -> StreamState:
        with self._lock:
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
        # a reused object continues from its `count` (ref demo/infer.py:17: threshold = base + step * count)
        return int(p.token_id), float(p.base_threshold) + float(p.step) * int(getattr(p, "count", 0)), float(p.step)

    def _positions(self, st: StreamState, ids_new: np.ndarray, grids, past_len: int) -> np.ndarray:
        if past_len == 0:
            pos, delta = protocol.rope_index_first_turn(ids_new, grids, self.cfg, self.text_offset_rule)
            st.rope_delta = delta
            return pos
        if st.rope_delta is None:
            st.rope_delta = 0
        return protocol.positions_with_cache(past_len, len(ids_new), st.rope_delta)

    def _resolve_sampling(self, do_sample, kw) -> dict:
        """HF semantics (generation/utils.py `_prepare_generation_config`): explicit arguments override the checkpoint's
        generation_config.json, which overrides HF's defaults (do_sample False, temperature 1.0, top_k 50, top_p 1.0).
        The reference's `live_cc` passes do_sample=True (ref demo/infer.py:68); `video_qa` / `live_cc_once_for_evaluation` pass
        nothing, i.e. the checkpoint decides (ref :236-241, 297-302)."""
        g = self.generation_config or {}
        if do_sample is None:
            do_sample = bool(g.get("do_sample", False))
        if not do_sample:
            return dict(do_sample=False)
        def pick(name, default):
            v = kw.get(name, g.get(name, default))
            return default if v is None and name != "top_k" else v
        temperature, top_k, top_p = pick("temperature", 1.0), pick("top_k", 50), pick("top_p", 1.0)
        top_k = 0 if top_k is None else int(top_k)
        if not (float(temperature) > 0):
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
        if top_k < 0 or not (0 < float(top_p) <= 1.0):
            raise ValueError(f"top_k must be >= 0 and top_p in (0, 1], got {top_k}, {top_p}")
        seed = kw.get("seed", g.get("seed"))
        return dict(do_sample=True, temperature=float(temperature), top_k=top_k, top_p=float(top_p), seed=None if seed is None else int(seed))

    @torch.inference_mode()
    def generate(self, input_ids: torch.Tensor = None, pixel_values_videos: Optional[torch.Tensor] = None,
                 video_grid_thw: Optional[torch.Tensor] = None, mm_token_type_ids=None,
                 past_key_values: Optional[StreamState] = None, return_dict_in_generate: bool = True,
                 do_sample: Optional[bool] = None, repetition_penalty: float = 1.0, logits_processor=None,
                 max_new_tokens: int = 16, min_new_tokens: Optional[int] = None, pad_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, frames: Optional[torch.Tensor] = None, frames_layout: str = "TCHW",
                 output_logits: bool = False, output_scores: bool = False, attention_mask=None, teacher_tokens=None, **unused):
        """`teacher_tokens` (tests): the generated tokens are forced to this sequence (what the HF oracle does with a forcing
        LogitsProcessor) while `output_logits` keeps the model's own raw logits of every step."""
        sampling_kw = self._resolve_sampling(do_sample, unused)
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).all()):
            raise NotImplementedError("padding masks are not supported (the reference passes none, infer.py:156)")
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("generate() is single-stream like the reference (batch 1); use generate_batch for many streams")
        if min_new_tokens is not None and min_new_tokens not in (0, max_new_tokens):
            raise NotImplementedError("min_new_tokens must be None/0 or equal to max_new_tokens")
        r = self.generate_batch([dict(
            input_ids=input_ids[0], pixel_values_videos=pixel_values_videos,
            video_grid_thw=video_grid_thw, frames=frames, frames_layout=frames_layout, state=past_key_values)],
            repetition_penalty=repetition_penalty, logits_processor=logits_processor, max_new_tokens=max_new_tokens,
            force_length=bool(min_new_tokens), eos_token_id=eos_token_id, output_logits=output_logits,
            output_scores=output_scores, teacher_tokens=None if teacher_tokens is None else [list(teacher_tokens)], **sampling_kw)[0]
        return r if return_dict_in_generate else r.sequences

    @torch.inference_mode()
    def generate_batch(self, requests: Sequence[dict], repetition_penalty: float = 1.0, logits_processor=None,
                       max_new_tokens: int = 16, force_length: bool = False, eos_token_id=None,
                       output_logits: bool = False, output_scores: bool = False, do_sample: bool = False, temperature: float = 1.0,
                       top_k: int = 0, top_p: float = 1.0, seed: Optional[int] = None, prefetch: Optional[Sequence[dict]] = None,
                       teacher_tokens: Optional[Sequence[Sequence[int]]] = None) -> List[GenerateOutput]:
        """Many streams, one call: the ViTs of all clips run as one batch, all prefills as one packed batch, and the
        decode steps advance every stream together (weights are streamed from HBM once per step for the whole batch).
        Each request: input_ids (1-D, full history like the reference's cat(past_ids, new_ids)), optional
        pixel_values_videos+video_grid_thw or uint8 frames, and `state` (StreamState or None).
        `prefetch`: clips ({'frames': uint8 GPU tensor, 'frames_layout': ...}) that a LATER call will pass as `frames`: their vision
        tower (compute-bound) is launched on a low-priority side stream right after this call's prefill and runs under its decode
        steps (HBM-bound weight streaming); the later call picks the embeddings up by the clip's storage (pointer, shape, layout).
        The frames must not be modified in between.  Results are bit-identical to the un-prefetched call.
        Thread-safe: the whole call (ViT, prefill, every decode step, the token read-back) runs under the model's engine lock."""
        with self._lock:
            created: List[StreamState] = []          # stream slots this call allocated for requests without a state
            try:
                return self._generate_batch_locked(requests, repetition_penalty, logits_processor, max_new_tokens, force_length, eos_token_id,
                                                   output_logits, output_scores, do_sample, temperature, top_k, top_p, seed, prefetch, teacher_tokens,
                                                   created)
            except BaseException:
                # a failed call gives its fresh slots back AT ONCE: the exception's traceback keeps this frame's StreamState objects alive for as
                # long as the caller holds the exception, and a caller that retries (StreamServer's per-stream isolation) needs the slots now
                for st in created:
                    st.release()
                raise

    def _generate_batch_locked(self, requests, repetition_penalty, logits_processor, max_new_tokens, force_length, eos_token_id, output_logits,
                               output_scores, do_sample, temperature, top_k, top_p, seed, prefetch, teacher_tokens, created) -> List[GenerateOutput]:
        cfg, eng = self.cfg, self.engine
        if eos_token_id is None:
            eos_ids = list(self.eos_token_ids)
        else:
            eos_ids = [int(t) for t in (eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id])]
            if len(eos_ids) > 2:
                raise NotImplementedError("at most two EOS ids are supported")
        eos, eos2 = eos_ids[0], (eos_ids[1] if len(eos_ids) > 1 else -1)
        if max_new_tokens < 1 or max_new_tokens > eng.max_history:
            raise ValueError(f"max_new_tokens={max_new_tokens} exceeds this model's max_history={eng.max_history}: construct the model "
                             f"with max_history >= the largest max_new_tokens you generate (video_qa uses 512)")
        thr = self._threshold_params(logits_processor)
        if seed is None:
            # HF draws from torch's global generator: `torch.manual_seed(s)` before a call reproduces its draws, two calls never repeat
            # them, and ranks seeded differently diverge.  The kernel's Philox stream is keyed by (seed, slot, per-slot draw counter); the
            # per-call seed is therefore DRAWN from torch's default generator -- only when the call really samples, so that
            # greedy calls leave the generator untouched like HF's.  Explicit `seed=` keeps a call reproducible by itself.
            if do_sample and top_k != 1:
                seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
            else:
                seed = 0
        n = len(requests)
        states, ids_new, pos3, clips, slots = [], [], [], [], []
        full_ids = []
        for rq in requests:
            st = rq.get("state")
            if st is None:
                st = self.new_stream()
                created.append(st)
            if st.released or st.model is not self:
                raise ValueError("past_key_values: this stream state was released (its KV slot may serve another stream) or belongs to "
                                 "another model")
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
                    clips.append(dict(frames=f, layout=lay, key=self._clip_key(f, lay)))
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
        vit = self._vit_for(clips) if clips else None
        V = cfg.vocab_size
        logits_buf = torch.empty(max_new_tokens, n, V, dtype=torch.bfloat16, device=self.device) if output_logits else None
        scores_buf = torch.empty(n, V, dtype=torch.float32, device=self.device) if output_scores else None
        sp = Sampling(repetition_penalty=repetition_penalty, eos_token=eos, suppress_eos=force_length,
                      thr_token=thr[0] if thr else -1, thr_base=thr[1] if thr else None, thr_step=thr[2] if thr else 0.0,
                      eos_token2=eos2, do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p, seed=seed)
        if teacher_tokens is not None:          # tests: [n streams][max_new_tokens] -> device table [step][stream]
            tt = np.asarray(teacher_tokens, dtype=np.int32)
            if tt.shape != (n, max_new_tokens) or (n > 1 and sum(len(x) for x in ids_new) > eng.max_new_rows) or n > self.DECODE_GROUP:
                raise ValueError("teacher_tokens must be [n_streams][max_new_tokens]; several streams must fit one prefill / one decode group")
            eng.set_forced_tokens(np.ascontiguousarray(tt.T))
        try:
            return self._generate_tail(n, states, slots, ids_new, pos3, vit, sp, scores_buf, logits_buf, prefetch, max_new_tokens, force_length,
                                       thr, logits_processor, full_ids)
        finally:
            if teacher_tokens is not None:
                eng.set_forced_tokens(None)

    def _generate_tail(self, n, states, slots, ids_new, pos3, vit, sp, scores_buf, logits_buf, prefetch, max_new_tokens, force_length, thr,
                       logits_processor, full_ids) -> List[GenerateOutput]:
        eng = self.engine
        self._prefill(slots, ids_new, pos3, vit, sp, scores_buf, logits_buf)
        if prefetch:
            self._prefetch_vit(prefetch)
        # decode: the device loop needs no host round trip per token (EOS freezes a slot on the device); long generations
        # (video_qa: max_new_tokens=512, ref demo/infer.py:236) are cut into chunks of 32 steps so that the host can stop
        # early once every stream has emitted EOS.
        chunk = max_new_tokens - 1 if force_length else 32
        G = self.DECODE_GROUP
        for b0 in range(0, n, G):    # one weight pass per step for up to 64 streams (LCC_MAX_DECODE_BATCH)
            grp = slots[b0:b0 + G]
            step = 1
            while step < max_new_tokens:
                k = min(chunk, max_new_tokens - step)
                eng.decode(grp, k, step, sp, scores_out=scores_buf[b0:b0 + G] if scores_buf is not None else None,
                           logits_out=logits_buf if (logits_buf is not None and n <= G) else None)
                step += k
                if step < max_new_tokens and all(eng.generated_count(s) < step for s in grp):
                    break
        outs = []
        for b, st in enumerate(states):
            toks = eng.read_tokens(st.slot, max_new_tokens)
            if thr and logits_processor:
                logits_processor[0].count += len(toks)        # keep the stateful processor's counter as HF would
            seq = torch.from_numpy(np.concatenate([full_ids[b], np.asarray(toks, dtype=np.int64)])).view(1, -1).to(self.device)
            lg = logits_buf[:len(toks), b] if logits_buf is not None and n <= G else None
            outs.append(GenerateOutput(sequences=seq, past_key_values=st, logits=lg,
                                       scores=scores_buf[b] if scores_buf is not None else None))
        return outs

    # ---- vision-tower prefetch on a side stream ----
    @staticmethod
    def _clip_key(f: torch.Tensor, lay: str):
        return (f.data_ptr(), tuple(f.shape), lay)

    def _side_stream(self) -> torch.cuda.Stream:
        if self._side is None:
            # (A CU-masked stream -- hipExtStreamCreateWithCUMask, 48..176 CUs for the ViT -- was measured much slower: 150-182
            # tokens/s instead of 256: the decode kernels on the main stream slowed to 150 us per layer for every mask size.)
            prio = 0
            try:   # numerically greatest = lowest priority: decode kernels win the arbitration for CUs
                prio = max(torch.cuda.Stream.priority_range())
            except Exception:
                pass
            self._side = torch.cuda.Stream(device=self.device, priority=prio)
        return self._side

    def _prefetch_vit(self, prefetch: Sequence[dict]) -> None:
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream()
        todo = []
        for c in prefetch:
            f, lay = c["frames"], c.get("frames_layout", c.get("layout", "TCHW"))
            if not f.is_cuda or not f.is_contiguous():
                continue
            k = self._clip_key(f, lay)
            if k not in self._vit_cache:
                todo.append((k, dict(frames=f, layout=lay)))
        if not todo:
            return
        side.wait_stream(main)      # the ViT workspace is free and this turn's prefill is enqueued: start under the decode steps
        cfg = self.cfg
        with torch.cuda.stream(side):
            # under the decode steps the tower keeps to `prefetch_grid_cap` CUs (engine.set_vit_grid_cap): spread over all of them it takes
            # the decode kernels' occupancy away and adds its whole duration to the steps it was meant to hide under
            eng_cap = self.prefetch_grid_cap if self.prefetch_grid_cap >= 0 else (128 if len(todo) >= 4 else 0)
            if eng_cap:
                self.engine.set_vit_grid_cap(eng_cap)
            try:
                emb = self._vit_encode([c for _, c in todo], stream=side)     # ONE batched launch sequence (large-M GEMMs), sliced per clip
            finally:
                if eng_cap:
                    self.engine.set_vit_grid_cap(0)
            ev = self._vit_last_event                                     # recorded by _vit_encode after the last ViT kernel
            off = 0
            for k, clip in todo:
                f, lay = clip["frames"], clip["layout"]
                H, W = (f.shape[1], f.shape[2]) if lay == "THWC" else (f.shape[2], f.shape[3])
                n = ((f.shape[0] + 1) // 2) * (H // cfg.patch_size) * (W // cfg.patch_size) // 4
                self._vit_cache[k] = (emb[off:off + n], ev, (clip["frames"], emb))
                off += n
        while len(self._vit_cache) > 64:        # unclaimed prefetches do not accumulate
            self._vit_cache.pop(next(iter(self._vit_cache)))

    def _vit_for(self, clips: Sequence[dict]) -> torch.Tensor:
        """Embeddings of this call's clips in request order: prefetched ones are claimed (after their event), the rest is encoded now."""
        main = torch.cuda.current_stream(self.device)
        hits = {i: self._vit_cache.pop(c["key"]) for i, c in enumerate(clips) if c.get("key") in self._vit_cache}
        miss = [i for i in range(len(clips)) if i not in hits]
        if not hits:
            return self._vit_encode(clips)
        parts: dict = {}
        for i, (emb, ev, keep) in hits.items():
            main.wait_event(ev)
            keep[1].record_stream(main)       # the batched embedding tensor the slice belongs to
            parts[i] = emb
        if miss:
            cfg = self.cfg
            got = self._vit_encode([clips[i] for i in miss])
            off = 0
            for i in miss:
                c = clips[i]
                if "frames" in c:
                    f, lay = c["frames"], c.get("layout", "THWC")
                    H, W = (f.shape[1], f.shape[2]) if lay == "THWC" else (f.shape[2], f.shape[3])
                    n = ((f.shape[0] + 1) // 2) * (H // cfg.patch_size) * (W // cfg.patch_size) // 4
                else:
                    n = c["grid"][0] * c["grid"][1] * c["grid"][2] // 4
                parts[i] = got[off:off + n]
                off += n
        return parts[0] if len(parts) == 1 else torch.cat([parts[i] for i in range(len(clips))], dim=0)

    # ---- long inputs: pieces that fit one launch sequence ----
    def _vit_encode(self, clips: Sequence[dict], stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """ViT + merger over all clips; clips that exceed `max_patches` are encoded in groups of temporal slices (a slice only
        attends to itself, so the concatenation is exact)."""
        cfg, eng = self.cfg, self.engine
        pieces: List[dict] = []
        for c in clips:
            if "frames" in c:
                f, lay = c["frames"], c.get("layout", "THWC")
                T = f.shape[0]
                H, W = (f.shape[1], f.shape[2]) if lay == "THWC" else (f.shape[2], f.shape[3])
                per_slice = (H // cfg.patch_size) * (W // cfg.patch_size)
                slices = max(1, eng.max_patches // per_slice)
                if per_slice > eng.max_patches:
                    raise ValueError(f"one {H}x{W} frame pair is {per_slice} patches > max_patches {eng.max_patches}")
                step = slices * cfg.temporal_patch_size
                for a in range(0, T, step):
                    pieces.append(dict(frames=f[a:a + step], layout=lay))
            else:
                t, h, w = (int(x) for x in c["grid"])
                per_slice = h * w
                slices = max(1, eng.max_patches // per_slice)
                pv = c["pixel_values"]
                for a in range(0, t, slices):
                    b = min(t, a + slices)
                    pieces.append(dict(pixel_values=pv[a * per_slice:b * per_slice], grid=(b - a, h, w)))
        groups, cur, cur_p = [], [], 0
        for pc in pieces:
            if "frames" in pc:
                f, lay = pc["frames"], pc["layout"]
                H, W = (f.shape[1], f.shape[2]) if lay == "THWC" else (f.shape[2], f.shape[3])
                n = ((f.shape[0] + 1) // 2) * (H // cfg.patch_size) * (W // cfg.patch_size)
            else:
                n = pc["grid"][0] * pc["grid"][1] * pc["grid"][2]
            if cur and cur_p + n > eng.max_patches:
                groups.append(cur); cur, cur_p = [], 0
            cur.append(pc); cur_p += n
        if cur:
            groups.append(cur)
        # EVERY vision-tower launch goes through here: the tower has ONE private workspace + meta ring (lcc_engine_bind_vit_buffers),
        # so the issuing stream first waits for the previous ViT call -- wherever it ran (main stream, prefetch side stream,
        # get_video_features) -- and leaves an event behind for the next one.  Never two ViT calls in flight.
        issuing = stream if stream is not None else torch.cuda.current_stream(self.device)
        if self._vit_last_event is not None:
            issuing.wait_event(self._vit_last_event)
        outs = [eng.vit_encode(g, stream=stream) for g in groups]
        out = outs[0]
        if len(outs) > 1:
            with torch.cuda.stream(issuing):
                out = torch.cat(outs, dim=0)
        ev = torch.cuda.Event()
        ev.record(issuing)            # after the last ViT kernel AND the concatenation: consumers of `out` wait for this event
        self._vit_last_event = ev
        return out

    def _prefill(self, slots, ids_new, pos3, vit, sp, scores_buf, logits_buf) -> None:
        """One packed prefill when everything fits `max_new_rows`; otherwise consecutive streams are packed into groups of at
        most `max_new_rows` rows (one launch sequence per group), and a stream that is longer than that by itself is prefilled
        alone in pieces over its carried KV (a prompt prefilled in one or in several calls gives bit-identical logits:
        tests/test_gpu_fullsize.py; only the LAST piece's sampled token survives -- each prefill call restarts the history
        column -- which is exactly the token of the un-split prefill)."""
        cfg, eng = self.cfg, self.engine
        cap = eng.max_new_rows
        lens = [len(x) for x in ids_new]
        if sum(lens) <= cap:
            eng.prefill(slots, ids_new, pos3, vit, sp, scores_out=scores_buf, logits_out=logits_buf)
            return
        is_vid = [(ids == cfg.video_token_id) | (ids == cfg.image_token_id) for ids in ids_new]
        n_vid = [int(m.sum()) for m in is_vid]
        row0 = np.concatenate([[0], np.cumsum(n_vid)]).astype(np.int64)      # first ViT row of every stream
        sc = (lambda a, e: scores_buf[a:e]) if scores_buf is not None else (lambda a, e: None)
        lg = (lambda a, e: logits_buf[0, a:e]) if logits_buf is not None else (lambda a, e: None)   # step 0 = the prefill's token
        b, n = 0, len(slots)
        while b < n:
            if lens[b] > cap:                                   # one long prompt (one-shot video_qa / mcq): pieces over its own KV
                ids, pos, nv_before = ids_new[b], pos3[b], 0
                for a in range(0, lens[b], cap):
                    e = min(lens[b], a + cap)
                    k = int(is_vid[b][a:e].sum())
                    rows = vit[row0[b] + nv_before:row0[b] + nv_before + k] if (vit is not None and k) else None
                    nv_before += k
                    eng.prefill([slots[b]], [ids[a:e]], [pos[:, a:e]], rows, sp, scores_out=sc(b, b + 1), logits_out=lg(b, b + 1))
                b += 1
                continue
            e, rows_sum = b, 0
            while e < n and lens[e] <= cap and rows_sum + lens[e] <= cap:
                rows_sum += lens[e]
                e += 1
            rows = vit[row0[b]:row0[e]] if (vit is not None and row0[e] > row0[b]) else None
            eng.prefill(slots[b:e], ids_new[b:e], pos3[b:e], rows, sp, scores_out=sc(b, e), logits_out=lg(b, e))
            b = e

    # ---- ViT only (frames/s benchmarks, parity tests) ----
    @torch.inference_mode()
    def get_video_features(self, pixel_values_videos: torch.Tensor = None, video_grid_thw=None, frames=None,
                           frames_layout: str = "TCHW") -> torch.Tensor:
        with self._lock:
            if frames is not None:
                return self._vit_encode([dict(frames=frames.to(self.device).contiguous(), layout=frames_layout)])
            g = torch.as_tensor(video_grid_thw).reshape(-1, 3).tolist()
            pv = pixel_values_videos.to(self.device, dtype=torch.float32).contiguous()
            clips, off = [], 0
            for t, h, w in g:
                clips.append(dict(pixel_values=pv[off:off + t * h * w], grid=(t, h, w)))
                off += t * h * w
            return self._vit_encode(clips)
