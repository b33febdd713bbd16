"""ORACLE / TEST INFRASTRUCTURE (never imported by `livecc_amd/`): the reference's OWN orchestrator, EXECUTED.

`ref demo/infer.py` (the file itself, loaded from /root/reference by path -- nothing of it is copied here) is run with its
third-party / C++ dependencies stood in for, and with `Qwen2VLForConditionalGeneration` bound to a model of our choosing:

    liger_kernel.transformers.apply_liger_kernel_to_qwen2_vl  -> no-op                      (ref demo/infer.py:2-3: Triton, GPU only)
    decord.VideoReader                                         -> reader over a seeded synthetic DecodedVideo (real frames + pts)
    torchvision.transforms.functional.resize                   -> oracle/resize_ref.py (torch's own CPU antialias bicubic = what
                                                                  torchvision's tensor path calls; torchvision is not installed)
    qwen_vl_utils.vision_process                               -> constants / smart_nframes / smart_resize of livecc_amd (the same
                                                                  stand-in oracle/make_frame_selection_golden.py uses; un-vendored)
    livecc_utils                                               -> the REAL package, imported from /root/reference/livecc-utils/src
    AutoProcessor.from_pretrained                              -> `ProcessorShim`: chat template + placeholder expansion + tokenizer of
                                                                  `livecc_amd.text.TextFrontEnd`, normalise + patchify of the oracle
                                                                  (HF's video processor needs torchvision)
    Qwen2VLForConditionalGeneration.from_pretrained            -> `RecordingModel` (answers with a deterministic function of its inputs and
                                                                  records every call) or `HFBackedModel` (HF's own class at tiny shapes on
                                                                  CPU = the reference's arithmetic)

Because the stand-ins live in `sys.modules`, the reference always runs in a SUBPROCESS (`python oracle/ref_infer_harness.py ...`); the
parent compares what it wrote (a call trace) with `livecc_amd.infer.LiveCCDemoInfer` driven over the same model, tokenizer and video:

  * tests/test_ref_infer_executed.py (CPU tier, needs /root/reference): the two orchestrators make IDENTICAL `generate` calls (ids,
    grids, pixel values, sampling arguments, cache hand-over) and return identical responses, for `live_cc_once_for_evaluation` and for
    the `demo/cli.py` loop over `live_cc`;
  * tests/golden/ref_infer_trace.json (written by `--mode hf --out ...`): the executed reference over HF's model -- the fixture the GPU
    tier replays through `livecc_amd.infer.LiveCCDemoInfer` + the native engine (tests/test_gpu_golden.py), since /root/reference does
    not exist on the GPU box.

Version drift handled where the reference meets transformers 5.15: its `prepare_inputs_for_generation` patch
(livecc_utils/generation_patch.py:37 reads `cache_position`, gone in 5.15) is accepted as an assignment and not installed on the HF
model, which forwards multimodal inputs on the first forward of every generate call by itself; 5.x needs `mm_token_type_ids`; the
attention mask the HF processor returns for the NEW ids only (ref demo/infer.py:287-296 passes it on) is dropped when its length no
longer matches `cat(past_ids, new_ids)`.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF_DEMO = "/root/reference/demo/infer.py"
REF_UTILS_SRC = "/root/reference/livecc-utils/src"
TOKENIZER_DIR = os.path.join(ROOT, "tests", "golden", "ref_infer_tokenizer")
TRACE_PATH = os.path.join(ROOT, "tests", "golden", "ref_infer_trace.json")

EMBED_SCALE = 3.4     # embedding-aligned lm_head of the fixture model (hf_weights): why 3.4 is explained there
QUERY = "what is happening now?"
VIDEO_NAME = "videos/synthetic_12s.mp4"


def available() -> bool:
    return os.path.exists(REF_DEMO) and os.path.isdir(REF_UTILS_SRC)


# ---------------------------------------------------------------------------------------------------------------------
# shared scenario pieces (used by the reference subprocess AND by the tests that drive livecc_amd over the same inputs)
# ---------------------------------------------------------------------------------------------------------------------
def make_video(seed: int = 11, n: int = 360, fps: float = 30.0, h: int = 96, w: int = 128):
    """Seeded synthetic decoded video: uint8 frames [n,h,w,3] + pts with a little jitter (decoder order, like decord)."""
    from livecc_amd.video import DecodedVideo
    rng = np.random.RandomState(seed)
    frames = torch.from_numpy(rng.randint(0, 256, size=(n, h, w, 3), dtype=np.uint8))
    pts = np.arange(n) / fps + rng.uniform(0, 0.004, n)
    return DecodedVideo(frames, np.sort(pts), fps)


def model_config(tokenizer, base: str = "tiny"):
    """`tiny` shapes with the special ids of `tokenizer` and a vocabulary that just covers it."""
    import dataclasses
    from livecc_amd import config as C
    tk = tokenizer
    vocab = 32 * ((len(tk) + 31) // 32)
    cfg = C.tiny(vocab) if base == "tiny" else C.PRESETS[base]()
    return dataclasses.replace(cfg, vocab_size=vocab, video_token_id=tk.convert_tokens_to_ids("<|video_pad|>"),
                               eos_token_id=tk.convert_tokens_to_ids("<|im_end|>"), image_token_id=tk.convert_tokens_to_ids("<|image_pad|>"),
                               vision_start_token_id=tk.convert_tokens_to_ids("<|vision_start|>"),
                               vision_end_token_id=tk.convert_tokens_to_ids("<|vision_end|>"),
                               bos_token_id=tk.convert_tokens_to_ids("<|endoftext|>"))


def load_tokenizer(path: str = TOKENIZER_DIR):
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path)


def sha(t) -> str:
    a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def fake_tokens(input_ids: np.ndarray, n: int, text_hi: int, eos: int):
    """The RecordingModel's "generation": a deterministic function of the call's ids, inside the tokenizer's text range; every third call
    ends early with EOS (the orchestrators must carry `sequences[:, :-1]` either way, ref demo/infer.py:174)."""
    s = int(np.asarray(input_ids, dtype=np.int64).sum() % 1000003)
    toks = [int(7 + (s * (k + 3) + 13 * k * k) % (text_hi - 7)) for k in range(n)]
    if s % 3 == 0 and n > 3:
        toks = toks[:n - 2] + [int(eos)]
    return toks


class _Handle:
    """What the RecordingModel hands back as `past_key_values`: an opaque, truthy object that remembers the cached length."""

    def __init__(self, n):
        self.n, self.released = n, False

    def get_seq_length(self):
        return self.n

    def release(self):
        self.released = True

    def __bool__(self):
        return True


class RecordingModel:
    """Stands where `Qwen2VLForConditionalGeneration` stands for BOTH orchestrators (the reference calls it with the HF processor's
    `pixel_values_videos`, livecc_amd with uint8 `frames`): records every generate call in comparable form."""

    def __init__(self, cfg):
        from livecc_amd.modeling import _Cfg
        self.cfg, self.config = cfg, _Cfg(cfg)
        self.device = torch.device("cpu")
        self.eos_token_ids = [cfg.eos_token_id]
        self.generation_config = {}
        self.prepare_inputs_for_generation = None
        self.calls = []
        self.text_hi = min(cfg.bos_token_id, cfg.vision_start_token_id, cfg.image_token_id, cfg.video_token_id, cfg.eos_token_id)

    def generate(self, input_ids=None, pixel_values_videos=None, video_grid_thw=None, past_key_values=None, frames=None,
                 frames_layout="TCHW", attention_mask=None, return_dict_in_generate=True, do_sample=None, repetition_penalty=1.0,
                 logits_processor=None, max_new_tokens=16, min_new_tokens=None, pad_token_id=None, **kw):
        from oracle import hf_oracle as O
        ids = torch.as_tensor(input_ids).reshape(-1).cpu().numpy().astype(np.int64)
        if frames is not None:                                      # livecc_amd hands over uint8 frames: what the processor would make of them
            f = frames if frames_layout == "TCHW" else frames.permute(0, 3, 1, 2)
            pv, grid = O.patchify_normalize_ref(f.cpu().contiguous(), self.cfg)
        else:
            pv, grid = pixel_values_videos, tuple(int(x) for x in torch.as_tensor(video_grid_thw).reshape(-1).tolist())
        past = 0 if past_key_values is None else past_key_values.get_seq_length()
        procs = [dict(kind=type(p).__name__, token_id=int(p.token_id), base=float(p.base_threshold), step=float(p.step), count=int(p.count))
                 for p in (logits_processor or [])]
        toks = fake_tokens(ids, max_new_tokens, self.text_hi, self.cfg.eos_token_id)
        self.calls.append(dict(input_ids=ids.tolist(), grid=list(grid), pixel_sha=sha(pv.float()), pixel_rows=int(pv.shape[0]), past_len=int(past),
                               do_sample=do_sample, repetition_penalty=float(repetition_penalty), max_new_tokens=int(max_new_tokens),
                               pad_token_id=None if pad_token_id is None else int(pad_token_id), logits_processor=procs, tokens=toks,
                               extra=sorted(k for k in kw if k not in ("mm_token_type_ids",))))
        seq = torch.from_numpy(np.concatenate([ids, np.asarray(toks, dtype=np.int64)])).view(1, -1)
        if logits_processor:
            for p in logits_processor:
                p.count += len(toks)                                # HF calls the processor once per generated token (ref :17-22)
        from livecc_amd.modeling import GenerateOutput
        return GenerateOutput(sequences=seq, past_key_values=_Handle(seq.shape[1] - 1))

    # the two methods livecc_amd's orchestrator may call on a real model
    def new_stream(self):
        return _Handle(0)


# ---------------------------------------------------------------------------------------------------------------------
# everything below runs only inside the reference subprocess
# ---------------------------------------------------------------------------------------------------------------------
VIDEOS = {}


class _Batch:
    def __init__(self, arr):
        self._a = arr

    def asnumpy(self):
        return self._a


class FakeVideoReader:
    """decord.VideoReader over a DecodedVideo (the calls ref video_process_patch.py / demo/infer.py:91-94 make)."""

    def __init__(self, path, num_threads=0):
        self.v = VIDEOS[path]
        self._frame_pts = None

    def __len__(self):
        return len(self.v)

    def get_avg_fps(self):
        return self.v.avg_fps

    def get_frame_timestamp(self, i):
        p = self.v.pts
        self._frame_pts = np.stack([p, p], axis=1)
        return self._frame_pts[i]

    def next(self):
        return self.v.frames[0].numpy()

    def get_batch(self, idxs):
        return _Batch(self.v.frames[torch.as_tensor([int(i) for i in idxs], dtype=torch.long)].numpy())


def install_stubs():
    from oracle import make_frame_selection_golden as G
    from oracle.resize_ref import resize_ref
    G.install_stubs()                                    # decord / torchvision / qwen_vl_utils stand-ins (frame-selection golden generator)
    sys.modules["decord"].VideoReader = FakeVideoReader
    fn = sys.modules["torchvision.transforms.functional"]

    def resize(video, size, interpolation=None, antialias=None):
        assert antialias and str(interpolation).lower().endswith("bicubic")
        return resize_ref(video, int(size[0]), int(size[1]))
    fn.resize = resize
    lk, lkt = types.ModuleType("liger_kernel"), types.ModuleType("liger_kernel.transformers")
    lkt.apply_liger_kernel_to_qwen2_vl = lambda *a, **k: None
    lk.transformers = lkt
    sys.modules.update({"liger_kernel": lk, "liger_kernel.transformers": lkt})
    qv = types.ModuleType("qwen_vl_utils")               # `from qwen_vl_utils import process_vision_info` (video_qa only: not exercised)
    qv.__path__ = []

    def process_vision_info(conversation):
        raise NotImplementedError("qwen_vl_utils.fetch_video is third-party and un-vendored: video_qa is not part of this harness")
    qv.process_vision_info = process_vision_info
    sys.modules["qwen_vl_utils"] = qv
    real_exists = os.path.exists
    os.path.exists = lambda p: True if p in VIDEOS else real_exists(p)      # ref video_process_patch.py:38 tests the path


class ProcessorShim:
    """The three processor calls of the reference (apply_chat_template, __call__, decode) + `.tokenizer` (ref demo/infer.py:48-58, 147-157, 175)."""

    def __init__(self, fe, cfg):
        self.fe, self.cfg, self.tokenizer = fe, cfg, fe.tokenizer

    def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=False, **kw):
        assert tokenize is False
        return self.fe.apply_chat_template(conversation, add_generation_prompt=add_generation_prompt)

    def __call__(self, text=None, images=None, videos=None, return_tensors="pt", return_attention_mask=True, **kw):
        from transformers import BatchFeature
        from oracle import hf_oracle as O
        assert images is None and return_tensors == "pt"
        grids, pvs = [], []
        for clip in videos or []:
            u8 = clip if clip.dtype == torch.uint8 else clip.round().clamp(0, 255).to(torch.uint8)
            assert torch.equal(u8.float(), clip.float()), "the reference's float clip holds the resize's integral uint8 values (:107)"
            pv, grid = O.patchify_normalize_ref(u8.contiguous(), self.cfg)
            pvs.append(pv)
            grids.append(grid)
        ids = torch.from_numpy(self.fe.encode(text, grids)).view(1, -1)
        data = dict(input_ids=ids)
        if return_attention_mask:
            data["attention_mask"] = torch.ones_like(ids)
        if grids:
            data["pixel_values_videos"] = torch.cat(pvs, dim=0)
            data["video_grid_thw"] = torch.as_tensor([list(g) for g in grids], dtype=torch.long)
        return BatchFeature(data)

    def decode(self, ids, skip_special_tokens=True):
        return self.fe.decode(torch.as_tensor(ids).reshape(-1).tolist(), skip_special_tokens=skip_special_tokens)


class HFBackedModel:
    """HF's `Qwen2VLForConditionalGeneration` (tiny shapes, CPU) behind the surface the reference uses; records every call."""

    def __init__(self, hf, cfg):
        from livecc_amd.modeling import _Cfg
        self.hf, self.cfg, self.config = hf, cfg, _Cfg(cfg)
        self.device = torch.device("cpu")
        self.calls = []
        self._delta = {}          # HF keeps rope_deltas on the module (Q2VL:857); here per cache object

    def __setattr__(self, k, v):
        if k == "prepare_inputs_for_generation":      # ref demo/infer.py:50; raises KeyError on transformers 5.15 (module docstring)
            return object.__setattr__(self, "_patched_prepare", v)
        object.__setattr__(self, k, v)

    @torch.inference_mode()
    def generate(self, input_ids=None, attention_mask=None, pixel_values_videos=None, video_grid_thw=None, past_key_values=None, **kw):
        from livecc_amd import protocol
        if attention_mask is not None and attention_mask.shape[-1] != input_ids.shape[-1]:
            attention_mask = None
        extra = dict(mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(input_ids.numpy(), self.cfg)))
        if pixel_values_videos is not None:
            extra.update(pixel_values_videos=pixel_values_videos, video_grid_thw=video_grid_thw)
        self.hf.model.rope_deltas = self._delta.get(id(past_key_values)) if past_key_values is not None else None
        procs = kw.pop("logits_processor", None)
        if procs is not None:
            from transformers import LogitsProcessorList
            procs = LogitsProcessorList(procs)
        out = self.hf.generate(input_ids=input_ids, attention_mask=attention_mask, past_key_values=past_key_values,
                               logits_processor=procs, output_logits=True, **extra, **kw)
        self._delta[id(out.past_key_values)] = self.hf.model.rope_deltas
        ids = input_ids[0].numpy()
        lg = torch.stack([l[0].float() for l in out.logits])
        top = lg.topk(2, dim=-1)
        self.calls.append(dict(input_ids=ids.tolist(), grid=[] if video_grid_thw is None else video_grid_thw.reshape(-1).tolist(),
                               pixel_sha=None if pixel_values_videos is None else sha(pixel_values_videos.float()),
                               past_len=0 if past_key_values is None else len(ids) - 0, tokens=out.sequences[0, len(ids):].tolist(),
                               do_sample=kw.get("do_sample"), repetition_penalty=float(kw.get("repetition_penalty", 1.0)),
                               max_new_tokens=int(kw.get("max_new_tokens", 0)), pad_token_id=kw.get("pad_token_id"),
                               top2_vals=[[round(float(x), 4) for x in r] for r in top.values.numpy()], top2_ids=top.indices.numpy().tolist(),
                               scale=[round(float(x), 4) for x in lg.abs().max(dim=-1).values.numpy()]))
        return out


def load_reference_infer(model_factory, processor_factory):
    """Execute ref demo/infer.py with the two HF names it imports bound to our factories.  Returns the module."""
    import importlib.util
    import transformers
    if REF_UTILS_SRC not in sys.path:
        sys.path.insert(0, REF_UTILS_SRC)
    saved = transformers.Qwen2VLForConditionalGeneration, transformers.AutoProcessor

    class _Model:
        from_pretrained = staticmethod(model_factory)

    class _Proc:
        from_pretrained = staticmethod(processor_factory)
    transformers.Qwen2VLForConditionalGeneration, transformers.AutoProcessor = _Model, _Proc
    try:
        spec = importlib.util.spec_from_file_location("ref_demo_infer", REF_DEMO)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        transformers.Qwen2VLForConditionalGeneration, transformers.AutoProcessor = saved
    return mod


def hf_weights(cfg, seed: int = 0):
    """The tiny HF model of the fixture, in fp32 (bf16-representable weights): HF init + an embedding-aligned lm_head (the `decisive`
    construction of livecc_amd/weights.py at tiny shapes) so that greedy tokens are decided by the model, not by rounding."""
    from oracle import hf_oracle as O
    from livecc_amd.weights import DECISIVE_A, DECISIVE_B
    hf = O.build_hf_model(cfg, dtype=torch.float32, seed=seed, init_scale=2.0)
    # EMBED_SCALE 3.4: HF-bf16's top-1 margin is >= 8 % of the logit scale on every step of both scenarios (bf16 noise at these shapes:
    # 1-2 %); at 2.5 the tokens vary more with the video but margins fall to 1-2 % (undecidable), at >= 4 the walk ignores the layers
    with torch.no_grad():
        emb = hf.model.language_model.embed_tokens.weight
        emb.mul_(EMBED_SCALE)
        V = emb.shape[0]
        perm = (torch.arange(V, dtype=torch.int64) * DECISIVE_A + DECISIVE_B) % V
        hf.lm_head.weight.copy_(emb[perm] / EMBED_SCALE)
    O.round_weights_to_bf16(hf)
    return hf


def scenario_once(infer, video=VIDEO_NAME):
    return infer.live_cc_once_for_evaluation(query=QUERY, video=video, video_start=1.0, video_end=9.0, max_new_tokens=8,
                                             repetition_penalty=1.15)


def scenario_live(infer, video=VIDEO_NAME, **kw):
    """ref demo/cli.py:10-24: the video clock advances second by second; every due chunk is commented."""
    state = {"video_path": video}
    out = []
    for t in range(14):
        state["video_timestamp"] = t
        for (a, b), response, state in infer.live_cc(message=QUERY if t < 6 else "and now?", state=state, max_pixels=4 * 28 * 28,
                                                     repetition_penalty=1.05, do_sample=False, streaming_eos_base_threshold=0.05,
                                                     streaming_eos_threshold_step=0.01, **kw):
            out.append([float(a), float(b), response])
        if state.get("video_end", False):
            break
    return out, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in state.items()
                 if k in ("last_timestamp", "last_video_pts_index", "message", "video_end")}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["recording", "hf"], default="recording")
    ap.add_argument("--tokenizer", default=TOKENIZER_DIR)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if not available():
        raise SystemExit("needs /root/reference (build container only)")
    from transformers import Qwen2VLForConditionalGeneration  # noqa: F401  (before the stand-ins: transformers probes torchvision by find_spec)
    from livecc_amd.text import TextFrontEnd
    tk = load_tokenizer(a.tokenizer)
    cfg = model_config(tk)
    VIDEOS[VIDEO_NAME] = make_video()
    install_stubs()
    if a.mode == "recording":
        model = RecordingModel(cfg)
    else:
        model = HFBackedModel(hf_weights(cfg).to(torch.bfloat16), cfg)
    created = {}

    def model_factory(path, torch_dtype=None, device_map=None, attn_implementation=None):
        created.update(path=path, torch_dtype=torch_dtype, device_map=device_map, attn_implementation=attn_implementation)
        return model

    def processor_factory(path, use_fast=None):
        created["use_fast"] = use_fast
        return ProcessorShim(TextFrontEnd(tk, cfg), cfg)
    ref = load_reference_infer(model_factory, processor_factory)
    infer = ref.LiveCCDemoInfer(model_path="chenjoya/LiveCC-7B-Instruct", device="cpu")
    res = dict(created={k: (str(v) if v is not None else None) for k, v in created.items()},
               streaming_eos_token_id=int(infer.streaming_eos_token_id), system_prompt_offset=int(infer.system_prompt_offset))
    n0 = len(model.calls)
    res["once"] = dict(responses=scenario_once(infer), calls=model.calls[n0:])
    n0 = len(model.calls)
    live, st = scenario_live(infer)
    res["live"] = dict(responses=live, state=st, calls=model.calls[n0:])
    with open(a.out, "w") as f:
        json.dump(res, f)
    print(f"wrote {a.out}: once {len(res['once']['calls'])} calls, live {len(res['live']['calls'])} calls")


if __name__ == "__main__":
    main()
