"""ORACLE / TEST INFRASTRUCTURE: tests/golden/qwen2vl72b_fp8_full_depth.npz -- the executed HF reference through ALL 80 decoder layers of
BASELINE.json configs[4] (Qwen2-VL-72B shapes, LLM Linear weights as OCP e4m3 + fp32 row scales).

    python oracle/make_golden_72b.py [--layers N] [--frames T] [--out PATH] [--profile]      (build container: 62 GB of host RAM, 8 cores: ~40 min)
    python oracle/make_golden_72b.py --variant decisive      -> tests/golden/qwen2vl72b_fp8_decisive.npz (round 6: the `decisive` seeded weights of
    livecc_amd/weights.py -- embedding-aligned lm_head -- so that configs[4] has greedy tokens DECIDED by the model on every step: with the
    i.i.d. weights of the fixture above HF's own top-1 margin is inside its bf16 noise at 72B and no step is decided)
    (--profile also writes tests/golden/qwen2vl72b_fp8_depth_profile.npz: the residual stream of both dtypes after every 4th layer at 4 prompt
    rows x 512 hidden dims + the all-dims rms(bf16 - fp32) of every layer -- the depth profile of the reference's own error)

A 72B model does not fit the build container (288 GB in fp32), and the GPU tier has no time for an HF forward at this size.  So HF's own
modules run LAYER-STREAMED: one `Qwen2VLForConditionalGeneration` with ONE decoder layer is built per dtype (vision tower, embeddings,
final norm, lm_head: the real 72B shapes); its forward runs once to let HF build what every layer consumes (vision tower -> merged
embeddings scattered into the text embeddings, 3-D position ids -> M-RoPE cos / sin, the causal mask), captured by a pre-hook on the
decoder layer; then that SAME `Qwen2VLDecoderLayer` module is called 80 times, its parameters refilled before each call with layer l's
seeded synthetic weights (`livecc_amd.weights.synthetic_param`, bit-identical to `WeightArena.fill_tiled` on the GPU) after the per-row e4m3
quantisation the native fp8 path computes with (`fake_quantize_llm_fp8`'s rule: fp32 model = exact q x scale = the truth; bf16 model = what
the reference's dtype makes of the dequantised checkpoint).  HF code executes every arithmetic step; nothing is restated.

Input: the 6-frame first turn of the benchmark protocol (4,368 patches, 1,131 prompt ids) followed by N_FORCED - 1 seeded teacher tokens:
ONE causal pass then yields the logits of N_FORCED consecutive generation steps (position S-1 = the prefill's token, S.. = the decode
steps the native engine runs through its fp8 GEMV path, teacher-forced along the same tokens).  Stored per step as in
oracle/make_golden_7b_long.py (`pack`): bf16 top-64, both runs at 4,096 sample ids, fp32 scale / top-2, full-vocabulary rms(bf16 - fp32).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dataclasses  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.weights import dequantize_fp8_rows, hf_param_shapes, quantize_fp8_rows, synthetic_param  # noqa: E402
from oracle import hf_oracle as O  # noqa: E402
from oracle.make_golden_7b_long import pack, sample_ids  # noqa: E402

SEED_IN, T, H, W = 1234, 6, 392, 728
N_FORCED, SEED_W = 4, 0
PATH = os.path.join(ROOT, "tests", "golden", "qwen2vl72b_fp8_full_depth.npz")
PATH_DECISIVE = os.path.join(ROOT, "tests", "golden", "qwen2vl72b_fp8_decisive.npz")
VARIANT = "tiled"          # --variant decisive: embedding table + lm_head of the decisive synthetic model (the layers are the same seeded tiles)
# depth profile (--profile): the residual stream after every PROFILE_EVERY-th decoder layer, both dtypes, at the last PROFILE_ROWS prompt rows x
# PROFILE_DIMS seeded hidden dims -- where along the 80 layers does an implementation's error against the fp32 truth build up?
PROFILE_PATH = os.path.join(ROOT, "tests", "golden", "qwen2vl72b_fp8_depth_profile.npz")
PROFILE_EVERY, PROFILE_ROWS, PROFILE_DIMS = 4, 4, 512


def profile_dims(hidden: int) -> np.ndarray:
    return np.sort(np.random.default_rng(SEED_IN + 7272).choice(hidden, size=PROFILE_DIMS, replace=False)).astype(np.int64)
FP8_LINEARS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj", "lm_head")


def inputs(cfg, frames_t=T):
    frames = torch.from_numpy(protocol.synth_frames(frames_t, H, W, seed=SEED_IN, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = np.asarray(protocol.TurnBuilder(cfg, seed=SEED_IN).turn_ids(0, protocol.num_video_tokens(grid, cfg)), dtype=np.int64)
    forced = teacher_tokens(cfg)
    return frames, pv, grid, ids, forced


def teacher_tokens(cfg) -> np.ndarray:
    """N_FORCED seeded ordinary text ids (no special tokens): the stream every side is forced along."""
    return np.random.default_rng(SEED_IN + 72).integers(1000, 100000, size=N_FORCED).astype(np.int64)


def _quantised(name: str, w: torch.Tensor) -> torch.Tensor:
    if w.dim() == 2 and any(k in name for k in FP8_LINEARS) and "visual" not in name and "embed_tokens" not in name:
        q, sc = quantize_fp8_rows(w.float())
        return dequantize_fp8_rows(q, sc)
    return w.float()


def build_shell(cfg1, dtype):
    """HF model with ONE decoder layer at the 72B widths: everything except the layer filled (vision tower, embeddings, norm, lm_head)."""
    from transformers import Qwen2VLForConditionalGeneration
    from livecc_amd.weights import _normalise_hf_key
    with torch.device("meta"):
        m = Qwen2VLForConditionalGeneration._from_config(cfg1.to_hf(attn_implementation="sdpa"), dtype=dtype)
    m = m.to_empty(device="cpu")
    shapes = hf_param_shapes(cfg1)
    with torch.no_grad():
        for k, p in m.named_parameters():
            name = _normalise_hf_key(k)
            if ".layers." in name and "visual" not in name:
                continue
            p.copy_(_quantised(name, synthetic_param(name, shapes, SEED_W, "cpu", variant=VARIANT)).to(dtype))
        for name, buf in m.named_buffers():
            if "inv_freq" in name:
                dim = buf.numel() * 2
                theta = 10000.0 if "visual" in name else cfg1.rope_theta
                buf.copy_(1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim)))
    m.eval()
    return m


def fill_layer(layers, cfg_full, l):
    """Layer l's parameters (seeded synthetic, fp8-quantised Linears) into the one decoder layer of every shell (same values, each dtype)."""
    shapes = hf_param_shapes(cfg_full)
    with torch.no_grad():
        named = [dict(layer.named_parameters()) for layer in layers]
        for k in named[0]:
            name = f"language_model.layers.{l}.{k}"
            w = _quantised(name, synthetic_param(name, shapes, SEED_W, "cpu", variant=VARIANT))
            for nm in named:
                nm[k].copy_(w.to(nm[k].dtype))


@torch.inference_mode()
def capture_layer_inputs(m, cfg1, ids_all, pv, grid):
    """One forward of the shell: what HF hands to the decoder layer (hidden states = embeddings with the merged video rows scattered in,
    mask, M-RoPE cos / sin)."""
    got = {}
    layer = m.model.language_model.layers[0]

    def hook(mod, args, kwargs):
        got["h"] = args[0].detach().clone()
        got["kwargs"] = {k: v for k, v in kwargs.items()}
        return None
    hnd = layer.register_forward_pre_hook(hook, with_kwargs=True)
    dtype = next(m.parameters()).dtype
    ids = torch.as_tensor(ids_all, dtype=torch.long).view(1, -1)
    m.model.rope_deltas = None
    try:
        m(input_ids=ids, pixel_values_videos=pv.to(dtype), video_grid_thw=torch.as_tensor([list(grid)], dtype=torch.long), use_cache=False,
          logits_to_keep=1, mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(ids.numpy(), cfg1)))
    finally:
        hnd.remove()
    return got["h"], got["kwargs"]


def generate(n_layers=None, frames_t=T, path=PATH, profile_path=None):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("qwen2vl-72b")
    L = cfg.num_hidden_layers if n_layers is None else int(n_layers)
    cfg1 = dataclasses.replace(cfg, num_hidden_layers=1)
    frames, pv, grid, ids, forced = inputs(cfg, frames_t)
    ids_all = np.concatenate([ids, forced[:-1]])
    sid = torch.from_numpy(sample_ids(cfg.vocab_size))
    t0 = time.time()
    shells = {dt: build_shell(cfg1, dt) for dt in (torch.float32, torch.bfloat16)}
    print(f"shells built in {time.time() - t0:.0f} s; prompt {len(ids)} ids + {N_FORCED - 1} forced, {pv.shape[0]} patches", flush=True)
    state = {}
    for dt, m in shells.items():
        t0 = time.time()
        state[dt] = capture_layer_inputs(m, cfg1, ids_all, pv, grid)
        print(f"{dt}: vision tower + embeddings in {time.time() - t0:.0f} s, hidden {tuple(state[dt][0].shape)}", flush=True)
    layers = [shells[dt].model.language_model.layers[0] for dt in shells]
    S = len(ids)
    pdims = torch.from_numpy(profile_dims(cfg.hidden_size))
    prows = slice(S - PROFILE_ROWS, S)      # prompt rows only: the native prefill (S rows) sees exactly these (causal)
    prof = dict(layers=[], h32=[], h16=[], rms_err_bf16_all_dims=[], rms_fp32_all_dims=[])

    def take(l):
        a, t = state[torch.bfloat16][0][0, prows].float(), state[torch.float32][0][0, prows].float()
        prof["rms_err_bf16_all_dims"].append(float((a - t).double().pow(2).mean().sqrt()))
        prof["rms_fp32_all_dims"].append(float(t.double().pow(2).mean().sqrt()))
        if l < 0 or l % PROFILE_EVERY == PROFILE_EVERY - 1 or l == L - 1:
            prof["layers"].append(l)
            prof["h32"].append(t[:, pdims].numpy().astype(np.float32))
            prof["h16"].append(a[:, pdims].numpy().astype(np.float32))
    take(-1)                                 # the embeddings handed to layer 0 (video rows: vision tower + merger output)
    with torch.inference_mode():
        for l in range(L):
            t0 = time.time()
            fill_layer(layers, cfg, l)
            t1 = time.time()
            for dt, m in shells.items():
                h, kw = state[dt]
                out = m.model.language_model.layers[0](h, **kw)
                state[dt] = ((out[0] if isinstance(out, tuple) else out), kw)
            take(l)
            print(f"layer {l}: weights {t1 - t0:.0f} s, forward {time.time() - t1:.0f} s, |h|_rms fp32 {float(state[torch.float32][0].float().pow(2).mean().sqrt()):.4f}"
                  f" rms(bf16 - fp32) at the probe rows {prof['rms_err_bf16_all_dims'][-1]:.5f}", flush=True)
        logits = {}
        for dt, m in shells.items():
            h = state[dt][0][:, -N_FORCED:]
            logits[dt] = m.lm_head(m.model.language_model.norm(h))[0].float()
    l16, l32 = logits[torch.bfloat16], logits[torch.float32]
    out = dict(meta=np.asarray([SEED_IN, frames_t, H, W, N_FORCED, SEED_W, L], dtype=np.int64), variant=np.asarray(VARIANT), sample_ids=sid.numpy(), ids=ids, tokens=forced,
               grid=np.asarray(grid, dtype=np.int64), hf_bf16_argmax=l16.argmax(-1).numpy().astype(np.int64), hf_fp32_argmax=l32.argmax(-1).numpy().astype(np.int64))
    pack(out, "t0", l16, l32, sid)
    np.savez_compressed(path, **out)
    if profile_path:
        np.savez_compressed(profile_path, meta=out["meta"], rows=np.arange(S - PROFILE_ROWS, S, dtype=np.int64), dims=pdims.numpy(),
                            layers=np.asarray(prof["layers"], dtype=np.int64), h_fp32=np.stack(prof["h32"]), h_bf16=np.stack(prof["h16"]),
                            rms_err_bf16_all_dims=np.asarray(prof["rms_err_bf16_all_dims"], dtype=np.float32),
                            rms_fp32_all_dims=np.asarray(prof["rms_fp32_all_dims"], dtype=np.float32))
        print("wrote", profile_path, os.path.getsize(profile_path), "bytes", flush=True)
    print("wrote", path, os.path.getsize(path), "bytes; rms(bf16 - fp32) per step", out["t0_rms_err_bf16_full_vocab"].round(4).tolist(), "scale",
          out["t0_scale"].round(2).tolist(), flush=True)
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    nl = int(a[a.index("--layers") + 1]) if "--layers" in a else None
    ft = int(a[a.index("--frames") + 1]) if "--frames" in a else T
    if "--variant" in a:
        VARIANT = a[a.index("--variant") + 1]
        assert VARIANT in ("tiled", "decisive")
    p = a[a.index("--out") + 1] if "--out" in a else (PATH_DECISIVE if VARIANT == "decisive" else PATH)
    generate(nl, ft, p, PROFILE_PATH if "--profile" in a else None)
