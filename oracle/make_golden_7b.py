"""ORACLE / TEST INFRASTRUCTURE: tests/golden/livecc7b_first_token.npz -- the executed HF reference at the REAL LiveCC-7B shapes.

    python oracle/make_golden_7b.py            (build container: ~62 GB of host RAM, minutes on 8 cores)

HF `Qwen2VLForConditionalGeneration` at LiveCC-7B shapes, filled with the seeded synthetic weights `tiled:0` (bit-identical to
`WeightArena.fill_tiled(0)` on the GPU), runs the 6-frame first turn of BASELINE.json configs[1] (4,368 patches through the vision tower,
a 1,131-token prefill) free-running, once in bf16 (the reference's dtype) and once in fp32 on the same bf16-representable weights (the
truth).  Stored: the raw lm_head logits of the last prompt position = the distribution of the turn's first generated token, both runs
(bf16 run exactly as uint16 bit patterns, fp32 as float32), the prompt ids, the grid and the seeds.  The GPU test
(tests/test_gpu_golden.py::test_livecc7b_first_token_against_the_committed_hf_logits) compares the HIP path with these numbers WITHOUT
running HF on the GPU box: a full-shape golden vector that travels.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from oracle import hf_oracle as O, layer_probe as P  # noqa: E402

SEED_IN, T, H, W = 1234, 6, 392, 728
PATH = os.path.join(ROOT, "tests", "golden", "livecc7b_first_token.npz")


def inputs(cfg):
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=SEED_IN, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=SEED_IN).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    return frames, pv, grid, ids


def generate():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    frames, pv, grid, ids = inputs(cfg)
    t0 = time.time()
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    print(f"built bf16 model in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    l16 = P.probe(hf, cfg, ids, pv, grid, light=True)["logits"]          # fp32 copy of bf16 values
    print(f"bf16 forward {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    hf = hf.float()
    l32 = P.probe(hf, cfg, ids, pv, grid, light=True)["logits"]
    print(f"fp32 convert + forward {time.time() - t0:.0f} s", flush=True)
    b16 = l16.to(torch.bfloat16)
    assert torch.equal(b16.float(), l16), "the bf16 run's logits are bf16 numbers"
    return dict(logits_bf16_bits=b16.view(torch.int16).numpy().view(np.uint16), logits_fp32=l32.numpy().astype(np.float32),
                ids=np.asarray(ids, dtype=np.int64), grid=np.asarray(grid, dtype=np.int64),
                meta=np.asarray([SEED_IN, T, H, W, 0], dtype=np.int64))     # last entry: weight seed of tiled:<seed>


def load(path=PATH):
    d = dict(np.load(path))
    bits = torch.from_numpy(d["logits_bf16_bits"].view(np.int16).copy())
    d["logits_bf16"] = bits.view(torch.bfloat16).float().numpy()
    return d


def compare(native_logits: np.ndarray, fx: dict) -> dict:
    """Statistics of the native first-token logits against the committed HF runs (numpy only: usable on the GPU box without HF)."""
    n, b, t = native_logits.astype(np.float64).ravel(), fx["logits_bf16"].astype(np.float64).ravel(), fx["logits_fp32"].astype(np.float64).ravel()
    scale = float(np.abs(t).max())
    en, eb = n - t, b - t
    srt = np.sort(t)
    rep = dict(scale=scale, rms_err_native=float(np.sqrt((en ** 2).mean())), rms_err_ref16=float(np.sqrt((eb ** 2).mean())),
               max_err_native=float(np.abs(en).max()), max_err_ref16=float(np.abs(eb).max()),
               max_abs_native_vs_ref16=float(np.abs(n - b).max()), argmax_native=int(n.argmax()), argmax_ref16=int(b.argmax()),
               argmax_fp32=int(t.argmax()), fp32_top1_margin=float(srt[-1] - srt[-2]))
    rep["rms_ratio"] = rep["rms_err_native"] / max(rep["rms_err_ref16"], 1e-30)
    rep["max_ratio"] = rep["max_err_native"] / max(rep["max_err_ref16"], 1e-30)
    return rep


if __name__ == "__main__" and "--decisive" not in sys.argv and "--turns" not in sys.argv:
    out = generate()
    np.savez_compressed(PATH, **out)
    fx = load()
    print("self-check (bf16 run as the 'native' side):", compare(fx["logits_bf16"], fx))
    print("wrote", PATH, os.path.getsize(PATH), "bytes")


# ---------------------------------------------------------------------------------------------------------------------
# the `decisive` weight set: HF's own free-running greedy tokens over two turns (north_star: "token-id exact under greedy")
# ---------------------------------------------------------------------------------------------------------------------
PATH_DECISIVE = os.path.join(ROOT, "tests", "golden", "livecc7b_decisive_stream.npz")
TOPK = 8


def generate_decisive(turns: int = 2, max_new: int = 16, penalty: float = 1.05):
    """HF bf16 at LiveCC-7B shapes on the `decisive:0` weights (top-1 margins of tens of noise units: livecc_amd/weights.py), the
    benchmark protocol's first `turns` turns (6 + 2 frames), 16 greedy tokens each with repetition_penalty 1.05, FREE-running: HF's own
    tokens, and per step its top-8 raw logits (ids + values) and the logit scale -- enough for a GPU test to demand token identity and
    to check that the decision was not made by rounding, without running HF on the GPU box."""
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    n_frames = 6 + 2 * (turns - 1)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=SEED_IN, layout="TCHW"))
    t0 = time.time()
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "decisive:0")
    print(f"built bf16 decisive model in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    res = O.replay_stream(hf, cfg, frames, protocol.TurnBuilder(cfg, seed=SEED_IN), max_new, penalty, max_turns=turns)
    print(f"{turns} turns in {time.time() - t0:.0f} s", flush=True)
    out = dict(n_turns=np.int64(len(res)), meta=np.asarray([SEED_IN, n_frames, H, W, max_new, 0], dtype=np.int64))
    for ti, r in enumerate(res):
        lg = torch.stack(r["logits"]).float()                          # [N, V] raw lm_head logits of every step
        top = lg.topk(TOPK, dim=-1)
        out[f"t{ti}_ids"] = np.asarray(r["turn_ids"], dtype=np.int64)
        out[f"t{ti}_grid"] = np.asarray(r["grid"], dtype=np.int64)
        out[f"t{ti}_tokens"] = np.asarray(r["new_tokens"], dtype=np.int64)
        out[f"t{ti}_top_ids"] = top.indices.numpy().astype(np.int64)
        out[f"t{ti}_top_vals"] = top.values.numpy().astype(np.float32)
        out[f"t{ti}_scale"] = lg.abs().max(dim=-1).values.numpy().astype(np.float32)
    return out


if __name__ == "__main__" and "--decisive" in sys.argv:
    out = generate_decisive()
    np.savez_compressed(PATH_DECISIVE, **out)
    for ti in range(int(out["n_turns"])):
        m = out[f"t{ti}_top_vals"][:, 0] - out[f"t{ti}_top_vals"][:, 1]
        print(f"turn {ti}: tokens {out[f't{ti}_tokens'].tolist()}  min raw top-1 margin / scale = {float((m / out[f't{ti}_scale']).min()):.3f}")
    print("wrote", PATH_DECISIVE, os.path.getsize(PATH_DECISIVE), "bytes")


# ---------------------------------------------------------------------------------------------------------------------
# the benchmark's two turns on the tiled:0 weights: HF bf16 free-running + HF fp32 teacher-forced along the same tokens
# ---------------------------------------------------------------------------------------------------------------------
PATH_TURNS = os.path.join(ROOT, "tests", "golden", "livecc7b_two_turns.npz")
TOPK_T, NSAMPLE_T = 64, 4096


def sample_ids(vocab: int) -> np.ndarray:
    return np.sort(np.random.default_rng(SEED_IN).choice(vocab, NSAMPLE_T, replace=False)).astype(np.int64)


def generate_turns(turns: int = 2, max_new: int = 16, penalty: float = 1.05):
    """BASELINE configs[1]'s first two turns (6 + 2 frames, 16 greedy tokens each, repetition_penalty 1.05) at LiveCC-7B shapes on the
    `tiled:0` weights: HF bf16 free-running (its own tokens), then HF fp32 on the same weights teacher-forced along those tokens.  Per
    step: the bf16 run's top-64 raw logits (ids + values), both runs' values at 4,096 fixed sample ids and at those top-64 ids, the logit
    scale, and the fp32 top-1 / top-2 -- enough for error-ratio and margin-aware token checks on the GPU box without HF (prepared in
    round 3; the GPU test against it is next round's)."""
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    n_frames = 6 + 2 * (turns - 1)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=SEED_IN, layout="TCHW"))
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    t0 = time.time()
    res = O.replay_stream(hf, cfg, frames, protocol.TurnBuilder(cfg, seed=SEED_IN), max_new, penalty, max_turns=turns)
    print(f"bf16: {turns} turns in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    hf = hf.float()
    s32 = O.OracleStream(hf, cfg)
    sid = torch.from_numpy(sample_ids(cfg.vocab_size))
    out = dict(n_turns=np.int64(len(res)), meta=np.asarray([SEED_IN, n_frames, H, W, max_new, 0], dtype=np.int64), sample_ids=sid.numpy())
    for ti, (r, (a, b)) in enumerate(zip(res, protocol.split_clip(n_frames))):
        pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
        r32 = s32.turn(r["turn_ids"], pv, grid, max_new_tokens=max_new, repetition_penalty=penalty, teacher_tokens=r["new_tokens"])
        l16, l32 = torch.stack(r["logits"]).float(), torch.stack(r32["logits"]).float()
        top = l16.topk(TOPK_T, dim=-1)
        t32 = l32.topk(2, dim=-1)
        out[f"t{ti}_ids"] = np.asarray(r["turn_ids"], dtype=np.int64)
        out[f"t{ti}_grid"] = np.asarray(grid, dtype=np.int64)
        out[f"t{ti}_tokens"] = np.asarray(r["new_tokens"], dtype=np.int64)
        out[f"t{ti}_top_ids"] = top.indices.numpy().astype(np.int64)
        out[f"t{ti}_top_vals_bf16"] = top.values.numpy().astype(np.float32)
        out[f"t{ti}_top_vals_fp32"] = torch.gather(l32, 1, top.indices).numpy().astype(np.float32)
        out[f"t{ti}_sample_vals_bf16"] = l16[:, sid].numpy().astype(np.float32)
        out[f"t{ti}_sample_vals_fp32"] = l32[:, sid].numpy().astype(np.float32)
        out[f"t{ti}_scale"] = l32.abs().max(dim=-1).values.numpy().astype(np.float32)
        out[f"t{ti}_fp32_top2_ids"] = t32.indices.numpy().astype(np.int64)
        out[f"t{ti}_fp32_top2_vals"] = t32.values.numpy().astype(np.float32)
        out[f"t{ti}_rms_err_bf16_full_vocab"] = (l16 - l32).double().pow(2).mean(dim=-1).sqrt().numpy().astype(np.float32)
    print(f"fp32: convert + {turns} turns in {time.time() - t0:.0f} s", flush=True)
    return out


if __name__ == "__main__" and "--turns" in sys.argv:
    out = generate_turns()
    np.savez_compressed(PATH_TURNS, **out)
    for ti in range(int(out["n_turns"])):
        e = np.sqrt(((out[f"t{ti}_sample_vals_bf16"] - out[f"t{ti}_sample_vals_fp32"]).astype(np.float64) ** 2).mean(axis=-1))
        print(f"turn {ti}: tokens {out[f't{ti}_tokens'].tolist()}")
        print(f"   rms(bf16 - fp32) over the sample ids / full vocabulary: {e.mean():.4f} / {float(out[f't{ti}_rms_err_bf16_full_vocab'].mean()):.4f}")
    print("wrote", PATH_TURNS, os.path.getsize(PATH_TURNS), "bytes")
