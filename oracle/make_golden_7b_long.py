"""ORACLE / TEST INFRASTRUCTURE: tests/golden/livecc7b_oneshot480.npz -- the executed HF reference at BASELINE.json configs[3]'s REAL shapes.

    python oracle/make_golden_7b_long.py [--no-fp32]      (build container: 62 GB of host RAM, 8 cores: ~10 min bf16 + ~45 min fp32)

The reference's `video_qa` / MCQ first turn (ref demo/infer.py:182-242, evaluation/distributed_mcq_predictor.py:72-105) at LiveCC-7B
shapes: all 480 frames 280x280 (96,000 patches in 240 temporal slices -> 24,000 visual tokens) + a 24-id query in ONE `generate`
call -- a 24,058-row prefill -- then N_NEW greedy tokens over the ~24k-key cache (repetition_penalty 1.05).  HF
`Qwen2VLForConditionalGeneration` filled with the seeded synthetic weights `tiled:0` (bit-identical to `WeightArena.fill_tiled(0)` on the
GPU) runs it free-running in bf16 (the reference's dtype: its own tokens) and then in fp32 on the same bf16-representable weights,
teacher-forced along those tokens (the truth).  Positions follow the INSTALLED transformers (5.15 text-offset rule, "hf5"): the GPU test
constructs the native model with text_offset_rule="hf5".

Stored per step (step 0 = the prefill's token, steps 1.. = decode steps at L ~ 24k): the bf16 run's top-64 raw logits (ids + values),
both runs' values at those ids and at 4,096 fixed sample ids, the fp32 logit scale, the fp32 top-2, and the full-vocabulary rms of
(bf16 - fp32) -- enough for error-ratio and margin-aware token checks on the GPU box WITHOUT running HF there
(tests/test_gpu_golden.py::test_livecc7b_oneshot480_against_the_committed_hf_logits).
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
from oracle import hf_oracle as O  # noqa: E402

SEED_IN, T, H, W = 1234, 480, 280, 280
QUERY_LEN, N_NEW, PENALTY = 24, 8, 1.05
TOPK, NSAMPLE = 64, 4096
PATH = os.path.join(ROOT, "tests", "golden", "livecc7b_oneshot480.npz")


def prompt_ids(cfg):
    """The one-shot prompt of bench.py's `replay_oneshot`: system + user header + Time + the whole clip's placeholders + a 24-id query."""
    grid = protocol.grid_of(T, H, W, cfg)
    b = protocol.TurnBuilder(cfg, seed=SEED_IN)
    b.query_len = QUERY_LEN
    return b.turn_ids(0, protocol.num_video_tokens(grid, cfg), with_query=True), grid


def sample_ids(vocab: int) -> np.ndarray:
    return np.sort(np.random.default_rng(SEED_IN).choice(vocab, NSAMPLE, replace=False)).astype(np.int64)


def pack(out, key, logits16, logits32, sid):
    top = logits16.topk(TOPK, dim=-1)
    out[f"{key}_top_ids"] = top.indices.numpy().astype(np.int64)
    out[f"{key}_top_vals_bf16"] = top.values.numpy().astype(np.float32)
    out[f"{key}_sample_vals_bf16"] = logits16[:, sid].numpy().astype(np.float32)
    out[f"{key}_scale_bf16"] = logits16.abs().max(dim=-1).values.numpy().astype(np.float32)
    if logits32 is not None:
        t32 = logits32.topk(2, dim=-1)
        out[f"{key}_top_vals_fp32"] = torch.gather(logits32, 1, top.indices).numpy().astype(np.float32)
        out[f"{key}_sample_vals_fp32"] = logits32[:, sid].numpy().astype(np.float32)
        out[f"{key}_scale"] = logits32.abs().max(dim=-1).values.numpy().astype(np.float32)
        out[f"{key}_fp32_top2_ids"] = t32.indices.numpy().astype(np.int64)
        out[f"{key}_fp32_top2_vals"] = t32.values.numpy().astype(np.float32)
        out[f"{key}_rms_err_bf16_full_vocab"] = (logits16 - logits32).double().pow(2).mean(dim=-1).sqrt().numpy().astype(np.float32)


def generate(with_fp32=True, path=PATH):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    ids, grid = prompt_ids(cfg)
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=SEED_IN, layout="TCHW"))
    pv, g2 = O.patchify_normalize_ref(frames, cfg)
    assert tuple(g2) == tuple(grid)
    sid = torch.from_numpy(sample_ids(cfg.vocab_size))
    t0 = time.time()
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    print(f"built bf16 model in {time.time() - t0:.0f} s; prompt {len(ids)} ids, {pv.shape[0]} patches", flush=True)
    t0 = time.time()
    r16 = O.OracleStream(hf, cfg).turn(ids, pv, grid, max_new_tokens=N_NEW, repetition_penalty=PENALTY, force_length=True)
    print(f"bf16: prefill + {N_NEW} tokens in {time.time() - t0:.0f} s: {r16['new_tokens']}", flush=True)
    l16 = torch.stack(r16["logits"]).float()
    out = dict(meta=np.asarray([SEED_IN, T, H, W, N_NEW, 0, QUERY_LEN], dtype=np.int64), sample_ids=sid.numpy(),
               ids_len=np.int64(len(ids)), ids_head=np.asarray(ids[:64], dtype=np.int64), ids_tail=np.asarray(ids[-64:], dtype=np.int64),
               grid=np.asarray(grid, dtype=np.int64), tokens=np.asarray(r16["new_tokens"], dtype=np.int64))
    pack(out, "t0", l16, None, sid)
    np.savez_compressed(path, **out)              # the bf16 half survives a failure of the fp32 leg
    print("wrote (bf16 half)", path, os.path.getsize(path), "bytes", flush=True)
    if not with_fp32:
        return out
    t0 = time.time()
    hf = hf.float()
    r32 = O.OracleStream(hf, cfg).turn(ids, pv, grid, max_new_tokens=N_NEW, repetition_penalty=PENALTY, force_length=True,
                                       teacher_tokens=r16["new_tokens"])
    print(f"fp32: convert + teacher-forced prefill + {N_NEW} tokens in {time.time() - t0:.0f} s", flush=True)
    l32 = torch.stack(r32["logits"]).float()
    pack(out, "t0", l16, l32, sid)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)
    return out


if __name__ == "__main__":
    o = generate(with_fp32="--no-fp32" not in sys.argv)
    if "t0_sample_vals_fp32" in o:
        e = np.sqrt(((o["t0_sample_vals_bf16"] - o["t0_sample_vals_fp32"]).astype(np.float64) ** 2).mean(axis=-1))
        print(f"rms(bf16 - fp32) over the sample ids / full vocabulary per step: {e.round(4).tolist()} / "
              f"{o['t0_rms_err_bf16_full_vocab'].round(4).tolist()}; scale {o['t0_scale'].round(2).tolist()}")
