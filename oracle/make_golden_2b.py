"""ORACLE / TEST INFRASTRUCTURE: tests/golden/qwen2vl2b_config0_stream.npz -- BASELINE.json configs[0] as committed reference output.

    python oracle/make_golden_2b.py            (build container, ~2 minutes)

HF `Qwen2VLForConditionalGeneration` at the REAL Qwen2-VL-2B shapes (28 layers, hidden 1536, 12/2 heads, tied 151,936-row lm_head, the
full 32-block vision tower), HF-init seed 0 (the weights `tests/test_gpu_e2e.py::test_baseline_config0_...` builds on the GPU box), one
8-frame 392x728 clip = chunks of 6 + 2 frames, greedy, repetition_penalty 1.05, 16 tokens per turn, bf16 (the reference's dtype),
FREE-running.  Stored per turn: the prompt ids, HF's own tokens, and per step the top-64 raw logits (ids + values), the values at 2,048
fixed sample ids (seeded) and the logit scale -- what a GPU test needs to compare the HIP path with the reference WITHOUT running HF on
the GPU box (prepared in round 3; the GPU test against it is next round's, once it can be run)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import qwen2vl_2b  # noqa: E402
from oracle import hf_oracle as O  # noqa: E402

SEED_W, SEED_IN, FRAMES, H, W, MAX_NEW, PENALTY, TOPK, NSAMPLE = 0, 1234, 8, 392, 728, 16, 1.05, 64, 2048
PATH = os.path.join(ROOT, "tests", "golden", "qwen2vl2b_config0_stream.npz")


def sample_ids(vocab: int) -> np.ndarray:
    return np.sort(np.random.default_rng(SEED_IN).choice(vocab, NSAMPLE, replace=False)).astype(np.int64)


def generate():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = qwen2vl_2b()
    t0 = time.time()
    hf = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=SEED_W, init_scale=1.0)
    print(f"built in {time.time() - t0:.0f} s", flush=True)
    frames = torch.from_numpy(protocol.synth_frames(FRAMES, H, W, seed=SEED_IN, layout="TCHW"))
    t0 = time.time()
    res = O.replay_stream(hf, cfg, frames, protocol.TurnBuilder(cfg, seed=SEED_IN), MAX_NEW, PENALTY, max_turns=2)
    print(f"2 turns in {time.time() - t0:.0f} s", flush=True)
    sid = sample_ids(cfg.vocab_size)
    out = dict(n_turns=np.int64(len(res)), meta=np.asarray([SEED_W, SEED_IN, FRAMES, H, W, MAX_NEW], dtype=np.int64), sample_ids=sid)
    for ti, r in enumerate(res):
        lg = torch.stack(r["logits"]).float()
        top = lg.topk(TOPK, dim=-1)
        out[f"t{ti}_ids"] = np.asarray(r["turn_ids"], dtype=np.int64)
        out[f"t{ti}_grid"] = np.asarray(r["grid"], dtype=np.int64)
        out[f"t{ti}_tokens"] = np.asarray(r["new_tokens"], dtype=np.int64)
        out[f"t{ti}_top_ids"] = top.indices.numpy().astype(np.int64)
        out[f"t{ti}_top_vals"] = top.values.numpy().astype(np.float32)
        out[f"t{ti}_sample_vals"] = lg[:, torch.from_numpy(sid)].numpy().astype(np.float32)
        out[f"t{ti}_scale"] = lg.abs().max(dim=-1).values.numpy().astype(np.float32)
    return out


if __name__ == "__main__":
    out = generate()
    np.savez_compressed(PATH, **out)
    for ti in range(int(out["n_turns"])):
        m = (out[f"t{ti}_top_vals"][:, 0] - out[f"t{ti}_top_vals"][:, 1]) / out[f"t{ti}_scale"]
        print(f"turn {ti}: tokens {out[f't{ti}_tokens'].tolist()}  top-1 margin / scale: min {float(m.min()):.4f} median {float(np.median(m)):.4f}")
    print("wrote", PATH, os.path.getsize(PATH), "bytes")
