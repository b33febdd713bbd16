"""ORACLE / TEST INFRASTRUCTURE: tests/golden/livecc7b_long480_stream.npz -- the executed HF reference for BASELINE.json configs[3] in its
STREAMING form at the REAL LiveCC-7B shapes: one 480-frame 280x280 video replayed through the reference protocol (a 6-frame turn + 237
two-frame turns, 12 greedy tokens per turn, repetition_penalty 1.05), the KV cache growing to ~32.6k keys.

    python oracle/make_golden_7b_long_stream.py        (build container: ~35 min on 8 cores, bf16 only)

HF bf16 (the reference's dtype), `tiled:0` weights, free-running.  Stored: the tokens of EVERY turn (the GPU test teacher-forces the native
engine along them so that both caches hold the same history) and, for the probe turns (60, 120, 180 and the last three), per step the top-64
raw logits (ids + values) and the logit scale.  No fp32 leg: 238 fp32 turns at 7B are hours on this host; the error-ratio statistics of
configs[3] come from the one-shot fixture (make_golden_7b_long.py), this one pins the incremental path -- 238 cache appends, positions
`past_len + i + rope_delta` up to 32k, decode attention over 1,000+ key tiles -- against the reference within the gross-error bound.
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

SEED, N_FRAMES, H, W, N_NEW, PENALTY, TOPK = 1234, 480, 280, 280, 12, 1.05, 64
PATH = os.path.join(ROOT, "tests", "golden", "livecc7b_long480_stream.npz")


def probe_turns(n_turns):
    return sorted(set([60, 120, 180, n_turns - 3, n_turns - 2, n_turns - 1]))


def generate(path=PATH):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    frames = torch.from_numpy(protocol.synth_frames(N_FRAMES, H, W, seed=SEED, layout="TCHW"))
    builder = protocol.TurnBuilder(cfg, seed=SEED)
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    chunks = protocol.split_clip(N_FRAMES)
    probes = probe_turns(len(chunks))
    stream = O.OracleStream(hf, cfg)
    out = dict(meta=np.asarray([SEED, N_FRAMES, H, W, N_NEW, 0, len(chunks)], dtype=np.int64), probe_turns=np.asarray(probes, dtype=np.int64))
    tokens = np.zeros((len(chunks), N_NEW), dtype=np.int64)
    t0 = time.time()
    for ti, (a, b) in enumerate(chunks):
        pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
        ids = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        r = stream.turn(ids, pv, grid, max_new_tokens=N_NEW, repetition_penalty=PENALTY, force_length=True)
        tokens[ti] = r["new_tokens"]
        if ti in probes:
            lg = torch.stack(r["logits"]).float()
            top = lg.topk(TOPK, dim=-1)
            out[f"t{ti}_top_ids"] = top.indices.numpy().astype(np.int32)
            out[f"t{ti}_top_vals"] = top.values.numpy().astype(np.float32)
            out[f"t{ti}_scale"] = lg.abs().max(dim=-1).values.numpy().astype(np.float32)
            out[f"t{ti}_kv_before"] = np.int64(stream.past_ids.shape[1] - len(r["new_tokens"]) + 1 - len(ids))
        if ti % 20 == 0 or ti in probes:
            print(f"turn {ti}: kv {stream.past_ids.shape[1]}  ({time.time() - t0:.0f} s)", flush=True)
            np.savez_compressed(path, tokens=tokens, **out)
    out["tokens"] = tokens
    out["final_kv"] = np.int64(stream.past_ids.shape[1])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; final KV", int(out["final_kv"]), flush=True)


if __name__ == "__main__":
    generate()
