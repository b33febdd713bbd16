"""ORACLE / TEST INFRASTRUCTURE: tests/golden/livecc7b_batch8.npz -- the executed HF reference for ONE GPU's share of BASELINE.json configs[2]
at the REAL LiveCC-7B shapes: 8 streams (seeds 1234..1241, the benchmark's own streams), the 6-frame first turn + one 2-frame turn each,
4 greedy tokens per turn (repetition_penalty 1.05).

    python oracle/make_golden_7b_batch8.py        (build container: 62 GB of host RAM, 8 cores: ~3 min bf16 + ~15 min fp32)

The reference has no batched path: `distributed_generate_livecc.py` runs its streams one after the other per GPU, so the oracle for a batch
IS HF run once per stream.  HF bf16 (the reference's dtype) runs every stream free; HF fp32 on the same bf16-representable `tiled:0`
weights follows its tokens.  Stored per stream, turn and step: the bf16 run's top-64 raw logits (ids + values), both runs' values at those
ids and at 1,024 fixed sample ids, the fp32 logit scale and the full-vocabulary rms of (bf16 - fp32).  The GPU test
(tests/test_gpu_golden.py::test_livecc7b_batch_of_8_streams_against_the_committed_hf_logits) drives the NATIVE batched path -- one
`generate_batch` call per turn for the 8 streams: batched vision tower over 34,944 / 11,648 patches, packed prefill of 9,048 / 3,088 rows,
one weight pass per decode step for the 8 streams -- teacher-forced along HF's tokens, and compares every stream with its own HF run.
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

SEED0, STREAMS, N_FRAMES, H, W = 1234, 8, 8, 392, 728
N_NEW, PENALTY, TOPK, NSAMPLE = 4, 1.05, 64, 1024
PATH = os.path.join(ROOT, "tests", "golden", "livecc7b_batch8.npz")


def sample_ids(vocab: int) -> np.ndarray:
    return np.sort(np.random.default_rng(SEED0).choice(vocab, NSAMPLE, replace=False)).astype(np.int64)


def stream_inputs(cfg, s):
    frames = torch.from_numpy(protocol.synth_frames(N_FRAMES, H, W, seed=SEED0 + s, layout="TCHW"))
    return frames, protocol.TurnBuilder(cfg, seed=SEED0 + s)


def generate(path=PATH):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    sid = torch.from_numpy(sample_ids(cfg.vocab_size))
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    out = dict(meta=np.asarray([SEED0, STREAMS, N_FRAMES, H, W, N_NEW, 0], dtype=np.int64), sample_ids=sid.numpy())
    runs16 = []
    t0 = time.time()
    for s in range(STREAMS):
        frames, builder = stream_inputs(cfg, s)
        runs16.append(O.replay_stream(hf, cfg, frames, builder, N_NEW, PENALTY, max_turns=2))
        print(f"bf16 stream {s}: {[r['new_tokens'] for r in runs16[-1]]}  ({time.time() - t0:.0f} s)", flush=True)
    hf = hf.float()
    t0 = time.time()
    for s in range(STREAMS):
        frames, _ = stream_inputs(cfg, s)
        s32 = O.OracleStream(hf, cfg)
        for ti, (r, (a, b)) in enumerate(zip(runs16[s], protocol.split_clip(N_FRAMES))):
            pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
            r32 = s32.turn(r["turn_ids"], pv, grid, max_new_tokens=N_NEW, repetition_penalty=PENALTY, teacher_tokens=r["new_tokens"])
            l16, l32 = torch.stack(r["logits"]).float(), torch.stack(r32["logits"]).float()
            top = l16.topk(TOPK, dim=-1)
            k = f"s{s}_t{ti}"
            out[f"{k}_ids"] = np.asarray(r["turn_ids"], dtype=np.int64)
            out[f"{k}_tokens"] = np.asarray(r["new_tokens"], dtype=np.int64)
            out[f"{k}_top_ids"] = top.indices.numpy().astype(np.int32)
            out[f"{k}_top_vals_bf16"] = top.values.numpy().astype(np.float32)
            out[f"{k}_top_vals_fp32"] = torch.gather(l32, 1, top.indices).numpy().astype(np.float32)
            out[f"{k}_sample_vals_bf16"] = l16[:, sid].numpy().astype(np.float32)
            out[f"{k}_sample_vals_fp32"] = l32[:, sid].numpy().astype(np.float32)
            out[f"{k}_scale"] = l32.abs().max(dim=-1).values.numpy().astype(np.float32)
            out[f"{k}_rms_err_bf16_full_vocab"] = (l16 - l32).double().pow(2).mean(dim=-1).sqrt().numpy().astype(np.float32)
        print(f"fp32 stream {s} done ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(path, **out)          # partial results survive an interruption
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)
    return out


if __name__ == "__main__":
    generate()
