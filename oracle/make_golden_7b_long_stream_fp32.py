"""ORACLE / TEST INFRASTRUCTURE: tests/golden/livecc7b_long480_stream_fp32.npz -- the fp32 truth for the STREAMING form of BASELINE.json
configs[3] (VERDICT r4 weak #1b / next #9: the long-stream fixture had no fp32 leg, so the incremental path over a deep cache was checked only
by the gross-error bound + tokens).

    python oracle/make_golden_7b_long_stream_fp32.py [--upto 120]      (build container: 62 GB of host RAM, 8 cores: ~1.5 h)

The first `UPTO + 1` turns of the committed 480-frame stream (tests/golden/livecc7b_long480_stream.npz: a 6-frame turn + two-frame turns, 12
tokens per turn, KV growing by ~138 keys per turn), teacher-forced along its committed tokens, through HF `generate` TWICE on the same
seeded weights: bf16 (the reference's dtype -- its raw logits are re-computed here at the sample ids) and fp32 (the truth, its own fp32 KV
cache built turn by turn).  Stored for the probe turns (60 and 120: ~8.3k / ~16.3k cached keys) per step: both runs at 4,096 fixed sample
ids, the fp32 scale and top-2, the full-vocabulary rms of (bf16 - fp32).  The GPU test follows the same tokens and asserts
rms(native - fp32) <= 1.25 x rms(HF_bf16 - fp32) on every probe step.
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
from oracle import make_golden_7b_long_stream as S  # noqa: E402
from oracle.make_golden_7b_long import sample_ids  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "livecc7b_long480_stream_fp32.npz")
PROBES = (60, 120)


def run(hf, cfg, frames, tokens, upto, probes, tag, t_start):
    builder = protocol.TurnBuilder(cfg, seed=S.SEED)
    stream = O.OracleStream(hf, cfg)
    got = {}
    for ti, (a, b) in enumerate(protocol.split_clip(S.N_FRAMES)[:upto + 1]):
        pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
        ids = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        r = stream.turn(ids, pv, grid, max_new_tokens=S.N_NEW, repetition_penalty=S.PENALTY, force_length=True, teacher_tokens=[int(x) for x in tokens[ti]])
        assert r["new_tokens"] == [int(x) for x in tokens[ti]]
        if ti in probes:
            got[ti] = torch.stack(r["logits"]).float()
        if ti % 10 == 0 or ti in probes:
            print(f"{tag} turn {ti}: kv {stream.past_ids.shape[1]}  ({time.time() - t_start:.0f} s)", flush=True)
    return got


def generate(upto=max(PROBES), path=PATH):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = get_config("livecc-7b")
    g = dict(np.load(S.PATH))
    tokens = g["tokens"]
    probes = [p for p in PROBES if p <= upto]
    frames = torch.from_numpy(protocol.synth_frames(S.N_FRAMES, S.H, S.W, seed=S.SEED, layout="TCHW"))
    sid = torch.from_numpy(sample_ids(cfg.vocab_size))
    t0 = time.time()
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    l16 = run(hf, cfg, frames, tokens, upto, probes, "bf16", t0)
    out = dict(meta=np.asarray([S.SEED, S.N_FRAMES, S.H, S.W, S.N_NEW, 0, upto], dtype=np.int64), probe_turns=np.asarray(probes, dtype=np.int64),
               sample_ids=sid.numpy())
    for ti in probes:       # the committed fixture's top-64 of these turns must be this run's values (same HF, same weights, same tokens)
        top_ids = torch.from_numpy(g[f"t{ti}_top_ids"].astype(np.int64))
        again = torch.gather(l16[ti], 1, top_ids).numpy()
        assert np.abs(again - g[f"t{ti}_top_vals"]).max() <= 1e-6 + 2.0 ** -6 * float(g[f"t{ti}_scale"].max()), f"turn {ti}: the bf16 rerun left the committed stream"
        out[f"t{ti}_sample_vals_bf16"] = l16[ti][:, sid].numpy().astype(np.float32)
    np.savez_compressed(path, **out)
    hf = hf.float()
    l32 = run(hf, cfg, frames, tokens, upto, probes, "fp32", t0)
    for ti in probes:
        a, t = l16[ti], l32[ti]
        t2 = t.topk(2, dim=-1)
        out[f"t{ti}_sample_vals_fp32"] = t[:, sid].numpy().astype(np.float32)
        out[f"t{ti}_scale"] = t.abs().max(dim=-1).values.numpy().astype(np.float32)
        out[f"t{ti}_fp32_top2_ids"] = t2.indices.numpy().astype(np.int64)
        out[f"t{ti}_fp32_top2_vals"] = t2.values.numpy().astype(np.float32)
        out[f"t{ti}_rms_err_bf16_full_vocab"] = (a - t).double().pow(2).mean(dim=-1).sqrt().numpy().astype(np.float32)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", {ti: out[f"t{ti}_rms_err_bf16_full_vocab"].round(4).tolist() for ti in probes}, flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    generate(int(a[a.index("--upto") + 1]) if "--upto" in a else max(PROBES))
