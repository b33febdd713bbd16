"""ORACLE / TEST INFRASTRUCTURE: CPU reference leg of bench.py and of the full-shape parity tests (never imported by the product).

Runs the reference's arithmetic -- HF `Qwen2VLForConditionalGeneration.generate` (ref demo/infer.py:165-172), SDPA attention, on
the host cores -- through the reference's streaming protocol at the given shapes:

    turn 0 : 6 frames (ViT over 3 temporal slices) + first prompt + N greedy tokens          ref demo/infer.py:121-129
    turn 1 : 2 frames + '<|im_end|>\\n...' continuation prompt on the CARRIED KV + N tokens    ref demo/infer.py:159-174
    ...

Run as a subprocess; prints one JSON object per line as it progresses (build, every generated token) so that the parent can
report a measured or a partially extrapolated rate inside a fixed wall-clock budget.

Weights: `--weights tiled:<seed>` = the seeded synthetic model of `livecc_amd.weights.fill_hf_model_tiled` -- bit-identical to
`WeightArena.fill_tiled(seed)` on the GPU, so the logits of this process ARE the oracle for the native path at full shapes;
`--weights decisive:<seed>` = its variant with decisive top-1 margins (livecc_amd/weights.py: DECISIVE_*), on which greedy token
identity is a hard test.
Parity mode: `--teacher tokens.npy` teacher-forces the generation along the native tokens ([turns, N] int) and `--logits-out`
receives the raw lm_head logits of every step ([turns, N, V] fp32) plus this model's own argmax before forcing.
`--dtype float32` = the fp32 truth for the error-ratio test (same bf16-representable weights).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit(**kw):
    print(json.dumps(kw), flush=True)


T_IMPORT = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="livecc-7b")
    ap.add_argument("--height", type=int, default=392)
    ap.add_argument("--width", type=int, default=728)
    ap.add_argument("--max-new-tokens", type=int, default=16)
    ap.add_argument("--turns", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1234, help="frames / prompt ids seed (bench stream 0)")
    ap.add_argument("--weights", default="tiled:0")
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--teacher", default=None)
    ap.add_argument("--logits-out", default=None)
    ap.add_argument("--attn", default="sdpa", choices=["sdpa", "eager"], help="HF attention implementation (the bench's second bf16 leg runs "
                    "`eager` on fewer threads: the reference's own bf16 arithmetic under another evaluation order = its noise floor)")
    ap.add_argument("--threads", type=int, default=0, help="torch intra-op threads (0 = torch's default: all cores)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from oracle import hf_oracle as O
    cfg = get_config(a.config)
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    dtype = getattr(torch, a.dtype)
    cores = os.cpu_count() or 1
    emit(event="imported", seconds=round(time.perf_counter() - T_IMPORT, 2))
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    emit(event="start", cores=cores, threads=torch.get_num_threads(), cpu=cpu_model, config=cfg.name, dtype=a.dtype, attn=a.attn)
    t0 = time.perf_counter()
    m = O.build_hf_model_synthetic(cfg, dtype, a.weights, attn_implementation=a.attn)
    emit(event="built", seconds=round(time.perf_counter() - t0, 2))

    n_frames = 6 + 2 * (a.turns - 1)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, a.height, a.width, seed=a.seed, layout="TCHW"))
    builder = protocol.TurnBuilder(cfg, seed=a.seed)
    teacher = np.load(a.teacher) if a.teacher else None
    stream = O.OracleStream(m, cfg)
    logits_all, own_all, tok_all = [], [], []
    t_run0 = time.perf_counter()
    for ti, (fa, fb) in enumerate(protocol.split_clip(n_frames)):
        if ti >= a.turns:
            break
        pv, grid = O.patchify_normalize_ref(frames[fa:fb], cfg)
        ids = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        emit(event="inputs", turn=ti, patches=int(pv.shape[0]), prompt_tokens=int(len(ids)), kv_before=0 if stream.past_ids is None else int(stream.past_ids.shape[1]))
        t_turn = time.perf_counter()
        n_prompt = (0 if stream.past_ids is None else stream.past_ids.shape[1]) + len(ids)

        def tick(input_ids, scores, _t=t_turn, _ti=ti, _np=n_prompt):
            emit(event="token", turn=_ti, i=int(input_ids.shape[1] - _np), t=round(time.perf_counter() - _t, 4))
        r = stream.turn(ids, pv, grid, max_new_tokens=a.max_new_tokens, repetition_penalty=1.05, force_length=True,
                        teacher_tokens=None if teacher is None else teacher[ti].tolist(), tick=tick)
        emit(event="turn_done", turn=ti, t=round(time.perf_counter() - t_turn, 4), tokens=r["new_tokens"])
        logits_all.append(torch.stack(r["logits"]).numpy())
        own_all.append([int(s.argmax()) for s in r["scores"]])
        tok_all.append(r["new_tokens"])
    if a.logits_out:
        np.savez(a.logits_out, logits=np.stack(logits_all), own_argmax=np.asarray(own_all), tokens=np.asarray(tok_all))
    emit(event="done", t=round(time.perf_counter() - t_run0, 4), turns=len(tok_all))


if __name__ == "__main__":
    main()
