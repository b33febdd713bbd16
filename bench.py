#!/usr/bin/env python
"""bench.py -- streaming commentary throughput of the LiveCC hot path on MI355X.

Workload (BASELINE.json configs[1]): LiveCC-7B shapes, one video stream per GPU replayed back-to-back, 2 fps,
60 frames at 392x728 (28x52 patch grid): one 6-frame turn + 27 two-frame turns, 16 greedy tokens per turn
(min_new_tokens = max_new_tokens, repetition_penalty 1.05) -> 448 commentary tokens, KV grows to ~12k.
Synthetic frames / prompt ids, seeded synthetic weights of the real architecture (no network: SURVEY 8d).
`--streams-per-gpu 8 --gpus 8` is BASELINE.json configs[2] (64 streams data-parallel over 8 GPUs).

A "step" = one complete replay of the stream(s) on every rank (frames already resident in HBM).  One JSON line on
rank 0; `value` = commentary tokens/s summed over all streams of all GPUs.
  `roofline`     = the dominant kernel (decode gate/up weight-streaming GEMV, HBM-bound) timed live with HIP events on its
                   launch stream inside the timed region; `decode_step` next to it = whole decode steps (weights + KV bytes
                   over the measured step time) against the same 8 TB/s;
  `cpu_baseline` = the HF CPU path (the reference's arithmetic) at the same 7B shapes on the host cores through the same
                   protocol: the 6-frame turn + one 2-frame turn on the carried KV (bounded sample), prefill / decode rates;
  `parity`       = that CPU run teacher-forced along the native tokens with the SAME seeded weights: worst |dlogit| relative to
                   the logit scale, the number of steps where the native token is the CPU path's own argmax and -- default at
                   --gpus 1 (`--parity full`) -- the error against the fp32 truth relative to the bf16 reference's own
                   (`err_ratio_vs_fp32` worst logit, `rms_err_ratio_vs_fp32_*` over the whole vocabulary) plus `decisive`: greedy
                   token identity (32/32) on the synthetic weights whose top-1 margin is >= 10 x the bf16 noise.

`--gpus N` with N > 1 launches N ranks by itself (python -m torch.distributed.run, one process per GPU over RCCL) when it is not
already running under a launcher; under torchrun it asserts WORLD_SIZE == N.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="livecc-7b")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--height", type=int, default=392)
    ap.add_argument("--width", type=int, default=728)
    ap.add_argument("--streams-per-gpu", type=int, default=1, help="8 with --gpus 8 = BASELINE.json configs[2] (64 streams)")
    ap.add_argument("--max-new-tokens", type=int, default=16)
    ap.add_argument("--fused-tails", type=int, default=None, choices=[0, 1],
                    help="debug A/B: fuse add+rmsnorm / rope+append into the decode GEMV tails (default: library default)")
    ap.add_argument("--workload", choices=["stream60", "long480", "oneshot480"], default="stream60",
                    help="long480 = BASELINE.json configs[3]: one 480-frame 280x280 video (24k visual tokens), 12 tokens per turn, "
                         "KV growing to ~32k; oneshot480 = the OTHER half of configs[3]: the reference's video_qa / MCQ first turn "
                         "(ref demo/infer.py:182-242): all 480 frames in ONE generate call (ViT over 96,000 patches in 240 segments, "
                         "one 24k-row LLM prefill served in pieces, then 32 decode tokens at L ~ 24k); both override "
                         "--frames/--height/--width/--max-new-tokens")
    ap.add_argument("--time-budget-s", type=float, default=660.0,
                    help="target wall time of the whole run: the OPTIONAL blocks (live2fps history ladder, configs[3] / configs[4] children and their "
                         "fixture tests) are skipped -- with the reason in their block -- once the time left, after 130 s reserved for the CPU "
                         "baseline + bf16 parity leg, no longer covers them, and so are -- in this order of priority -- the fp32, decisive-weights and "
                         "HF-vs-HF legs of `--parity full`; the main line, roofline, cpu_baseline and the bf16 parity always run")
    ap.add_argument("--prefill-rows", type=int, default=4096, help="oneshot480: rows of one prefill launch sequence (max_new_rows)")
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: LLM Linear weights as OCP e4m3 + fp32 row scales (BASELINE.json configs[4], 72B on one GPU)")
    ap.add_argument("--gemv-variant", type=int, default=None, help="debug A/B: lcc_debug_set_gemv_variant")
    ap.add_argument("--attn-variant", type=int, default=None, help="debug A/B: lcc_debug_set_attn_variant")
    ap.add_argument("--gemm-variant", type=int, default=None, help="debug A/B: lcc_debug_set_gemm_variant")
    ap.add_argument("--decode-path", type=int, default=None, help="debug A/B: lcc_debug_set_decode_path")
    ap.add_argument("--skinny-rows", type=int, default=None, help="debug A/B: lcc_debug_set_skinny_rows (16 = round-3 routing of 17-64 decode "
                                                                  "streams through the 64-row GEMM tiles; default 64 = weight-streaming GEMVs)")
    ap.add_argument("--decode-chain", type=int, default=None, choices=[0, 1], help="debug A/B: lcc_debug_set_decode_chain")
    ap.add_argument("--fused-attn", type=int, default=None, help="debug A/B: lcc_debug_set_fused_attn (bit 0 fused decode attention for "
                                                                  ">= 16 (stream, KV head) pairs, bit 2 always, bit 1 in-launch split merge)")
    ap.add_argument("--main-stream-priority", choices=["default", "high"], default="default",
                    help="A/B: run the LLM phases on a high-priority HIP stream (the prefetched vision tower stays on the lowest-priority one)")
    ap.add_argument("--no-prefetch", action="store_true", help="A/B: do not overlap the next turn's vision tower with this turn's decode steps")
    ap.add_argument("--cpu-baseline", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--cpu-config", default=None, help="shapes of the CPU baseline (default: same as --config)")
    ap.add_argument("--cpu-budget", type=float, default=240.0, help="wall-clock budget of the CPU baseline leg, seconds")
    ap.add_argument("--parity", choices=["auto", "off", "bf16", "full"], default="auto",
                    help="compare the native path with the CPU reference leg on the same seeded weights (when the CPU leg runs at the "
                         "benchmarked shapes); full (= auto at --gpus 1, +~5 min of host time) also runs the fp32 truth for the error-ratio "
                         "test and the greedy token-identity check on the decisive synthetic weights; auto with --gpus > 1 = bf16")
    ap.add_argument("--share8", choices=["auto", "on", "off"], default="auto",
                    help="also time ONE GPU's share of BASELINE.json configs[2] (8 co-scheduled streams) in the same run and emit it as "
                         "`configs2_share` (auto: on for the default line -- LiveCC-7B, stream60, bf16, one stream per GPU, --gpus 1)")
    ap.add_argument("--live2fps", choices=["auto", "on", "off"], default="auto",
                    help="also measure the live-paced capacity of the multi-stream server (frames arriving at 2 fps, no look-ahead): p50 / p99 "
                         "arrival -> text latency per N and the largest N under a 1-s deadline, emitted as `live2fps` (auto: on for the default line)")
    ap.add_argument("--more-configs", choices=["auto", "on", "off"], default="auto",
                    help="also run BASELINE.json configs[3] (one-shot 480 frames) and configs[4] (72B fp8) as child runs + their committed-fixture parity "
                         "tests and emit them as `configs3_oneshot480` / `configs4_72b_fp8` (auto: on for the default line)")
    ap.add_argument("--standin", action="store_true",
                    help="launcher self-test: a stand-in model on CPU ranks over gloo (no GPU work, numbers meaningless)")
    a = ap.parse_args(argv)
    if a.parity == "auto":      # the driver's N = 1 line carries the full parity record (fp32 truth + decisive-weights token identity)
        a.parity = "full" if a.gpus == 1 else "bf16"
    if a.workload == "long480":
        a.frames, a.height, a.width, a.max_new_tokens = 480, 280, 280, 12
    if a.workload == "oneshot480":
        a.frames, a.height, a.width, a.max_new_tokens = 480, 280, 280, 32
    return a


# ------------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N starts N ranks by itself (ref evaluation/livesports3kcc/distributed_generate_livecc.py:105-122 fans out
# with local_mp; here one process per GPU under torch.distributed.run so that RCCL can broadcast the weights)
# ------------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["LCC_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class StandInModel:
    """Launcher self-test only (`--standin`): answers generate_batch with fixed tokens after a short sleep.  NOT a CPU path of the
    product -- it computes nothing."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.engine = None

    class _KV:
        def release(self):
            pass

    def generate_batch(self, reqs, max_new_tokens=16, **kw):
        time.sleep(0.002)
        outs = []
        for r in reqs:
            seq = torch.cat([torch.as_tensor(r["input_ids"]).view(-1), torch.arange(max_new_tokens) + 5]).view(1, -1)
            outs.append(type("O", (), {"sequences": seq, "past_key_values": r.get("state") or self._KV()})())
        return outs


def replay(model, cfg, frames_list, builders_seed, max_new, protocol, torch_mod, prefetch=True):
    """One back-to-back replay of all local streams, batched turn by turn.  Returns (tokens, frames).  `prefetch`: the frames are
    resident, so every turn hands the NEXT turn's clips to generate_batch, whose vision tower then runs on a side stream under this
    turn's decode steps (same work, same results, overlapped)."""
    n = len(frames_list)
    builders = [protocol.TurnBuilder(cfg, seed=s) for s in builders_seed]
    states = [None] * n
    past = [None] * n
    tokens = 0
    nframes = frames_list[0].shape[0]
    chunks = protocol.split_clip(nframes)
    for ti, (a, b) in enumerate(chunks):
        reqs = []
        for i in range(n):
            clip = frames_list[i][a:b]
            grid = protocol.grid_of(clip.shape[0], clip.shape[1], clip.shape[2], cfg)
            new = builders[i].turn_ids(ti, protocol.num_video_tokens(grid, cfg))
            ids = new if past[i] is None else np.concatenate([past[i], new])
            reqs.append(dict(input_ids=torch_mod.from_numpy(ids), frames=clip, frames_layout="THWC", state=states[i]))
        nxt = None
        if prefetch and ti + 1 < len(chunks):
            na, nb = chunks[ti + 1]
            nxt = [dict(frames=frames_list[i][na:nb], frames_layout="THWC") for i in range(n)]
        outs = model.generate_batch(reqs, repetition_penalty=1.05, max_new_tokens=max_new, force_length=True, **({"prefetch": nxt} if nxt else {}))
        for i, o in enumerate(outs):
            states[i] = o.past_key_values
            seq = o.sequences[0].cpu().numpy()
            past[i] = seq[:-1]
            tokens += max_new
    for s in states:
        s.release()
    return tokens, n * nframes


def replay_oneshot(model, cfg, frames_list, builders_seed, max_new, protocol, torch_mod):
    """The reference's video_qa / MCQ first turn (ref demo/infer.py:182-242, evaluation/distributed_mcq_predictor.py:72-105) for every
    local stream: the WHOLE clip + a query in one generate call -- the vision tower over every temporal slice, one long prefill
    (cut into max_new_rows pieces over the carried KV by the model), max_new forced greedy tokens.  Returns (tokens, frames)."""
    n = len(frames_list)
    reqs = []
    for i in range(n):
        clip = frames_list[i]
        grid = protocol.grid_of(clip.shape[0], clip.shape[1], clip.shape[2], cfg)
        b = protocol.TurnBuilder(cfg, seed=builders_seed[i])
        b.query_len = 24
        ids = b.turn_ids(0, protocol.num_video_tokens(grid, cfg), with_query=True)
        reqs.append(dict(input_ids=torch_mod.from_numpy(ids), frames=clip, frames_layout="THWC", state=None))
    outs = model.generate_batch(reqs, repetition_penalty=1.05, max_new_tokens=max_new, force_length=True)
    for o in outs:
        o.sequences[0, -1].item()          # the reference reads the answer back (processor.decode)
        o.past_key_values.release()
    return n * max_new, n * frames_list[0].shape[0]


ONESHOT_TEXT_IDS = 14 + 3 + 10 + 2 + 24 + 5      # TurnBuilder: system, user header, 'Time=a-bs', vision start/end, 24-id query, trailer


def oneshot_flops(cfg, nframes, height, width, n_text):
    """Algorithmic flops of one one-shot call up to its first token (SURVEY 8d formulas): ViT per temporal slice of n patches
    2 n (patch-embed + 32 blocks) + 32 * 4 n^2 * 1280 + merger; LLM prefill of S rows 2 * layer params * S + causal attention
    4 * S * (S / 2) * q_dim per layer + lm_head once."""
    n = (height // 14) * (width // 14)
    slices = (nframes + 1) // 2
    E, M, H = cfg.vit_embed_dim, cfg.vit_mlp_dim, cfg.hidden_size
    blk = E * 3 * E + E * E + 2 * E * M
    vit = slices * (2 * n * (cfg.patch_dim * E + cfg.vit_depth * blk) + cfg.vit_depth * 4 * n * n * E
                    + 2 * (n // 4) * (4 * E * 4 * E + 4 * E * H))
    S = slices * (n // 4) + n_text
    lin = cfg.hidden_size * cfg.qkv_dim + cfg.q_dim * cfg.hidden_size + 3 * cfg.hidden_size * cfg.intermediate_size
    llm = cfg.num_hidden_layers * (2 * lin * S + 4 * S * (S / 2) * cfg.q_dim) + 2 * cfg.vocab_size * H
    return dict(vit_flops=float(vit), llm_prefill_flops=float(llm), prefill_rows=int(S), patches=int(slices * n))


def stream_phase_flops(cfg, nframes, height, width, kept_per_turn, protocol):
    """Algorithmic flops of the MFMA-bound phases of ONE stream's replay (SURVEY 8d formulas), turn by turn: the vision tower over the
    turn's temporal slices (patch embed + blocks + per-slice attention + merger) and the LLM prefill of the turn's S new rows against the
    L keys already cached (Linear layers + causal attention, lm_head once per turn).  `kept_per_turn`: generated tokens that stay in the
    cache after a turn (max_new_tokens - 1: `past_ids = sequences[:, :-1]`)."""
    n = (height // 14) * (width // 14)
    E, M, H = cfg.vit_embed_dim, cfg.vit_mlp_dim, cfg.hidden_size
    blk = E * 3 * E + E * E + 2 * E * M
    lin = cfg.hidden_size * cfg.qkv_dim + cfg.q_dim * cfg.hidden_size + 3 * cfg.hidden_size * cfg.intermediate_size
    b = protocol.TurnBuilder(cfg, seed=0)
    vit = llm = 0.0
    kv = rows = 0
    for ti, (a, e) in enumerate(protocol.split_clip(nframes)):
        slices = (e - a + 1) // 2
        vit += slices * (2 * n * (cfg.patch_dim * E + cfg.vit_depth * blk) + cfg.vit_depth * 4 * n * n * E + 2 * (n // 4) * (4 * E * 4 * E + 4 * E * H))
        S = len(b.turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(e - a, height, width, cfg), cfg)))
        llm += cfg.num_hidden_layers * (2 * lin * S + 4 * S * (kv + S / 2) * cfg.q_dim) + 2 * cfg.vocab_size * H
        kv += S + kept_per_turn
        rows += S
    return dict(vit_flops=float(vit), llm_prefill_flops=float(llm), prefill_rows=int(rows), final_kv=int(kv))


def decode_step_roofline(cfg, engine, spg, kv_list, fp8):
    """Whole decode steps sampled by the engine (every 4th step, hipEvents on the launch stream): every Linear weight of the layers + lm_head
    once for the batch, plus each stream's KV read and its new KV row written, plus the logits (SURVEY 8d "decode step" formula), over the
    measured step time against 8 TB/s."""
    st_ms = engine.profile_read_steps(16384)
    if not len(st_ms):
        return None
    wbytes = (cfg.decode_weight_bytes() // 2) if fp8 else cfg.decode_weight_bytes()
    kv_avg = float(np.mean(kv_list)) if len(kv_list) else 0.0
    step_bytes = wbytes + spg * (kv_avg * cfg.kv_bytes_per_token + cfg.kv_bytes_per_token + cfg.vocab_size * 4)
    s_ms = float(np.mean(st_ms))
    ach = step_bytes / (s_ms * 1e-3) / 1e9
    out = dict(bound="hbm", unit="GB/s", peak=8000.0, achieved=round(ach, 1), frac=round(ach / 8000.0, 4), avg_step_us=round(s_ms * 1e3, 1),
               steps_timed=int(len(st_ms)), algorithmic_bytes_per_step=int(step_bytes), weight_bytes=int(wbytes), mean_kv_len=round(kv_avg, 1),
               us_per_layer=round((s_ms * 1e3) / cfg.num_hidden_layers, 2))
    # the vision tower of the NEXT turn runs on a side stream under the first decode steps of a turn (prefetch): steps late in the call
    # (index >= 8) see the GPU alone -- the kernel-efficiency number; the object itself is every sampled step
    try:
        rel = engine.profile_read_step_index(16384)[:len(st_ms)]
        late = np.asarray(st_ms)[rel >= 8]
        if len(late):
            l_ms = float(np.mean(late))
            out["late_steps_without_vision_tower_overlap"] = dict(avg_step_us=round(l_ms * 1e3, 1), steps_timed=int(len(late)),
                                                                  frac=round(step_bytes / (l_ms * 1e-3) / 1e9 / 8000.0, 4))
    except Exception:
        pass
    return out


def standalone_decode_steps(cfg, model, frames, seeds, args, protocol, spg, kv_list, fp8):
    """Decode steps WITHOUT a vision tower sharing the GPU: one extra replay with the prefetch off (outside every timed region), whole
    steps sampled by the engine as usual.  The `late_steps_without_vision_tower_overlap` heuristic (step index >= 8) holds only while the
    prefetched tower is shorter than eight decode steps -- true for 1-8 streams, false for 32, where the tower outlasts the whole decode
    phase (profiles/r04/step_breakdown_32streams_noprefetch.json: 6.06 ms standalone vs 9.8 ms "late" under the overlap)."""
    model.engine.profile(True, 16384)
    replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch, False)
    torch.cuda.synchronize()
    model.engine.profile(False)
    r = decode_step_roofline(cfg, model.engine, spg, kv_list, fp8)
    if r is None:
        return None
    r.pop("late_steps_without_vision_tower_overlap", None)
    out = dict(avg_step_us=r["avg_step_us"], frac=r["frac"], achieved=r["achieved"], us_per_layer=r["us_per_layer"], steps_timed=r["steps_timed"])
    k_ms = model.engine.profile_read(16384)          # the dominant kernel's own samples of this replay (one per decode step)
    if len(k_ms):
        out["dominant_kernel_avg_launch_us"] = round(float(np.mean(k_ms)) * 1e3, 2)
    return out


def configs2_share(cfg, make_model, dev, args, protocol, streams=8, steps=2, rank=0, world=1, D=None, bcast=None):
    """BASELINE.json configs[2] (64 streams data-parallel over 8 GPUs = 8 co-scheduled streams per GPU), timed in the same run as the
    configs[1] line.  EVERY rank runs its share: the same 60-frame replay with 8 streams batched turn by turn (batched vision tower,
    packed prefill, one weight pass per decode step for the 8 streams) between a barrier and a max-over-ranks clock -> tokens/s of the
    whole job, tokens/s/stream (minimum over ranks), frames/s; and the MFMA-bound roofline of the job: a second replay that stops after
    each turn's first token (vision tower + LLM prefill of every turn, no decode steps, no prefetch overlap), algorithmic flops
     of all ranks over the slowest rank's wall time against world x 2.5 PFLOP/s dense bf16.
    world == 1 (the driver's --gpus 1 line): `configs2_share`, ONE GPU's share.  world > 1: `configs2`, the job itself -- 8 x world
    streams, weights broadcast once over RCCL (`bcast`: seconds per rank + the xGMI bound), no collective in prefill / decode (ref
    evaluation/livesports3kcc/distributed_generate_livecc.py:46-50, 105-122: N processes, strided shards, nothing shared)."""
    n_tok_turn = (args.height // 28) * (args.width // 28)
    sync = (lambda: torch.cuda.synchronize(dev)) if torch.device(dev).type == "cuda" else (lambda: None)
    allmax = (lambda v: D.max_over_ranks(v, dev)) if D is not None else (lambda v: v)
    allsum = (lambda v: D.sum_over_ranks(v, dev)) if D is not None else (lambda v: v)
    gather = (lambda v: D.gather_floats(v, dev)) if D is not None else (lambda v: [float(v)])
    barrier = (lambda: D.barrier(dev)) if D is not None else (lambda: None)
    # Everything that can fail on ONE rank only (building the 8-stream model: HBM, its first replay) runs before the first collective, and
    # the ranks then agree on a success flag: if any rank failed, EVERY rank leaves with an error block instead of the healthy ranks blocking
    # in the next barrier for ever.
    err, model, frames, seeds = None, None, None, None
    # stream s of the 8 x world streams runs on rank s % world (ref distributed_generate_livecc.py:49-50: idxs[i::N]; distributed.shard_streams)
    from livecc_amd.distributed import shard_streams
    my_ids = shard_streams(range(streams * world), rank, world)
    assert len(my_ids) == streams
    try:
        model = make_model(streams)
        frames = [torch.from_numpy(protocol.synth_frames(args.frames, args.height, args.width, seed=1234 + sid)).to(dev) for sid in my_ids]
        seeds = [1234 + sid for sid in my_ids]
        replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch, True)          # warm-up
        sync()
    except Exception as e:          # noqa: BLE001
        err = repr(e)
    n_bad = allsum(1.0 if err else 0.0)
    if n_bad > 0:
        return dict(error=err or f"{int(n_bad)} other rank(s) failed to build / warm up their share; skipped on every rank", ranks_failed=int(n_bad))
    if model.engine is not None:
        model.engine.profile(True, 16384)
    barrier()
    sync()
    t0 = time.perf_counter()
    toks = nfr = 0
    for _ in range(steps):
        a, b = replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch, True)
        toks += a
        nfr += b
    sync()
    my_dt = time.perf_counter() - t0
    barrier()
    dt = allmax(time.perf_counter() - t0)
    step_roof = None
    if model.engine is not None:
        model.engine.profile(False)
        kv_list = kv_lengths_of_decode_steps(cfg, args.frames, args.height, args.width, args.max_new_tokens, protocol)
        step_roof = decode_step_roofline(cfg, model.engine, streams, kv_list, False)
        if step_roof is not None:
            step_roof["standalone_replay_without_prefetch"] = standalone_decode_steps(cfg, model, frames, seeds, args, protocol, streams, kv_list, False)
    # MFMA-bound phases alone: every turn's vision tower + prefill, one token per turn (no decode steps), no prefetch overlap
    replay(model, cfg, frames, seeds, 1, protocol, torch, False)
    barrier()
    sync()
    t1 = time.perf_counter()
    for _ in range(steps):
        replay(model, cfg, frames, seeds, 1, protocol, torch, False)
    sync()
    barrier()
    dt1 = allmax(time.perf_counter() - t1) / steps
    fl = stream_phase_flops(cfg, args.frames, args.height, args.width, 0, protocol)
    flops = world * streams * (fl["vit_flops"] + fl["llm_prefill_flops"])
    tf = flops / dt1 / 1e12
    total_toks, total_fr = allsum(float(toks)), allsum(float(nfr))
    per_rank = gather(toks / my_dt)
    ids_sum_per_rank = [int(round(x)) for x in gather(float(sum(my_ids)))]
    per_stream_min = min(per_rank) / streams
    del model, frames
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    n_all = streams * world
    out = dict(
        workload=(f"{cfg.name}, {n_all} streams = {streams} co-scheduled streams on each of {world} GPU(s), 2 fps, {args.frames} frames {args.height}x{args.width}, "
                  f"{args.max_new_tokens} tokens/turn, greedy" + (" = BASELINE.json configs[2]" if n_all == 64 else
                                                                   (" = one GPU's share of BASELINE.json configs[2] (64 streams over 8 GPUs)" if world == 1 else
                                                                    f" = BASELINE.json configs[2] scaled to {world} GPUs"))),
        streams=n_all, streams_per_gpu=streams, n_gpus=world, steps=steps, value=round(total_toks / dt, 2), unit="tokens/s",
        tokens_per_s_per_stream=round(per_stream_min, 2), tokens_per_s_per_stream_is="minimum over ranks",
        tokens_per_s_per_rank=[round(x, 2) for x in per_rank],
        stream_to_rank="stream s -> rank s % n_gpus (strided shards, as the reference's idxs[i::N])", stream_ids_rank0=list(my_ids),
        stream_id_sum_per_rank=ids_sum_per_rank,
        frames_per_s=round(total_fr / dt, 2), ms_per_replay=round(dt / steps * 1e3, 1),
        north_star_target_tokens_per_s_per_stream=30.0, meets_target=bool(per_stream_min >= 30.0),
        roofline=dict(bound="mfma", kernel=f"vision tower + LLM prefill of every turn (every GEMM / attention launch), {streams} streams batched per GPU; replay "
                                           "stopped after each turn's first token, no prefetch overlap",
                      achieved=round(tf, 1), peak=2500.0 * world, unit="TFLOP/s", frac=round(tf / (2500.0 * world), 4), traffic=None,
                      algorithmic_flops_per_replay=flops, vit_flops_per_stream=fl["vit_flops"], llm_prefill_flops_per_stream=fl["llm_prefill_flops"],
                      prefill_rows_per_stream=fl["prefill_rows"], seconds_per_replay=round(dt1, 4),
                      llm_gemm_mfma_util=llm_gemm_mfma_util_if_current(os.path.join(ROOT, "profiles", "roofline_traffic.json"))),
        decode_step=step_roof)
    if world == 1:
        out["meets_target_on_this_gpu"] = out["meets_target"]
        out["eight_gpu_job"] = ("not in this record (--gpus 1): `bench.py --gpus N` times it as `configs2` -- N ranks each running this share, weights "
                                "broadcast once over RCCL, no collective in prefill / decode")
    else:
        out["data_path_collectives"] = 0
        out["weight_broadcast"] = bcast
    return out


def live2fps(cfg, arena, dev, args, protocol, ladder=(8, 16, 32, 48, 64, 96, 112, 128), video_s=12.0, deadline_s=1.0, history_keys=0):
    """SURVEY 8f-2 under LIVE pacing (north_star: "concurrent 2 fps streams"; ref demo/infer.py:105-129, 165-175: one blocking generate per
    due chunk): N streams whose frames ARRIVE at 2 fps on the server's wall clock, `livecc_amd.server.StreamServer` batching whatever is
    due at each step, `max_new_tokens` forced greedy tokens per chunk, NO look-ahead (prefetch off: a chunk's vision tower starts only
    once its frames exist).  Stream i starts i / N s into the first second (no artificial phase alignment); its clock is shifted by one
    frame interval so that the server's pacing rule (a 2-frame chunk is due when the video clock passes its FIRST frame time) fires at
    the arrival of the chunk's LAST frame.  Latency of a streaming chunk = that arrival -> the chunk's text returned (its last token
    read back).  The 6-frame initial chunk (the reference takes 3 s at once) is reported apart.  Reports p50 / p99 / max per N and the
    largest N of the ladder whose p99 stays under `deadline_s` with every chunk served.
    `history_keys` > 0 (round 6, VERDICT r5 weak #7: capacity is a function of the history length): every stream JOINS with that many keys
    already in its KV slot (the slot length is set, `past_ids` is padded to match; the cached K / V values are whatever the arena holds --
    attention cost does not depend on them), so its first chunk is a continuing turn and every chunk's prefill / decode attention runs over
    history_keys + the live part: the capacity of a GPU whose streams have been live for ~history_keys / 400 seconds."""
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.server import StreamServer
    fti = protocol.FRAME_TIME_INTERVAL
    nfr = int(round(video_s / fti))
    pts = np.arange(nfr) * fti
    n_tok_turn = (args.height // 28) * (args.width // 28)
    n_chunks = 1 + (nfr - protocol.INITIAL_FPS_FRAMES) // protocol.STREAMING_FPS_FRAMES
    kv_need = 32 * (((n_chunks + 2) * (n_tok_turn + 64 + args.max_new_tokens) + history_keys) // 32 + 4)
    base_frames = [torch.from_numpy(protocol.synth_frames(nfr, args.height, args.width, seed=4321 + i)).to(dev) for i in range(8)]
    rows, capacity = [], None
    for N in ladder:
        model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=N, max_kv_len=min(32768, max(4096, kv_need)),
                                               max_new_rows=N * (3 * n_tok_turn + 128), max_patches=N * 12 * n_tok_turn + 64,
                                               max_history=max(16, args.max_new_tokens))
        srv = StreamServer(LiveCCDemoInfer(model=model), max_new_tokens=args.max_new_tokens, force_length=True, prefetch=False)
        if not rows:      # warm-up outside every clock: one initial chunk + one streaming chunk of a single stream
            srv.add_stream("w", base_frames[0], pts, t_start=0.0, max_pixels=args.height * args.width)
            srv.step(0.0); srv.step(3.01)
            srv.remove_stream("w")
        t_live = {}
        for i in range(N):
            f = base_frames[i % 8] if i < 8 else base_frames[i % 8].roll(i // 8, dims=0).contiguous()     # distinct content per stream
            t_live[i] = i / N
            srv.add_stream(i, f, pts, t_start=t_live[i] + fti, max_pixels=args.height * args.width)
            if history_keys > 0:
                stt = srv.streams[i]
                stt.kv = model.new_stream()
                stt.kv.rope_delta = 0
                model.engine.set_slot_length(stt.kv.slot, history_keys, history_keys)
                stt.past_ids = np.full(history_keys, 1000, dtype=np.int64)
        lat, lat_t, first, steps = [], [], [], []
        base = time.monotonic()
        inner = srv.step

        def timed_step(now, hf_spaces=False):
            t0 = time.monotonic()
            res = inner(now, hf_spaces)
            if res:
                steps.append((time.monotonic() - t0, len(res)))
            return res
        srv.step = timed_step

        def on_result(sid, span, text, state):
            t = time.monotonic() - base
            if span[0] == 0.0:                             # the initial 6-frame chunk: taken as buffered history the moment the stream joins
                first.append(t - (t_live[sid] + fti))      # (the reference's rule, ref demo/infer.py:107-110): join -> text
            else:
                lat.append(t - (t_live[sid] + span[1] - fti))      # arrival of the chunk's last frame on the stream's live clock -> text
                lat_t.append(t)
        torch.cuda.synchronize(dev)
        srv.run(until=video_s + 8.0, realtime=True, on_result=on_result, t0=base)
        wall = time.monotonic() - base
        served = len(lat) + len(first)
        expected = N * n_chunks
        for sid in list(srv.streams):
            srv.remove_stream(sid)
        # `srv.step = timed_step` closes over the bound method: a reference cycle (srv -> closure -> srv) that keeps the model -- up to
        # 120 GB of KV slots at the long-history ladder -- alive until the cyclic collector runs; break it and collect before the next model
        srv.step = inner
        del srv, model, timed_step, inner
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        a = np.sort(np.asarray(lat)) if lat else np.asarray([float("inf")])
        busy = float(sum(d for d, _ in steps))
        row = dict(streams=N, chunks_served=served, chunks_expected=expected, p50_s=round(float(np.percentile(a, 50)), 3),
                   p99_s=round(float(np.percentile(a, 99)), 3), max_s=round(float(a[-1]), 3),
                   initial_chunk_s_mean=round(float(np.mean(first)), 3) if first else None,
                   scheduler_steps=len(steps), mean_chunks_per_step=round(served / max(1, len(steps)), 2),
                   gpu_busy_frac=round(busy / wall, 3), tokens_per_s=round(served * args.max_new_tokens / wall, 1), wall_s=round(wall, 2))
        if video_s > 20.0 and lat:      # a soak run (tools/r5_live_soak.py): does the latency drift as the KV history grows?
            la, lt = np.asarray(lat), np.asarray(lat_t)
            row["p99_s_by_10s_window"] = [round(float(np.percentile(la[(lt >= w) & (lt < w + 10.0)], 99)), 3) if ((lt >= w) & (lt < w + 10.0)).any() else None
                                          for w in np.arange(0.0, video_s + 10.0, 10.0)]
        row["meets_deadline"] = bool(served == expected and row["p99_s"] <= deadline_s)
        rows.append(row)
        if row["meets_deadline"]:
            capacity = N
        if not row["meets_deadline"] and (served < expected or row["p99_s"] > 2 * deadline_s):
            break          # overloaded: larger N only queue longer
    return dict(
        history_keys_at_join=history_keys, history_keys_at_end=history_keys + n_chunks * (n_tok_turn + 24 + args.max_new_tokens) + 2 * n_tok_turn,
        workload=f"{cfg.name}, N live streams on ONE GPU, frames arriving at 2 fps ({args.height}x{args.width}), {video_s:.0f} s of video each = one 6-frame "
                 f"initial chunk + {n_chunks - 1} two-frame chunks, {args.max_new_tokens} forced greedy tokens per chunk, StreamServer continuous batching, "
                 "no look-ahead (vision-tower prefetch off)",
        latency_is="arrival of a 2-frame chunk's last frame -> its text returned (last token read back); streaming chunks only",
        deadline_s=deadline_s, ladder=rows, capacity_streams_under_deadline=capacity,
        eight_streams=next((r for r in rows if r["streams"] == 8), None))


def _sub_bench(extra, timeout_s):
    """Another workload of this file in a child process (its own model and arena; the parent keeps its HBM): the child's JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra) + ["--cpu-baseline", "off", "--parity", "off", "--share8", "off", "--live2fps", "off",
                                                                        "--more-configs", "off"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LCC_BENCH_SELF_LAUNCHED")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"child bench {extra} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-400:]}")
    return json.loads(lines[-1])


def _fixture_test(test_file, expr, keys, timeout_s):
    """One committed-fixture parity test of the GPU tier as a child pytest (the checker stays in tests/ + oracle/: bench.py only reads the
    record it writes): {passed, seconds, record of `keys`}."""
    tmp = tempfile.mkdtemp(prefix="lcc_fixture_")
    try:
        env = dict(os.environ, LCC_PARITY_OUT=tmp, LCC_ALLOW_BUDGET_SKIPS="1")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", test_file), "-m", "gpu", "-q", "-x", "-k", expr, "-p", "no:cacheprovider"],
                           capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        rec = {}
        rp = os.path.join(tmp, "parity_report.json")
        if os.path.exists(rp):
            allrec = json.load(open(rp))
            rec = {k: allrec[k] for k in keys if k in allrec}
            for v in rec.values():      # per-depth tables stay in the test's own report: the line carries their range
                dp = v.get("depth_profile") if isinstance(v, dict) else None
                if isinstance(dp, dict) and dp.get("rms_ratio"):
                    rr = [x for x, l in zip(dp["rms_ratio"], dp["layers"]) if l >= 0]
                    v["depth_profile"] = dict(layers_probed=len(rr), rms_ratio_min=min(rr), rms_ratio_max=max(rr))
        return dict(test=f"tests/{test_file} -k '{expr}'", passed=r.returncode == 0, seconds=round(time.perf_counter() - t0, 1),
                    tail=r.stdout.strip().splitlines()[-1][:160] if r.stdout.strip() else "", **rec)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def more_configs(args, left_s=lambda: 1e9, sections=None):
    """BASELINE.json configs[3] and configs[4] in the driver's own record: each as a child run of this file at its own
    shapes + the committed-fixture parity test of that configuration; a failure of either leaves an `error` in its block, never in the line.
    `left_s()` = seconds left for optional work (--time-budget-s): a block that no longer fits says so instead of running (typical costs on
    MI355X: one-shot child 35 s + its fixture test 15 s; 72B child 80 s + its two fixture tests 85 s)."""
    out = {}
    sections = sections if sections is not None else {}
    t_sec = time.perf_counter()

    def lap(name):
        nonlocal t_sec
        sections[name] = round(time.perf_counter() - t_sec, 1)
        t_sec = time.perf_counter()
    if left_s() < 60:
        out["configs3_oneshot480"] = dict(skipped=f"--time-budget-s: {left_s():.0f} s left for optional blocks, the one-shot child needs ~60")
        out["configs4_72b_fp8"] = dict(skipped=f"--time-budget-s: {left_s():.0f} s left for optional blocks, the 72B child needs ~100")
        return out
    try:
        j = _sub_bench(["--workload", "oneshot480", "--steps", "1", "--warmup", "1"], 600)
        rf = j.get("roofline") or {}
        blk = dict(workload=j["config"]["workload"], value=j["value"], unit=j["unit"], frames_per_s=j["frames_per_s"],
                   frames_per_s_to_first_token=rf.get("frames_per_s_to_first_token"), ms_to_first_token=rf.get("avg_call_ms"),
                   prefill_rows=rf.get("prefill_rows"), patches=rf.get("patches"),
                   roofline=dict(bound="mfma", kernel=rf.get("kernel"), achieved=rf.get("achieved"), peak=rf.get("peak"), unit=rf.get("unit"), frac=rf.get("frac"),
                                 traffic=None, algorithmic_flops_per_call=rf.get("algorithmic_flops_per_call")),
                   decode_step_at_24k_keys=((rf.get("decode_gate_up") or {}).get("decode_step")))
        out["configs3_oneshot480"] = blk
        lap("configs3_child")
        try:
            blk["parity"] = _fixture_test("test_gpu_golden.py", "oneshot480", ["livecc7b_oneshot480_vs_committed_golden"], 600)
        except Exception as e:
            blk["parity"] = dict(error=repr(e))
        lap("configs3_fixture_test")
    except Exception as e:
        out["configs3_oneshot480"] = dict(error=repr(e))
    if left_s() < 100:
        out["configs4_72b_fp8"] = dict(skipped=f"--time-budget-s: {left_s():.0f} s left for optional blocks, the 72B child needs ~100")
        return out
    try:
        j = _sub_bench(["--config", "qwen2vl-72b", "--weights", "fp8", "--steps", "1", "--warmup", "1"], 900)
        rf = j.get("roofline") or {}
        ds = rf.get("decode_step") or {}
        blk = dict(workload=j["config"]["workload"], value=j["value"], unit=j["unit"], frames_per_s=j["frames_per_s"], ms_per_replay=j["ms_per_step"],
                   mfma="bf16 (e4m3 weights expanded in registers)", dtype=j["dtype"],
                   roofline=dict(bound="hbm", kernel=rf.get("kernel"), achieved=rf.get("achieved"), peak=rf.get("peak"), unit=rf.get("unit"), frac=rf.get("frac"),
                                 traffic=None, avg_launch_us=rf.get("avg_launch_us"), algorithmic_bytes_per_launch=rf.get("algorithmic_bytes_per_launch")),
                   decode_step=dict(avg_step_us=ds.get("avg_step_us"), frac=ds.get("frac"), achieved=ds.get("achieved"), weight_bytes=ds.get("weight_bytes"),
                                    standalone_replay_without_prefetch=ds.get("standalone_replay_without_prefetch")))
        out["configs4_72b_fp8"] = blk
        lap("configs4_child")
        if left_s() < 100:
            blk["parity"] = dict(skipped=f"--time-budget-s: {left_s():.0f} s left; tests/test_gpu_layer_parity.py -k 72b (~85 s) is part of the GPU tier")
        else:
            try:
                blk["parity"] = _fixture_test("test_gpu_layer_parity.py", "72b", ["stream_72b_1layer_fp8", "qwen2vl72b_fp8_full_depth_vs_committed_golden"], 900)
            except Exception as e:
                blk["parity"] = dict(error=repr(e))
            lap("configs4_fixture_tests")
    except Exception as e:
        out["configs4_72b_fp8"] = dict(error=repr(e))
    return out


def kv_lengths_of_decode_steps(cfg, nframes, height, width, max_new, protocol):
    """KV length seen by every decode step of one stream replay (for the decode_step roofline's KV bytes)."""
    b = protocol.TurnBuilder(cfg, seed=0)
    kv, out = 0, []
    for ti, (a, e) in enumerate(protocol.split_clip(nframes)):
        grid = protocol.grid_of(e - a, height, width, cfg)
        kv += len(b.turn_ids(ti, protocol.num_video_tokens(grid, cfg)))
        for k in range(1, max_new):          # decode step k appends the k-th generated token and attends to kv + k keys
            out.append(kv + k)
        kv += max_new - 1                     # past_ids = sequences[:, :-1]
    return out


# ------------------------------------------------------------------------------------------------------------------------
# CPU reference leg (oracle/cpu_baseline.py in a subprocess under a wall-clock budget) + parity against it
# ------------------------------------------------------------------------------------------------------------------------
def run_cpu_leg(cfg_name, args, budget_s, turns, teacher=None, logits_out=None, dtype="bfloat16", weights="tiled:0", attn="sdpa", threads=0):
    import selectors
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--config", cfg_name, "--height", str(args.height),
           "--width", str(args.width), "--max-new-tokens", str(args.max_new_tokens), "--turns", str(turns), "--dtype", dtype,
           "--weights", weights, "--seed", "1234", "--attn", attn, "--threads", str(threads)]
    if teacher is not None:
        cmd += ["--teacher", teacher]
    if logits_out is not None:
        cmd += ["--logits-out", logits_out]
    t_start = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    ev = []
    sel = selectors.DefaultSelector()
    sel.register(p.stdout, selectors.EVENT_READ)
    cut = False
    while True:
        left = budget_s - (time.perf_counter() - t_start)
        if left <= 0:
            cut = True
            break
        if not sel.select(timeout=min(left, 5.0)):
            if p.poll() is not None:
                break
            continue
        line = p.stdout.readline()
        if not line:
            break
        try:
            ev.append(json.loads(line))
        except ValueError:
            pass
        if ev and ev[-1].get("event") == "done":
            break
    if p.poll() is None and (cut or not (ev and ev[-1].get("event") == "done")):
        p.kill()          # exactly the child we started
    p.wait()
    return ev, cut


def cpu_baseline(cfg_name, args, budget_s=240.0, turns=2, teacher=None, logits_out=None):
    """The reference's CPU path (HF generate, bf16, SDPA) on the host cores through the benchmark's own protocol: the 6-frame
    first turn and `turns - 1` two-frame turns on the carried KV.  The child reports every generated token, so a run cut by the
    budget still yields measured prefill time and decode rate (then the remaining tokens are extrapolated and `sample` says so)."""
    ev, cut = run_cpu_leg(cfg_name, args, budget_s, turns, teacher, logits_out)
    info = next((e for e in ev if e["event"] == "start"), {})
    built = next((e for e in ev if e["event"] == "built"), None)
    n = args.max_new_tokens
    base = dict(unit="tokens/s/stream", cores=info.get("threads", info.get("cores", os.cpu_count())), logical_cpus=info.get("cores"),
                kind="reference", cpu=info.get("cpu", ""), build_seconds=built["seconds"] if built else None)
    per_turn, total_t, total_tok, total_frames, notes = [], 0.0, 0, 0, []
    for ti in range(turns):
        inp = next((e for e in ev if e["event"] == "inputs" and e.get("turn") == ti), None)
        toks = [e for e in ev if e["event"] == "token" and e.get("turn") == ti]
        done = next((e for e in ev if e["event"] == "turn_done" and e.get("turn") == ti), None)
        if inp is None or not toks:
            break
        prefill = toks[0]["t"]
        if done is not None:
            dt, per_tok = done["t"], (done["t"] - prefill) / max(1, n - 1)
        elif len(toks) >= 2:
            per_tok = (toks[-1]["t"] - toks[0]["t"]) / (len(toks) - 1)
            dt = prefill + per_tok * (n - 1)
            notes.append(f"turn {ti} cut by the {budget_s:.0f}s budget after {len(toks)} tokens (remaining decode steps extrapolated)")
        else:
            break
        frames = 6 if ti == 0 else 2
        per_turn.append(dict(turn=ti, frames=frames, patches=inp["patches"], prompt_tokens=inp["prompt_tokens"], kv_before=inp["kv_before"],
                             prefill_s=round(prefill, 3), decode_s_per_token=round(per_tok, 4), turn_s=round(dt, 3)))
        total_t += dt; total_tok += n; total_frames += frames
    what = (f"HF transformers CPU path (bf16, sdpa, {info.get('threads', '?')} threads) at {cfg_name} shapes through the benchmark protocol: "
            + "; ".join(f"turn {t['turn']}: ViT on {t['frames']} frames ({t['patches']} patches) + {t['prompt_tokens']}-token prefill on "
                        f"{t['kv_before']} cached keys + {n} greedy tokens" for t in per_turn))
    if not per_turn:
        return dict(base, value=None, sample=what + f"; nothing measurable inside the {budget_s:.0f}s budget "
                                                     f"(built={built is not None}, cut={cut}, events={ev[-3:]})")
    return dict(base, value=round(total_tok / total_t, 4), frames_per_s=round(total_frames / total_t, 4), seconds=round(total_t, 2),
                prefill_tokens_per_s=round(sum(t["prompt_tokens"] for t in per_turn) / sum(t["prefill_s"] for t in per_turn), 2),
                decode_tokens_per_s=round(1.0 / (sum(t["decode_s_per_token"] for t in per_turn) / len(per_turn)), 4),
                turns=per_turn, sample=what + ("; " + "; ".join(notes) if notes else ""))


def native_parity_turns(model, cfg, args, protocol, turns, dev):
    """The same `turns` of stream 0 on the MI355X with raw logits kept: (tokens [turns, N], logits [turns, N, V] fp32)."""
    n_frames = 6 + 2 * (turns - 1)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, args.height, args.width, seed=1234)).to(dev)
    builder = protocol.TurnBuilder(cfg, seed=1234)
    state, past, toks, logits = None, None, [], []
    for ti, (a, b) in enumerate(protocol.split_clip(n_frames)):
        if ti >= turns:
            break
        grid = protocol.grid_of(b - a, args.height, args.width, cfg)
        new = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        ids = new if past is None else np.concatenate([past, new])
        r = model.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], frames_layout="THWC", past_key_values=state,
                           repetition_penalty=1.05, max_new_tokens=args.max_new_tokens, min_new_tokens=args.max_new_tokens,
                           output_logits=True, do_sample=False)
        state = r.past_key_values
        seq = r.sequences[0].cpu().numpy()
        past = seq[:-1]
        toks.append(seq[len(ids):].tolist())
        logits.append(r.logits.float().cpu().numpy())
    state.release()
    return np.asarray(toks), np.stack(logits)


def pmc_traffic_if_current(path):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/roofline_traffic.json) -- or None when
    the kernel's source file has changed since that pass (the file records its sha256: a counter of ANOTHER kernel version is not reported)."""
    import hashlib
    d = json.load(open(path))
    src, want = d.get("kernel_source"), d.get("kernel_source_sha16")
    if src and want:
        with open(os.path.join(ROOT, src), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest()[:16] != want:
                return None
    return d.get("gemv_gate_up_hbm_bytes_per_launch")


def llm_gemm_mfma_util_if_current(path):
    """T2 evidence: MfmaUtil of all four LLM GEMMs at M = 3088 / 386 + the FLOP-weighted figure per layer, from the committed
    rocprofv3 PMC pass (tools/r6_pmc_llm_gemms.py -> profiles/roofline_traffic.json) -- or None when csrc/gemm.hip has changed since."""
    import hashlib
    try:
        blk = json.load(open(path)).get("llm_gemm_mfma_util")
        if not blk:
            return None
        with open(os.path.join(ROOT, blk["gemm_source"]), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest()[:16] != blk["gemm_source_sha16"]:
                return None
        return dict(per_gemm={k: dict(mfma_util_pct=v["mfma_util_pct"], duration_us_under_pmc=v["duration_us_under_pmc"], kernel=v["kernel"], splits=v["splits"])
                              for k, v in blk["per_gemm"].items()},
                    flop_weighted=blk["per_layer"], source="profiles/roofline_traffic.json (rocprofv3 --pmc MfmaUtil; stale-source check passed)")
    except Exception:
        return None


def parity_report(native_tokens, native_logits, ref, ref32=None):
    """ref / ref32: npz of oracle/cpu_baseline.py (`logits` [turns, N, V], `own_argmax` [turns, N]).  Only complete turns count."""
    lg, own = ref["logits"], ref["own_argmax"]
    t = min(lg.shape[0], native_logits.shape[0])
    nl, lg, own, nt = native_logits[:t], lg[:t], own[:t], native_tokens[:t]
    scale = np.abs(lg).max(axis=-1)                                     # per step
    d16 = np.abs(nl - lg).max(axis=-1)
    out = dict(turns_compared=int(t), steps=int(d16.size), weights="identical seeded bf16 weights on both sides (tiled:0)",
               rel_dlogit_vs_bf16=round(float((d16 / scale).max()), 5), mean_rel_dlogit_vs_bf16=round(float((d16 / scale).mean()), 5),
               tokens_equal=int((nt == own).sum()), tokens_total=int(nt.size))
    # steps where the bf16 oracle's own top-1/top-2 RAW-logit margin exceeds twice the measured |native - oracle| difference: there the
    # raw argmaxes agree by construction; the sampled tokens are taken after the repetition penalty (x 1.05 on seen ids, the same on
    # both sides), so this is a reported statistic, not a theorem
    srt16 = np.sort(lg, axis=-1)
    dec16 = (srt16[..., -1] - srt16[..., -2]) > 2 * d16
    out["tokens_decided_vs_bf16"] = int(dec16.sum())
    out["tokens_equal_where_decided_vs_bf16"] = int(((nt == own) & dec16).sum())
    if ref32 is not None:
        l32 = ref32["logits"][:t]
        e_native = np.abs(nl - l32).max(axis=-1)
        e_ref = np.abs(lg - l32).max(axis=-1)
        s32 = np.abs(l32).max(axis=-1)
        out["err_ratio_vs_fp32"] = round(float((e_native / np.maximum(e_ref, 1e-3 * s32)).max()), 4)
        # the same comparison with a tight statistic: rms over the whole vocabulary (152k errors per step) instead of the single
        # worst logit, whose ratio fluctuates by +-15 % between two equally accurate implementations
        r_native = np.sqrt(((nl - l32).astype(np.float64) ** 2).mean(axis=-1))
        r_ref = np.sqrt(((lg - l32).astype(np.float64) ** 2).mean(axis=-1))
        out["rms_err_ratio_vs_fp32_worst_step"] = round(float((r_native / r_ref).max()), 4)
        out["rms_err_ratio_vs_fp32_all_steps"] = round(float(np.sqrt((r_native ** 2).mean() / (r_ref ** 2).mean())), 4)
        out["rel_err_native_vs_fp32"] = round(float((e_native / s32).max()), 5)
        out["rel_err_reference_bf16_vs_fp32"] = round(float((e_ref / s32).max()), 5)
        # a native token must be the fp32 argmax wherever the fp32 top-1/top-2 margin exceeds the bf16 reference's own error
        srt = np.sort(l32, axis=-1)
        margin = srt[..., -1] - srt[..., -2]
        decided = margin > 2 * e_ref
        out["tokens_decided_by_margin"] = int(decided.sum())
        out["tokens_equal_where_decided"] = int(((nt == ref32["own_argmax"][:t]) & decided).sum())
    return out


def hf_vs_hf_report(ref_a, ref_b, native_tokens, how):
    """The reference's OWN bf16 noise floor: the same two turns, same weights, same teacher tokens through HF bf16
    twice under different evaluation orders (`ref_a`: sdpa on all cores -- the `cpu_baseline` leg; `ref_b`: eager attention on fewer
    threads = another GEMM blocking and another softmax / accumulation order).  Same statistics as native-vs-HF next to it: what two runs
    of the REFERENCE differ by is the resolution at which any bf16 implementation can be compared with it."""
    la, lb = ref_a["logits"], ref_b["logits"]
    t = min(la.shape[0], lb.shape[0])
    la, lb = la[:t], lb[:t]
    scale = np.abs(la).max(axis=-1)
    d = np.abs(la - lb).max(axis=-1)
    oa, ob, nt = ref_a["own_argmax"][:t], ref_b["own_argmax"][:t], native_tokens[:t]
    return dict(how=how, turns_compared=int(t), steps=int(d.size), rel_dlogit_hf_vs_hf=round(float((d / scale).max()), 5),
                mean_rel_dlogit_hf_vs_hf=round(float((d / scale).mean()), 5), tokens_equal_hf_vs_hf=int((oa == ob).sum()), tokens_total=int(oa.size),
                rms_dlogit_hf_vs_hf=round(float(np.sqrt(((la - lb).astype(np.float64) ** 2).mean())), 5),
                native_tokens_equal_run_a=int((nt == oa).sum()), native_tokens_equal_run_b=int((nt == ob).sum()))


def decisive_report(native_tokens, native_logits, ref):
    """Greedy token identity on the `decisive` synthetic weights (livecc_amd/weights.py): the HF CPU path, teacher-forced along the
    native tokens, must prefer exactly the native token at every step; margin_over_noise = HF's own top-1/top-2 logit margin over
    the measured |native - HF| logit difference of that step (the decision is not made by rounding)."""
    lg, own = ref["logits"], ref["own_argmax"]
    t = min(lg.shape[0], native_logits.shape[0])
    nl, lg, own, nt = native_logits[:t], lg[:t], own[:t], native_tokens[:t]
    srt = np.sort(lg, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    noise = np.abs(nl - lg).max(axis=-1)
    mon = margin / np.maximum(noise, 1e-9)
    return dict(weights="decisive:0 (embedding-aligned lm_head, identical on both sides)", turns_compared=int(t), tokens_total=int(nt.size),
                tokens_equal=int((nt == own).sum()), min_margin_over_noise=round(float(mon.min()), 2),
                median_margin_over_noise=round(float(np.median(mon)), 2), steps_with_margin_over_10x_noise=int((mon >= 10).sum()),
                rel_dlogit_vs_bf16=round(float((noise / np.abs(lg).max(axis=-1)).max()), 5))


def main():
    t_bench0 = time.perf_counter()
    sections = {}            # wall seconds per block of this run (rank 0), reported as `sections_s`
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    from livecc_amd import distributed as D, protocol
    from livecc_amd.config import get_config
    rank, local, world = D.init_from_env(backend="gloo" if args.standin else None)
    numa = D.pin_to_gpu_numa_node(local) if (world > 1 and not args.standin) else dict(pinned=False, reason="single rank")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: pass --gpus {world} or let "
                         f"bench.py launch the ranks itself")
    cfg = get_config(args.config)
    spg = args.streams_per_gpu
    fp8 = args.weights == "fp8"
    n_tok_turn = (args.height // 28) * (args.width // 28)
    if args.standin:
        dev = torch.device("cpu")
        model, bcast_s, arena = StandInModel(cfg), 0.0, None
        t = torch.zeros(4)
        bcast_s = D.broadcast_weights(t if rank == 0 else t)      # exercises the broadcast path over gloo
    else:
        from livecc_amd.modeling import LiveCCForConditionalGeneration
        from livecc_amd.weights import WeightArena
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        dev = torch.device(f"cuda:{local}")
        torch.cuda.set_device(dev)
        # weights: generated on rank 0, broadcast once over RCCL/xGMI -- no collective afterwards
        arena = WeightArena(cfg, dev, llm_fp8=fp8)
        if rank == 0:
            arena.fill_tiled(seed=0)
        bcast_s = D.broadcast_weights(arena.flat, src=0)
        kv_need = 32 * ((args.frames // 2 + 2) * (n_tok_turn + 64) // 32 + 4)
        from livecc_amd import _lib, ops
        if args.fused_tails is not None:
            _lib.load().lcc_debug_set_fused_tails(args.fused_tails)
        if args.gemm_variant is not None:
            ops.set_gemm_variant(args.gemm_variant)
        if args.attn_variant is not None:
            ops.set_attn_variant(args.attn_variant)
        if args.gemv_variant is not None:
            ops.set_gemv_variant(args.gemv_variant)
        if args.decode_chain is not None:
            _lib.load().lcc_debug_set_decode_chain(args.decode_chain)
        if args.fused_attn is not None:
            _lib.load().lcc_debug_set_fused_attn(args.fused_attn)
        if args.decode_path is not None:
            _lib.check(_lib.load().lcc_debug_set_decode_path(args.decode_path), "lcc_debug_set_decode_path")
        if args.skinny_rows is not None:
            ops.set_skinny_rows(args.skinny_rows)
        if args.workload == "oneshot480":
            kv_total = (args.frames // 2) * n_tok_turn + 128 + args.max_new_tokens
            model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=spg, max_kv_len=32 * ((kv_total + 31) // 32) + 64,
                                                   max_new_rows=args.prefill_rows, max_patches=40 * 4 * n_tok_turn,
                                                   max_history=max(16, args.max_new_tokens))
        else:
            model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=spg, max_kv_len=min(32768, max(4096, kv_need)),
                                                   max_new_rows=spg * (3 * n_tok_turn + 128), max_patches=spg * 12 * n_tok_turn + 64,
                                                   max_history=max(16, args.max_new_tokens))
    frames = [torch.from_numpy(protocol.synth_frames(args.frames, args.height, args.width, seed=1234 + rank * spg + i)).to(dev)
              for i in range(spg)]
    seeds = [1234 + rank * spg + i for i in range(spg)]
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)

    pf = not args.no_prefetch
    oneshot = args.workload == "oneshot480"
    if oneshot:
        def run_once():
            return replay_oneshot(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch)
    else:
        def run_once():
            return replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch, pf)
    if args.main_stream_priority == "high" and dev.type == "cuda":
        # the LLM phases on a HIGH-priority stream: the prefetched vision tower of the next turn (lowest-priority side stream,
        # modeling._side_stream) then only takes what the decode kernels leave
        lo, hi = torch.cuda.Stream.priority_range()
        hp = torch.cuda.Stream(device=dev, priority=min(lo, hi))
        sync()
        inner = run_once

        def run_once():
            with torch.cuda.stream(hp):
                return inner()
    for _ in range(args.warmup):
        run_once()
    if model.engine is not None:
        model.engine.profile(True, 16384)
    D.barrier(dev)
    sync()
    t0 = time.perf_counter()
    toks = nfr = 0
    for _ in range(args.steps):
        a, b = run_once()
        toks += a
        nfr += b
    sync()
    my_dt = time.perf_counter() - t0
    D.barrier(dev)
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    if model.engine is not None:
        model.engine.profile(False)
    # the same replay WITHOUT the next turn's vision tower prefetched under this turn's decode steps (what a live 2-fps stream gets): reported
    # beside `value`, same barrier + max-over-ranks clock, one pass
    value_no_prefetch = None
    if pf and not oneshot:
        D.barrier(dev)
        sync()
        t_np = time.perf_counter()
        a_np, _ = replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch, False)
        sync()
        D.barrier(dev)
        dt_np = D.max_over_ranks(time.perf_counter() - t_np, dev)
        value_no_prefetch = D.sum_over_ranks(float(a_np), dev) / dt_np
    total_tokens = D.sum_over_ranks(float(toks), dev)
    total_frames = D.sum_over_ranks(float(nfr), dev)
    per_rank = D.gather_floats(toks / my_dt, dev)
    bcast_max = D.max_over_ranks(bcast_s, dev)
    # per-rank whole-decode-step time (ms) and broadcast time: an 8-GPU run shows at a glance whether every rank streams at the same rate
    my_step_ms = float(np.mean(model.engine.profile_read_steps(16384))) if model.engine is not None and not args.standin else 0.0
    if not np.isfinite(my_step_ms):
        my_step_ms = 0.0
    step_ms_per_rank = D.gather_floats(my_step_ms, dev)
    bcast_per_rank = D.gather_floats(bcast_s, dev)
    numa_pinned = D.sum_over_ranks(1.0 if numa.get("pinned") else 0.0, dev)
    # BASELINE.json configs[2]: world == 1 -> one GPU's share (`configs2_share`); world > 1 -> every rank runs its 8-stream share and the
    # line carries the job (`configs2`: 8 x world streams).  `value` stays the configs[1] replay above, so that N = 1 agrees with BENCH.
    share = None
    share_on = args.share8 == "on" or (args.share8 == "auto" and spg == 1 and args.workload == "stream60" and not fp8 and
                                       (args.standin or (cfg.name == "livecc-7b" and args.frames == 60)))
    if share_on:
        if args.standin:
            def make_model(streams):
                if os.environ.get("LCC_BENCH_FAIL_RANK") == str(rank):      # test hook: ONE rank fails while it builds its share
                    raise RuntimeError(f"injected failure on rank {rank}")
                return StandInModel(cfg)
        else:
            def make_model(streams):
                kv_need8 = 32 * ((args.frames // 2 + 2) * (n_tok_turn + 64) // 32 + 4)
                return LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=streams, max_kv_len=min(32768, max(4096, kv_need8)),
                                                      max_new_rows=streams * (3 * n_tok_turn + 128), max_patches=streams * 12 * n_tok_turn + 64,
                                                      max_history=max(16, args.max_new_tokens))
        bc = dict(seconds_max=round(bcast_max, 3), seconds_per_rank=[round(x, 3) for x in bcast_per_rank], arena_bytes=int(arena.nbytes()) if arena is not None else 0,
                  xgmi_bound_s=round(arena.nbytes() / 153e9, 3) if arena is not None else 0.0,
                  how="one chunked torch.distributed.broadcast of the flat arena from rank 0 (RCCL over xGMI, ~153 GB/s per link)")
        try:
            share = configs2_share(cfg, make_model, dev, args, protocol, rank=rank, world=world, D=D if world > 1 else None, bcast=bc)
        except Exception as e:       # never takes the main line down (every rank fails or none: the collectives inside are symmetric)
            share = dict(error=repr(e))
    if world > 1:
        D.shutdown(dev)          # all collectives are done: leave together (rank 0 goes on alone with the report)
    if rank != 0:
        return
    I, H = cfg.intermediate_size, cfg.hidden_size
    roof = step_roof = None
    if model.engine is not None:
        ms = model.engine.profile_read(16384)
        # dominant kernel: the decode gate/up weight-streaming GEMV (+ SwiGLU); algorithmic bytes per launch:
        # weights 2I*H*2 + activations in M*H*2 + out M*I*2, M = streams per GPU
        alg_bytes = 2 * I * H * (1 if fp8 else 2) + (2 * I * 4 if fp8 else 0) + spg * H * 2 + spg * I * 2   # fp8: 1 B/weight + row scales
        if len(ms):
            avg_ms = float(np.mean(ms))
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None
            tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tp):
                try:   # the committed PMC pass measured the LiveCC-7B bf16 single-stream launch; other shapes have no counter data
                    if cfg.name == "livecc-7b" and spg == 1 and not fp8:
                        traffic = pmc_traffic_if_current(tp)
                except Exception:
                    traffic = None
            roof = dict(bound="hbm", kernel=("fp8 " if fp8 else "") + "decode gate/up weight-streaming GEMV + SwiGLU", achieved=round(ach, 1),
                        peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4), traffic=traffic, avg_launch_us=round(avg_ms * 1e3, 2),
                        launches_timed=int(len(ms)), algorithmic_bytes_per_launch=alg_bytes)
        if oneshot:
            S0 = oneshot_flops(cfg, args.frames, args.height, args.width, ONESHOT_TEXT_IDS)["prefill_rows"]
            kv_list = [S0 + k for k in range(1, args.max_new_tokens)]
        else:
            kv_list = kv_lengths_of_decode_steps(cfg, args.frames, args.height, args.width, args.max_new_tokens, protocol)
        step_roof = decode_step_roofline(cfg, model.engine, spg, kv_list, fp8)
        if step_roof is not None and roof is not None:
            if not oneshot and not args.no_prefetch:
                sa = standalone_decode_steps(cfg, model, frames, seeds, args, protocol, spg, kv_list, fp8)
                step_roof["standalone_replay_without_prefetch"] = sa
                if sa and sa.get("dominant_kernel_avg_launch_us"):
                    # the same kernel timed live WITHOUT a vision tower sharing the GPU: the figure the rocprofv3 average of profiles/ (a
                    # --no-prefetch run) is comparable with; `achieved` / `frac` above stay the all-launches average of the timed region
                    us = sa.pop("dominant_kernel_avg_launch_us")
                    roof["without_tower_overlap"] = dict(avg_launch_us=us, achieved=round(alg_bytes / (us * 1e-6) / 1e9, 1),
                                                         frac=round(alg_bytes / (us * 1e-6) / 1e9 / 8000.0, 4))
            roof["decode_step"] = step_roof
        if oneshot:
            # the one-shot call is MFMA-bound up to its first token: the vision tower over every slice + the long prefill.  Timed live
            # with events around prefill-only calls (max_new_tokens = 1) on the launch stream; flops = SURVEY 8d formulas.
            fl = oneshot_flops(cfg, args.frames, args.height, args.width, ONESHOT_TEXT_IDS)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 2
            e0.record()
            for _ in range(reps):
                replay_oneshot(model, cfg, frames, seeds, 1, protocol, torch)
            e1.record()
            e1.synchronize()
            ms1 = e0.elapsed_time(e1) / reps
            tf = spg * (fl["vit_flops"] + fl["llm_prefill_flops"]) / (ms1 * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel="one-shot video_qa turn up to the first token: ViT over every temporal slice + the whole-clip LLM "
                                             "prefill (every GEMM / attention launch of the call)",
                        achieved=round(tf, 1), peak=2500.0, unit="TFLOP/s", frac=round(tf / 2500.0, 4), traffic=None,
                        avg_call_ms=round(ms1, 2), calls_timed=reps, algorithmic_flops_per_call=spg * (fl["vit_flops"] + fl["llm_prefill_flops"]),
                        vit_flops=fl["vit_flops"], llm_prefill_flops=fl["llm_prefill_flops"], prefill_rows=fl["prefill_rows"], patches=fl["patches"],
                        prefill_rows_per_launch_sequence=args.prefill_rows, frames_per_s_to_first_token=round(spg * args.frames / (ms1 * 1e-3), 1),
                        decode_gate_up=roof)
    sections["main_line_and_roofline"] = round(time.perf_counter() - t_bench0, 1)

    def left_s():            # seconds left for OPTIONAL blocks: the CPU baseline + parity legs (~110-130 s) always run at the end
        return args.time_budget_s - 130.0 - (time.perf_counter() - t_bench0)
    live = None
    want_live = args.live2fps == "on" or (args.live2fps == "auto" and world == 1 and spg == 1 and cfg.name == "livecc-7b" and args.workload == "stream60"
                                          and args.frames == 60 and not fp8)
    if want_live and not args.standin:
        try:
            del model
            torch.cuda.empty_cache()
            t_sec = time.perf_counter()
            live = live2fps(cfg, arena, dev, args, protocol, ladder=(8, 96, 112))
            sections["live2fps"] = round(time.perf_counter() - t_sec, 1)
            try:      # capacity @ history: the same ladder with 22k keys already cached per stream (a stream that has been live for a minute)
                if left_s() < 70:
                    raise TimeoutError(f"--time-budget-s: {left_s():.0f} s left for optional blocks, the three rungs need ~70")
                t_sec = time.perf_counter()
                lh = live2fps(cfg, arena, dev, args, protocol, ladder=(32, 48, 64), history_keys=22000)
                sections["live2fps_at_history"] = round(time.perf_counter() - t_sec, 1)
                live["at_long_history"] = dict(capacity_streams_under_deadline=lh["capacity_streams_under_deadline"], ladder=lh["ladder"],
                                                      history_keys_at_join=lh["history_keys_at_join"], history_keys_at_end=lh["history_keys_at_end"])
                live["capacity_at_history"] = {f"{live['history_keys_at_end']} keys": live["capacity_streams_under_deadline"],
                                               f"{lh['history_keys_at_end']} keys": lh["capacity_streams_under_deadline"]}
            except TimeoutError as e:
                live["at_long_history"] = dict(skipped=str(e))
            except Exception as e:
                live["at_long_history"] = dict(error=repr(e))
        except Exception as e:       # never takes the main line down
            live = dict(error=repr(e))
        model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=spg, max_kv_len=min(32768, max(4096, kv_need)),
                                               max_new_rows=spg * (3 * n_tok_turn + 128), max_patches=spg * 12 * n_tok_turn + 64,
                                               max_history=max(16, args.max_new_tokens))      # the parity legs below use it
    more = None
    if (args.more_configs == "on" or (args.more_configs == "auto" and want_live)) and not args.standin:
        try:
            del model
            import gc
            gc.collect()          # the child processes (72B fp8: 73 GB of weights + its KV) need the HBM the parent's models held
            torch.cuda.empty_cache()
            more = more_configs(args, left_s, sections)
        except Exception as e:
            more = dict(error=repr(e))
        model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=spg, max_kv_len=min(32768, max(4096, kv_need)),
                                               max_new_rows=spg * (3 * n_tok_turn + 128), max_patches=spg * 12 * n_tok_turn + 64,
                                               max_history=max(16, args.max_new_tokens))
    cpu = par = None
    t_cpu0 = time.perf_counter()
    want_cpu = (args.cpu_baseline == "on" or (args.cpu_baseline == "auto" and world == 1)) and not args.standin
    if want_cpu:
        cpu_cfg = args.cpu_config or args.config
        do_par = args.parity != "off" and cpu_cfg == args.config and not fp8
        tmp = tempfile.mkdtemp(prefix="lcc_parity_")
        try:
            torch.cuda.empty_cache()
            teacher = out16 = None
            if do_par:
                ntok, nlog = native_parity_turns(model, cfg, args, protocol, 2, dev)
                teacher, out16 = os.path.join(tmp, "teacher.npy"), os.path.join(tmp, "ref16.npz")
                np.save(teacher, ntok)
            cpu = cpu_baseline(cpu_cfg, args, args.cpu_budget, 2, teacher, out16)
            if do_par and os.path.exists(out16):
                # the extra legs of `--parity full`, most valuable first, each only while --time-budget-s still covers its typical cost on
                # the box's 128 cores (fp32 truth ~150 s, decisive weights ~90 s, the reference against itself ~130 s); a leg that does not
                # fit leaves `skipped` in the block (every one of them also exists as a committed fixture of the GPU tier)
                def fits(cost_s):
                    return args.time_budget_s - (time.perf_counter() - t_bench0) >= cost_s
                full = args.parity == "full"
                skipped = {}
                ref32 = None
                t_leg = time.perf_counter()
                if full and fits(170):
                    out32 = os.path.join(tmp, "ref32.npz")
                    run_cpu_leg(cpu_cfg, args, 4 * args.cpu_budget, 2, teacher, out32, dtype="float32")
                    ref32 = np.load(out32) if os.path.exists(out32) else None
                    sections["parity_fp32_leg"] = round(time.perf_counter() - t_leg, 1)
                elif full:
                    skipped["fp32_truth"] = "--time-budget-s: the HF fp32 leg (~150 s) no longer fits; tests/test_gpu_golden.py holds the committed fp32 fixtures"
                par = parity_report(ntok, nlog, np.load(out16), ref32)
                if full and fits(100):
                    # "token-id exact under greedy" (north_star) on weights where the argmax is decided by the model, not by rounding:
                    # the SAME arena refilled in place with the decisive variant, the same two turns, HF CPU teacher-forced
                    t_leg = time.perf_counter()
                    arena.fill_tiled(seed=0, variant="decisive")
                    dtok, dlog = native_parity_turns(model, cfg, args, protocol, 2, dev)
                    td, od = os.path.join(tmp, "teacher_d.npy"), os.path.join(tmp, "refd.npz")
                    np.save(td, dtok)
                    run_cpu_leg(cpu_cfg, args, 2 * args.cpu_budget, 2, td, od, weights="decisive:0")
                    par["decisive"] = decisive_report(dtok, dlog, np.load(od)) if os.path.exists(od) else dict(note="CPU leg cut by its budget")
                    sections["parity_decisive_leg"] = round(time.perf_counter() - t_leg, 1)
                elif full:
                    skipped["decisive"] = "--time-budget-s: the decisive-weights leg (~90 s) no longer fits; tests/test_gpu_golden.py -k decisive holds the committed fixture"
                if full and fits(150):
                    # the reference against ITSELF: a second bf16 leg under another evaluation order (eager attention, a quarter of the threads)
                    t_leg = time.perf_counter()
                    out16b = os.path.join(tmp, "ref16b.npz")
                    thr_b = max(8, (os.cpu_count() or 32) // 4)
                    run_cpu_leg(cpu_cfg, args, 2 * args.cpu_budget, 2, teacher, out16b, attn="eager", threads=thr_b)
                    if os.path.exists(out16b):
                        par["reference_noise_floor"] = hf_vs_hf_report(np.load(out16), np.load(out16b), ntok,
                                                                        f"HF bf16 sdpa / all cores vs HF bf16 eager / {thr_b} threads, same weights, same teacher tokens")
                        nf = par["reference_noise_floor"]
                        par["rel_dlogit_vs_bf16_over_reference_noise_floor"] = round(par["rel_dlogit_vs_bf16"] / max(nf["rel_dlogit_hf_vs_hf"], 1e-9), 3)
                    sections["parity_noise_floor_leg"] = round(time.perf_counter() - t_leg, 1)
                elif full:
                    skipped["reference_noise_floor"] = ("--time-budget-s: the HF-vs-HF leg (~130 s) no longer fits (run with --time-budget-s 900 for it; last "
                                                        "measured: profiles/r06/bench_time_budget_900_parity.json)")
                if skipped:
                    par["skipped"] = skipped
            elif do_par:
                par = dict(note="the CPU leg was cut by its budget before it could write its logits", turns_compared=0)
            if cpu.get("value") is None and cpu_cfg != "qwen2vl-2b":
                # BASELINE.json configs[0]: the reference's own CPU-runnable case is Qwen2-VL-2B
                cpu = cpu_baseline("qwen2vl-2b", args, args.cpu_budget / 2, 2)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = cpu or dict(value=None, unit="tokens/s/stream", cores=os.cpu_count(), kind="reference", sample=f"failed: {e!r}")
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    if want_cpu:
        sections["cpu_baseline_and_parity"] = round(time.perf_counter() - t_cpu0, 1)
    sections["total"] = round(time.perf_counter() - t_bench0, 1)
    n_streams = world * spg
    tag = ""
    if cfg.name == "livecc-7b" and args.frames == 60 and args.workload == "stream60" and not fp8:
        tag = " (BASELINE.json configs[1])" if spg == 1 else (" (BASELINE.json configs[2])" if n_streams == 64 else "")
    if args.workload == "long480" and cfg.name == "livecc-7b":
        tag = " (BASELINE.json configs[3]: 24k visual tokens, KV to ~32k)"
    if oneshot:
        tag = " ONE-SHOT: the whole clip in one generate call (ref demo/infer.py:182-242 video_qa first turn; BASELINE.json configs[3])"
    if oneshot:
        tag = " ONE-SHOT: the whole clip in one generate call (ref demo/infer.py:182-242 video_qa first turn; BASELINE.json configs[3])"
    if cfg.name == "qwen2vl-72b" and fp8:
        tag = " (BASELINE.json configs[4])"
    out = {
        "metric": f"commentary tokens/s (all streams) + frames/s ingested, {'LiveCC-7B' if cfg.name == 'livecc-7b' else cfg.name} streaming", "value": round(total_tokens / dt, 3),
        "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "value_no_prefetch": round(value_no_prefetch, 3) if value_no_prefetch is not None else None,
        "dtype": "bf16 (LLM Linear weights stored as fp8 e4m3 + fp32 row scales and expanded to bf16 in registers: every MFMA instruction is a bf16 one, "
                 "activations are never quantised)" if fp8 else "bf16",
        "data": "standin (launcher self-test, no GPU work)" if args.standin else "synthetic frames + synthetic prompt ids, seeded synthetic weights of the real architecture",
        "config": {"workload": f"{cfg.name} {spg} stream(s) per GPU, 2 fps, {args.frames} frames {args.height}x{args.width}, "
                               f"{args.max_new_tokens} tokens/{'call' if oneshot else 'turn'}, greedy, repetition_penalty 1.05" + tag +
                               (", weights stored as fp8 e4m3 + fp32 row scales and EXPANDED TO bf16 IN REGISTERS: every MFMA is a bf16 one (not an fp8-MFMA path)" if fp8 else "") +
                               (", M-RoPE text offset after the vision block: transformers-4.5x rule (`hf4`, the product default) -- restated from memory, UNPINNED: no 4.5x "
                                "wheel exists offline to execute it against" if (oneshot and getattr(model, "text_offset_rule", "") == "hf4") else ""),
                   "streams": n_streams, "streams_per_gpu": spg, "parallelism": f"dp{world} (streams sharded, weights broadcast)",
                   "mfma": "bf16 x bf16 -> fp32" + (" (fp8 weights expanded in registers; no fp8 MFMA instruction is issued: on gfx950 only the MX block-scaled "
                                                    "forms run faster than bf16 and they need activation scales the reference arithmetic does not have)" if fp8 else "")},
        "tokens_per_s_per_stream": round(total_tokens / dt / n_streams, 3), "frames_per_s": round(total_frames / dt, 3),
        "weight_broadcast_s": round(bcast_max, 3), "rccl_ranks": world, "launcher": "self" if os.environ.get("LCC_BENCH_SELF_LAUNCHED") else ("torchrun" if world > 1 else "single"),
        "tokens_per_s_per_rank": [round(x, 2) for x in per_rank],
        "decode_step_ms_per_rank": [round(x, 3) for x in step_ms_per_rank],
        "weight_broadcast_s_per_rank": [round(x, 3) for x in bcast_per_rank],
        "weight_broadcast_xgmi_bound_s": round(arena.nbytes() / 153e9, 3) if (arena is not None and world > 1) else 0.0,
        "ranks_pinned_to_gpu_numa_node": int(numa_pinned),
        "sections_s": sections, "roofline": roof, "cpu_baseline": cpu, "parity": par, ("configs2_share" if world == 1 else "configs2"): share, "live2fps": live, **(more or {}),
        "timed_region": "frames resident in HBM as uint8 (resize / H2D outside the timed region); back-to-back replay: the NEXT turn's vision tower "
                        "is prefetched on a side stream under this turn's decode steps" + (" (disabled: --no-prefetch)" if args.no_prefetch else "") +
                        " -- available to a replay / a server that fetches ahead, NOT to a live 2-fps stream whose next frames do not exist yet",
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
