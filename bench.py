#!/usr/bin/env python
"""bench.py -- streaming commentary throughput of the LiveCC hot path on MI355X.

Workload (BASELINE.json configs[1]): LiveCC-7B shapes, one video stream per GPU replayed back-to-back, 2 fps,
60 frames at 392x728 (28x52 patch grid): one 6-frame turn + 27 two-frame turns, 16 greedy tokens per turn
(min_new_tokens = max_new_tokens, repetition_penalty 1.05) -> 448 commentary tokens, KV grows to ~12k.
Synthetic frames / prompt ids, random weights of the real architecture (no network: SURVEY 8d).

A "step" = one complete replay of the stream(s) on every rank (frames already resident in HBM).  One JSON line on
rank 0; `value` = commentary tokens/s summed over all streams of all GPUs.  `roofline` = the dominant kernel (decode
gate/up weight-streaming GEMV, HBM-bound) timed live with HIP events on its launch stream inside the timed region;
`cpu_baseline` = the HF CPU oracle (the reference's arithmetic) at the same 7B shapes on the host cores, one
streaming turn (bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="livecc-7b")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--height", type=int, default=392)
    ap.add_argument("--width", type=int, default=728)
    ap.add_argument("--streams-per-gpu", type=int, default=1)
    ap.add_argument("--max-new-tokens", type=int, default=16)
    ap.add_argument("--fused-tails", type=int, default=None, choices=[0, 1],
                    help="debug A/B: fuse add+rmsnorm / rope+append into the decode GEMV tails (default: library default)")
    ap.add_argument("--workload", choices=["stream60", "long480"], default="stream60",
                    help="long480 = BASELINE.json configs[3]: one 480-frame 280x280 video (24k visual tokens), 12 tokens per turn, "
                         "KV growing to ~32k; overrides --frames/--height/--width/--max-new-tokens")
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: LLM Linear weights as OCP e4m3 + fp32 row scales (BASELINE.json configs[4], 72B on one GPU)")
    ap.add_argument("--gemv-variant", type=int, default=None, help="debug A/B: lcc_debug_set_gemv_variant")
    ap.add_argument("--attn-variant", type=int, default=None, help="debug A/B: lcc_debug_set_attn_variant")
    ap.add_argument("--gemm-variant", type=int, default=None, help="debug A/B: lcc_debug_set_gemm_variant")
    ap.add_argument("--cpu-baseline", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--cpu-config", default=None, help="shapes of the CPU baseline (default: same as --config)")
    ap.add_argument("--cpu-budget", type=float, default=240.0, help="wall-clock budget of the CPU baseline leg, seconds")
    a = ap.parse_args()
    if a.workload == "long480":
        a.frames, a.height, a.width, a.max_new_tokens = 480, 280, 280, 12
    return a


def replay(model, cfg, frames_list, builders_seed, max_new, protocol, torch_mod):
    """One back-to-back replay of all local streams, batched turn by turn.  Returns (tokens, frames)."""
    n = len(frames_list)
    builders = [protocol.TurnBuilder(cfg, seed=s) for s in builders_seed]
    states = [None] * n
    past = [None] * n
    tokens = 0
    nframes = frames_list[0].shape[0]
    for ti, (a, b) in enumerate(protocol.split_clip(nframes)):
        reqs = []
        for i in range(n):
            clip = frames_list[i][a:b]
            grid = protocol.grid_of(clip.shape[0], clip.shape[1], clip.shape[2], cfg)
            new = builders[i].turn_ids(ti, protocol.num_video_tokens(grid, cfg))
            ids = new if past[i] is None else np.concatenate([past[i], new])
            reqs.append(dict(input_ids=torch_mod.from_numpy(ids), frames=clip, frames_layout="THWC", state=states[i]))
        outs = model.generate_batch(reqs, repetition_penalty=1.05, max_new_tokens=max_new, force_length=True)
        for i, o in enumerate(outs):
            states[i] = o.past_key_values
            seq = o.sequences[0].cpu().numpy()
            past[i] = seq[:-1]
            tokens += max_new
    for s in states:
        s.release()
    return tokens, n * nframes


def cpu_baseline(cfg_name, args, budget_s=240.0):
    """The reference's CPU path (HF generate, bf16, SDPA) on the host cores, one streaming turn at `cfg_name` shapes, run
    by oracle/cpu_baseline.py in a subprocess under a wall-clock budget.  The child reports every generated token, so a
    run cut by the budget still yields measured prefill time and decode rate (then the 16-token turn is extrapolated
    and the `sample` string says so)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--config", cfg_name, "--height", str(args.height),
           "--width", str(args.width), "--max-new-tokens", str(args.max_new_tokens)]
    t_start = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    ev = []
    import selectors
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
    if p.poll() is None:
        p.kill()          # exactly the child we started
    p.wait()
    info = next((e for e in ev if e["event"] == "start"), {})
    inputs = next((e for e in ev if e["event"] == "inputs"), {})
    built = next((e for e in ev if e["event"] == "built"), None)
    toks = [e for e in ev if e["event"] == "token"]
    done = next((e for e in ev if e["event"] == "done"), None)
    n = args.max_new_tokens
    # `cores` = the threads the baseline actually used (torch's intra-op pool), not the logical CPU count of the box
    base = dict(unit="tokens/s/stream", cores=info.get("threads", info.get("cores", os.cpu_count())), logical_cpus=info.get("cores"),
                kind="reference", cpu=info.get("cpu", ""), build_seconds=built["seconds"] if built else None)
    what = (f"HF transformers CPU path (bf16, sdpa, {info.get('threads', '?')} threads) at {cfg_name} shapes: one streaming turn = "
            f"ViT on 2 frames ({inputs.get('patches', '?')} patches) + {inputs.get('prompt_tokens', '?')}-token prefill + {n} greedy tokens")
    if done is not None:
        dt = done["t"]
        return dict(base, value=round(n / dt, 4), frames_per_s=round(2 / dt, 4), seconds=dt, sample=what)
    if len(toks) >= 2:
        prefill = toks[0]["t"]
        per_tok = (toks[-1]["t"] - toks[0]["t"]) / (len(toks) - 1)
        dt = prefill + per_tok * n
        return dict(base, value=round(n / dt, 4), frames_per_s=round(2 / dt, 4), seconds=round(dt, 2),
                    sample=what + f"; cut by the {budget_s:.0f}s budget after {len(toks)} tokens: prefill {prefill:.1f}s measured, "
                                  f"decode {per_tok:.2f}s/token measured, turn time extrapolated")
    return dict(base, value=None, sample=what + f"; nothing measurable inside the {budget_s:.0f}s budget "
                                                f"(built={built is not None}, tokens={len(toks)}, cut={cut}, events={ev[-3:]})")


def main():
    args = parse()
    from livecc_amd import distributed as D, protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    rank, local, world = D.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    cfg = get_config(args.config)
    spg = args.streams_per_gpu

    # weights: generated on rank 0, broadcast once over RCCL/xGMI -- no collective afterwards
    fp8 = args.weights == "fp8"
    arena = WeightArena(cfg, dev, llm_fp8=fp8)
    if rank == 0:
        arena.fill_random(seed=0)
    bcast_s = D.broadcast_weights(arena.flat, src=0)
    n_tok_turn = (args.height // 28) * (args.width // 28)
    kv_need = 32 * ((args.frames // 2 + 2) * (n_tok_turn + 64) // 32 + 4)
    if args.fused_tails is not None:
        from livecc_amd import _lib
        _lib.load().lcc_debug_set_fused_tails(args.fused_tails)
    if args.gemm_variant is not None:
        from livecc_amd import ops
        ops.set_gemm_variant(args.gemm_variant)
    if args.attn_variant is not None:
        from livecc_amd import ops
        ops.set_attn_variant(args.attn_variant)
    if args.gemv_variant is not None:
        from livecc_amd import ops
        ops.set_gemv_variant(args.gemv_variant)
    model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=spg, max_kv_len=min(32768, max(4096, kv_need)),
                                           max_new_rows=spg * (3 * n_tok_turn + 128), max_patches=spg * 12 * n_tok_turn + 64,
                                           max_history=max(16, args.max_new_tokens))
    frames = [torch.from_numpy(protocol.synth_frames(args.frames, args.height, args.width, seed=1234 + rank * spg + i)).to(dev)
              for i in range(spg)]
    seeds = [1234 + rank * spg + i for i in range(spg)]

    for _ in range(args.warmup):
        replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch)
    model.engine.profile(True, 8192)
    D.barrier(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    toks = nfr = 0
    for _ in range(args.steps):
        a, b = replay(model, cfg, frames, seeds, args.max_new_tokens, protocol, torch)
        toks += a
        nfr += b
    torch.cuda.synchronize(dev)
    D.barrier(dev)
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    model.engine.profile(False)
    total_tokens = D.sum_over_ranks(float(toks), dev)
    total_frames = D.sum_over_ranks(float(nfr), dev)
    if rank != 0:
        return
    ms = model.engine.profile_read(8192)
    # dominant kernel: gemv_skinny_kernel<2,2> (gate/up + SwiGLU), algorithmic bytes per launch:
    # weights 2I*H*2 + activations in M*H*2 + out M*I*2, M = streams per GPU
    I, H = cfg.intermediate_size, cfg.hidden_size
    alg_bytes = 2 * I * H * (1 if fp8 else 2) + (2 * I * 4 if fp8 else 0) + spg * H * 2 + spg * I * 2   # fp8: 1 B/weight + row scales
    roof = None
    if len(ms):
        avg_ms = float(np.mean(ms))
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tp):
            try:   # the committed PMC pass measured the LiveCC-7B bf16 single-stream launch; other shapes have no counter data
                if cfg.name == "livecc-7b" and spg == 1 and not fp8:
                    traffic = json.load(open(tp)).get("gemv_gate_up_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = dict(bound="hbm", kernel=("gemv_w8_kernel<2,2>" if fp8 else "gemv_skinny_kernel<2,2>") + " (decode gate/up + SwiGLU)", achieved=round(ach, 1), peak=8000.0,
                    unit="GB/s", frac=round(ach / 8000.0, 4), traffic=traffic, avg_launch_us=round(avg_ms * 1e3, 2),
                    launches_timed=int(len(ms)), algorithmic_bytes_per_launch=alg_bytes)
    cpu = None
    want_cpu = args.cpu_baseline == "on" or (args.cpu_baseline == "auto" and world == 1)
    if want_cpu:
        try:
            torch.cuda.empty_cache()
            cpu = cpu_baseline(args.cpu_config or args.config, args, args.cpu_budget)
            if cpu.get("value") is None and (args.cpu_config or args.config) != "qwen2vl-2b":
                # BASELINE.json configs[0]: the reference's own CPU-runnable case is Qwen2-VL-2B
                cpu = cpu_baseline("qwen2vl-2b", args, args.cpu_budget / 2)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = dict(value=None, unit="tokens/s/stream", cores=os.cpu_count(), kind="reference", sample=f"failed: {e!r}")
    n_streams = world * spg
    out = {
        "metric": f"commentary tokens/s (all streams) + frames/s ingested, {'LiveCC-7B' if cfg.name == 'livecc-7b' else cfg.name} streaming", "value": round(total_tokens / dt, 3),
        "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 (fp8 e4m3 LLM weights, bf16 MFMA)" if fp8 else "bf16", "data": "synthetic frames + synthetic prompt ids, random weights of the real architecture",
        "config": {"workload": f"{cfg.name} single stream per GPU, 2 fps, {args.frames} frames {args.height}x{args.width}, "
                               f"{args.max_new_tokens} tokens/turn, greedy, repetition_penalty 1.05"
                               + (" (BASELINE.json configs[1])" if cfg.name == "livecc-7b" and args.frames == 60 and args.workload == "stream60" and spg == 1 and not fp8 else "")
                               + (" (BASELINE.json configs[3]: 24k visual tokens, KV to ~32k)" if args.workload == "long480" and cfg.name == "livecc-7b" else "")
                               + (" (BASELINE.json configs[4])" if cfg.name == "qwen2vl-72b" and fp8 else ""),
                   "streams": n_streams, "streams_per_gpu": spg, "parallelism": f"dp{world} (streams sharded, weights broadcast)"},
        "tokens_per_s_per_stream": round(total_tokens / dt / n_streams, 3), "frames_per_s": round(total_frames / dt, 3),
        "weight_broadcast_s": round(bcast_s, 3), "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
