"""Per-layer parity (-m gpu): every vision block and every decoder layer of the native engine compared IN ISOLATION with HF's.

VERDICT r2 #1b / SURVEY section 7 step 1.  The end-to-end logit tests say how far the HIP path ends up from the fp32 truth after
28 layers; they cannot say WHERE an error comes from.  Here every layer is teacher-forced: HF bf16 runs once and its per-layer inputs
are recorded (oracle/layer_probe.py); HF fp32 and the native engine (lcc_debug_set_llm_taps / lcc_debug_set_vit_taps: input of layer l
replaced by the oracle's) then both compute layer l on exactly that input.  Per layer and stage:

        ratio_rms = rms(native_out - fp32_out) / rms(hf_bf16_out - fp32_out)

Bound: ratio_rms <= 1.25 for every decoder layer (after attention and after the MLP), every vision block, PatchEmbed, PatchMerger and
the final norm + lm_head -- the native arithmetic of each stage is as close to the truth as the reference's own bf16 arithmetic, layer
by layer (target of the round: <= 1.05 at LiveCC-7B shapes).  The rms over a whole [rows, hidden] tensor is a tight statistic (10^5-10^6
elements); the max-norm ratios are recorded too (tests/util.record -> parity_report.json) but a maximum over 10^6 rounding errors of
two independent implementations fluctuates by +-15 % on its own.
"""
import os

import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu

RATIO_BOUND = 1.25


def native_probe(native, cfg, ids, frames_dev, ref16, layout="TCHW"):
    """One prefill of the native engine with every layer's input replaced by the bf16 oracle's; returns the same keys as
    oracle.layer_probe.probe (tensors on CPU)."""
    eng = native.engine
    S, L, D = len(ids), cfg.num_hidden_layers, cfg.vit_depth
    P = ref16["vit_in"].shape[1]
    vov = torch.cat([ref16["vit_in"], ref16["merger_in"].reshape(1, P, -1)]).to(torch.bfloat16)
    lov = torch.cat([ref16["llm_in"], ref16["final_in"].reshape(1, S, -1)]).to(torch.bfloat16)
    vt = eng.set_vit_taps(P, vov)
    lt = eng.set_llm_taps(S, lov)
    try:
        r = native.generate(input_ids=torch.from_numpy(np.asarray(ids)).view(1, -1), frames=frames_dev, frames_layout=layout,
                            max_new_tokens=1, output_logits=True, do_sample=False)
        torch.cuda.synchronize()
        vt, lt = vt.cpu(), lt.cpu()
        logits = r.logits[0].float().cpu()
        r.past_key_values.release()
    finally:
        eng.set_vit_taps(0)
        eng.set_llm_taps(0)
    return dict(patch_embed=vt[0], vit_out=vt[1:], embeds=lt[0], llm_mid=lt[1::2], llm_out=lt[2::2], logits=logits)


def _compare(name, nat, a16, t32, cfg, ids):
    from oracle import layer_probe as P
    rows = P.layer_error_table(nat, a16, t32, ["patch_embed", "vit_out", "llm_mid", "llm_out", "logits"])
    # PatchMerger: the video rows of the native embeddings (tap 0) against the oracle's merger output on the same merger input
    vid = torch.from_numpy(np.asarray(ids) == cfg.video_token_id)
    mg = dict(merger=nat["embeds"][vid])
    rows += P.layer_error_table(mg, dict(merger=a16["vit_merged"]), dict(merger=t32["vit_merged"]), ["merger"])
    # text rows of the embeddings are a gather: bit-exact
    assert torch.equal(nat["embeds"][~vid].float(), a16["embeds"][~vid].float()), "embedding gather must be bit-exact"
    worst = max(rows, key=lambda r: r["ratio_rms"])
    by_stage = {}
    for r in rows:
        by_stage.setdefault(r["stage"], []).append(r["ratio_rms"])
    summary = {k: dict(worst=max(v), mean=float(np.mean(v)), n=len(v)) for k, v in by_stage.items()}
    record(name, dict(bound=RATIO_BOUND, worst=worst, summary=summary, rows=rows))
    bad = [r for r in rows if r["ratio_rms"] > RATIO_BOUND and r["err_native_rms"] > 1e-4 * r["scale_rms"]]
    assert not bad, f"{name}: {len(bad)} stage(s) further from the fp32 truth than {RATIO_BOUND} x the bf16 reference: {bad[:4]}"
    return summary


@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_every_layer_matches_hf_on_the_oracles_input(dev, preset):
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O, layer_probe as P
    cfg = get_config(preset)
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=5, init_scale=1.5)
    hf32 = O.build_hf_model(cfg, dtype=torch.float32, seed=5, init_scale=1.5)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=2048, max_new_rows=1024, max_patches=4096,
                                                          max_history=4)
    T, H, W = 6, 112, 168
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=9, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=9).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    a16 = P.probe(hf16, cfg, ids, pv, grid)
    t32 = P.probe(hf32, cfg, ids, pv, grid, P.inputs_of(a16))
    nat = native_probe(native, cfg, ids, frames.to(dev), a16)
    _compare(f"per_layer_parity[{preset}]", nat, a16, t32, cfg, ids)


@pytest.mark.skipif(os.environ.get("LCC_SKIP_SLOW") == "1", reason="LCC_SKIP_SLOW=1")
def test_every_layer_at_livecc_7b_shapes_matches_hf_on_the_oracles_input(dev, slow_budget):
    """BASELINE.json configs[1] shapes: the 6-frame first turn (4,368 patches through the 32-block 1280-dim tower, 1,131 rows through
    the 28 decoder layers of LiveCC-7B), seeded synthetic weights shared by construction (tiled:0).  About five minutes of host time
    (HF bf16 + two fp32 forwards at 7B).

    The same models also give the END-TO-END first-token check against the fp32 truth (the unconditional fp32 assert of VERDICT r2 #1a
    lives here since the GPU tier has a 20-minute limit; the 32-step version is bench.py's default `parity` object and, opt-in, the
    LCC_PARITY_FP32=1 leg of test_livecc_7b_turns_match_hf_cpu_path_on_identical_weights): native free-running logits of the turn's
    first generated token vs HF fp32 free-running, against HF bf16 free-running vs fp32 --
        rms over the vocabulary of (native - fp32) <= 1.25 x rms(bf16 - fp32);  worst logit <= 1.5 x the reference's + 1e-3 x scale."""
    slow_budget(330)
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import hf_oracle as O, layer_probe as P
    cfg = get_config("livecc-7b")
    T, H, W = 6, 392, 728
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=1234, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=1234).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    hf = O.build_hf_model_synthetic(cfg, torch.bfloat16, "tiled:0")
    a16 = P.probe(hf, cfg, ids, pv, grid)
    hf = hf.float()                                    # same bf16-representable weights, fp32 arithmetic = the truth
    t32 = P.probe(hf, cfg, ids, pv, grid, P.inputs_of(a16))
    f32_logits = P.probe(hf, cfg, ids, pv, grid, light=True)["logits"]      # fp32 FREE-running (no overrides): the end-to-end truth
    del hf
    arena = WeightArena(cfg, dev).fill_tiled(seed=0)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=2048, max_new_rows=1280, max_patches=4608, max_history=4)
    nat = native_probe(native, cfg, ids, frames.to(dev), a16)
    summary = _compare("per_layer_parity[livecc-7b]", nat, a16, t32, cfg, ids)
    print("per-layer parity at LiveCC-7B shapes:", summary)
    # end to end, free-running on both sides (a16 was recorded without overrides: it IS HF bf16's own forward)
    r = native.generate(input_ids=torch.from_numpy(np.asarray(ids)).view(1, -1), frames=frames.to(dev), frames_layout="TCHW", max_new_tokens=1,
                        output_logits=True, do_sample=False)
    n_log = r.logits[0].float().cpu().double().view(-1)
    r.past_key_values.release()
    t_log, b_log = f32_logits.double().view(-1), a16["logits"].double().view(-1)
    scale = float(t_log.abs().max())
    en, eb = (n_log - t_log), (b_log - t_log)
    rep = dict(scale=scale, rms_err_native=float(en.pow(2).mean().sqrt()), rms_err_ref16=float(eb.pow(2).mean().sqrt()),
               max_err_native=float(en.abs().max()), max_err_ref16=float(eb.abs().max()),
               argmax_native=int(n_log.argmax()), argmax_ref16=int(b_log.argmax()), argmax_fp32=int(t_log.argmax()))
    rep["rms_ratio"] = rep["rms_err_native"] / max(rep["rms_err_ref16"], 1e-30)
    rep["max_ratio"] = rep["max_err_native"] / max(rep["max_err_ref16"], 1e-30)
    record("livecc7b_first_token_vs_fp32_free_running", rep)
    print("first-token logits vs fp32 (free running):", rep)
    assert rep["rms_ratio"] <= 1.25, rep
    assert rep["max_err_native"] <= 1.5 * rep["max_err_ref16"] + 1e-3 * scale, rep


def test_qwen2vl_72b_shaped_layer_with_fp8_weights_matches_hf_on_the_dequantised_weights(dev):
    """BASELINE.json configs[4] at the REAL 72B widths (VERDICT r3 missing #3): ONE Qwen2-VL-72B decoder layer (hidden 8192, 64 / 8 heads,
    intermediate 29568: 0.86 B parameters) + the 152,064 x 8192 embedding and lm_head, LLM Linear weights as OCP e4m3 bytes + fp32 row
    scales -- the single-GPU 72B weight path -- behind one vision block.  Oracles: HF on the SAME quantised values (`fake_quantize_llm_fp8`):
    fp32 = exact q x scale (the truth), bf16 = what the reference's dtype makes of the dequantised checkpoint.  (i) layer by layer on the
    bf16 oracle's inputs: rms(native - fp32) <= 1.25 x rms(HF_bf16 - fp32) after the attention block, after the MLP and for final norm +
    lm_head (the fp8 prefill GEMMs: MFMA kernels on fp8 fragments converted after the LDS read); (ii) the 6-frame + 2-frame turns with 5
    tokens each, native and HF fp32 teacher-forced along HF bf16's own tokens (the fp8 decode GEMVs of pipeline v2): on every step the error
    against fp32 <= 1.5 x the bf16 oracle's (worst logit) and <= 1.25 x in rms."""
    import copy
    import dataclasses
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O, layer_probe as P
    cfg = dataclasses.replace(get_config("qwen2vl-72b"), num_hidden_layers=1, vit_depth=1, name="qwen2vl-72b-1layer")
    hf32 = O.build_hf_model_synthetic(cfg, torch.float32, "tiled:0")
    O.fake_quantize_llm_fp8(hf32)
    hf16 = copy.deepcopy(hf32).to(torch.bfloat16)
    native = LiveCCForConditionalGeneration.from_hf_model(hf32, cfg, dev, llm_fp8=True, max_streams=1, max_kv_len=2048, max_new_rows=1280,
                                                          max_patches=4608, max_history=8)
    assert native.weights.llm_fp8 and native.weights.view("llm.0.gate_up_w").dtype == torch.uint8
    T, H, W = 6, 392, 728
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=1234, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=1234).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    a16 = P.probe(hf16, cfg, ids, pv, grid)
    t32 = P.probe(hf32, cfg, ids, pv, grid, P.inputs_of(a16))
    nat = native_probe(native, cfg, ids, frames.to(dev), a16)
    summary = _compare("per_layer_parity[qwen2vl-72b-1layer-fp8]", nat, a16, t32, cfg, ids)
    print("72B-shaped fp8 layer:", summary)
    # (ii) the streaming replay: 6-frame turn + 2-frame turn over the carried cache, 5 tokens each (4 decode steps per turn: the fp8 decode
    # GEMVs of pipeline v2).  HF bf16 runs free; HF fp32 and the native engine follow its tokens.  At this synthetic one-layer configuration
    # the reference's OWN bf16 run is 10-14 % of the logit scale away from the fp32 truth in the second turn (bf16 cos / sin at positions
    # > 1,100 with nothing to average over: measured with HF alone, tools/diag_72b.py + profiles/r04/diag_72b_call4.jsonl) -- an absolute
    # bound against the bf16 oracle would test HF's noise, so the bound is the one of DESIGN section 5: the native error against the fp32
    # truth <= 1.5 x the bf16 reference's own (worst logit) and <= 1.25 x in rms over the vocabulary, on every step.
    frames2 = torch.from_numpy(protocol.synth_frames(8, H, W, seed=77, layout="TCHW"))
    res16 = O.replay_stream(hf16, cfg, frames2, protocol.TurnBuilder(cfg, seed=77), 5, 1.05, max_turns=2)
    s32 = O.OracleStream(hf32, cfg)
    state, past, worst = None, None, dict(max_ratio=0.0, rms_ratio=0.0, rel_err_ref16=0.0, rel_err_native=0.0)
    for ti, (r, (a, b)) in enumerate(zip(res16, protocol.split_clip(8))):
        pv2, g2 = O.patchify_normalize_ref(frames2[a:b], cfg)
        r32 = s32.turn(r["turn_ids"], pv2, g2, max_new_tokens=5, repetition_penalty=1.05, teacher_tokens=r["new_tokens"])
        full = np.asarray(r["turn_ids"]) if past is None else np.concatenate([past, np.asarray(r["turn_ids"])])
        rn = native.generate(input_ids=torch.from_numpy(full).view(1, -1), frames=frames2[a:b], past_key_values=state, repetition_penalty=1.05,
                             max_new_tokens=5, min_new_tokens=5, output_logits=True, do_sample=False, teacher_tokens=r["new_tokens"])
        state = rn.past_key_values
        past = np.concatenate([full, np.asarray(r["new_tokens"][:-1], dtype=np.int64)])
        for k in range(5):
            ln, l16, l32 = rn.logits[k].float().cpu().double(), r["logits"][k].double(), r32["logits"][k].double()
            scale = float(l32.abs().max())
            en, eo = float((ln - l32).abs().max()), float((l16 - l32).abs().max())
            rr = float((ln - l32).pow(2).mean().sqrt() / (l16 - l32).pow(2).mean().sqrt())
            worst = dict(max_ratio=max(worst["max_ratio"], en / (eo + 1e-3 * scale)), rms_ratio=max(worst["rms_ratio"], rr),
                         rel_err_ref16=max(worst["rel_err_ref16"], eo / scale), rel_err_native=max(worst["rel_err_native"], en / scale))
            assert en <= 1.5 * eo + (1e-3 + 2.0 ** -7) * scale, f"turn {ti} step {k}: native error vs fp32 {en:.4g} > 1.5 x the bf16 reference's {eo:.4g}"
            assert rr <= 1.25, f"turn {ti} step {k}: rms error ratio {rr:.3f}"
    state.release()
    record("stream_72b_1layer_fp8", worst)
    print("72B-shaped fp8 stream:", worst)


def test_qwen2vl_72b_fp8_full_depth_against_the_committed_hf_logits(dev):
    """BASELINE.json configs[4] through ALL 80 decoder layers (VERDICT r4 missing #4): Qwen2-VL-72B shapes, LLM Linear weights as OCP e4m3 +
    fp32 row scales (the 73-GB single-GPU arena), the 6-frame first turn of the benchmark protocol (4,368 patches, a 1,131-row prefill) + 3
    decode steps, teacher-forced along the fixture's tokens -- against tests/golden/qwen2vl72b_fp8_full_depth.npz: HF's own modules executed
    layer-streamed in the build container on the same seeded synthetic weights after the same per-row e4m3 quantisation (oracle/
    make_golden_72b.py: fp32 = exact q x scale = the truth, bf16 = what the reference's dtype makes of the dequantised checkpoint).
    After 80 layers the reference's OWN bf16 run is 4.4 % of the logit scale (rms 0.37-0.39, scale 7.8-9.4) away from the fp32 truth, so
    the gross-error guard is stated in units of that committed error.  Per step: |native - HF_bf16| at HF's top-64 ids <= max(6e-2 x scale,
    6 x rms(HF_bf16 - fp32)); rms over the 4,096 sample ids of (native - fp32) <= 1.25 x rms(HF_bf16 - fp32) -- the sensitive bound; the
    native argmax equals HF fp32's wherever the fp32 top-1 / top-2 margin exceeds 8 x that rms.
    Depth profile (tests/golden/qwen2vl72b_fp8_depth_profile.npz, the same generator with --profile): the residual stream of HF fp32 / HF bf16
    after every 4th layer at 4 prompt rows x 512 hidden dims against the engine's taps -- the same ratio bound at every probed depth.  (Round 5:
    the profile sat at 1.16-1.26 from layer 3 on until the arena's GPU-side e4m3 quantiser was made byte-identical to the oracle's CPU one --
    `weights.quantize_fp8_rows`; 0.15 % of the weights had been one e4m3 step apart -- and is 0.95-1.03 since; logits 0.95-1.02.)"""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_72b as G
    g = dict(np.load(G.PATH))
    seed_in, T, H, W, n_forced, seed_w, L = (int(x) for x in g["meta"])
    cfg = get_config("qwen2vl-72b")
    assert L == cfg.num_hidden_layers == 80 and (seed_in, T, H, W, n_forced) == (G.SEED_IN, G.T, G.H, G.W, G.N_FORCED)
    grid = protocol.grid_of(T, H, W, cfg)
    ids = np.asarray(protocol.TurnBuilder(cfg, seed=seed_in).turn_ids(0, protocol.num_video_tokens(grid, cfg)), dtype=np.int64)
    forced = G.teacher_tokens(cfg)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(forced, g["tokens"]) and np.array_equal(g["sample_ids"], G.sample_ids(cfg.vocab_size))
    arena = WeightArena(cfg, dev, llm_fp8=True).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=2048, max_new_rows=1280, max_patches=4608, max_history=8)
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=seed_in, layout="TCHW")).to(dev)
    # depth profile: the residual stream after every 4th layer at 4 prompt rows x 512 hidden dims (engine taps: copies only, same arithmetic)
    prof = dict(np.load(G.PROFILE_PATH))
    assert [int(x) for x in prof["meta"]] == [int(x) for x in g["meta"]] and np.array_equal(prof["dims"], G.profile_dims(cfg.hidden_size))
    taps = native.engine.set_llm_taps(len(ids))
    try:
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames, frames_layout="TCHW", repetition_penalty=1.0,
                            max_new_tokens=n_forced, min_new_tokens=n_forced, output_logits=True, do_sample=False, teacher_tokens=[int(t) for t in forced])
        torch.cuda.synchronize()
        rows_t, dims_t = torch.from_numpy(prof["rows"]).to(dev), torch.from_numpy(prof["dims"]).to(dev)
        tap_idx = torch.tensor([0 if l < 0 else 2 * int(l) + 2 for l in prof["layers"]], device=dev)
        nat_h = taps[tap_idx][:, rows_t][:, :, dims_t].float().cpu().numpy().astype(np.float64)      # [probe layers, rows, dims]
    finally:
        native.engine.set_llm_taps(0)
        del taps
    assert r.sequences[0, len(ids):].tolist() == [int(t) for t in forced]
    lg = r.logits.float().cpu().numpy()
    r.past_key_values.release()
    t32h, b16h = prof["h_fp32"].astype(np.float64), prof["h_bf16"].astype(np.float64)
    e_nat = np.sqrt(((nat_h - t32h) ** 2).mean(axis=(1, 2)))
    e_ref = np.sqrt(((b16h - t32h) ** 2).mean(axis=(1, 2)))
    depth = dict(layers=[int(l) for l in prof["layers"]], rms_ratio=[round(float(a / max(b, 1e-30)), 4) for a, b in zip(e_nat, e_ref)],
                 ref_bf16_rms_err=[round(float(b), 5) for b in e_ref], native_rms_err=[round(float(a), 5) for a in e_nat],
                 rms_native_minus_ref_bf16=[round(float(x), 5) for x in np.sqrt(((nat_h - b16h) ** 2).mean(axis=(1, 2)))])
    print("72B fp8 depth profile (residual stream, rms(native - fp32) / rms(HF_bf16 - fp32)):", depth)
    sid = g["sample_ids"]
    rec = dict(layers=L, prompt_rows=len(ids), steps=n_forced, depth_profile=depth, rel_dlogit_top=[], rms_ratio=[], decided=0, decided_equal=0, argmax_equal_fp32=0,
               ref_bf16_rms_err_over_scale=[float(a / b) for a, b in zip(g["t0_rms_err_bf16_full_vocab"], g["t0_scale"])])
    for k in range(n_forced):
        scale = float(g["t0_scale"][k])
        d = float(np.abs(lg[k][g["t0_top_ids"][k]] - g["t0_top_vals_bf16"][k].astype(np.float64)).max()) / max(scale, 100.0 * float(g["t0_rms_err_bf16_full_vocab"][k]))
        n, b16, t32 = lg[k][sid].astype(np.float64), g["t0_sample_vals_bf16"][k].astype(np.float64), g["t0_sample_vals_fp32"][k].astype(np.float64)
        ratio = float(np.sqrt(((n - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean()))
        rec["rel_dlogit_top"].append(d)
        rec["rms_ratio"].append(ratio)
        top2 = g["t0_fp32_top2_vals"][k]
        own, want = int(lg[k].argmax()), int(g["t0_fp32_top2_ids"][k][0])
        rec["argmax_equal_fp32"] += int(own == want)
        if float(top2[0] - top2[1]) > 8.0 * float(g["t0_rms_err_bf16_full_vocab"][k]):
            rec["decided"] += 1
            rec["decided_equal"] += int(own == want)
    record("qwen2vl72b_fp8_full_depth_vs_committed_golden", rec)
    print("72B fp8, 80 layers, vs the committed HF logits:", rec)
    assert max(rec["rel_dlogit_top"]) <= 6e-2, rec
    assert max(rec["rms_ratio"]) <= 1.25, rec
    assert max(depth["rms_ratio"]) <= 1.25, depth
    assert rec["decided_equal"] == rec["decided"], rec
    # ---- decisive-weight leg (round 6, VERDICT r5 weak #1: the fixture above has 0 decided tokens -- flat synthetic logits).  The SAME arena is
    # refilled in place with the `decisive` variant (embedding-aligned lm_head, livecc_amd/weights.py) and the same call is repeated against
    # tests/golden/qwen2vl72b_fp8_decisive.npz (oracle/make_golden_72b.py --variant decisive: the same layer-streamed HF execution): now every
    # step's greedy token is decided by the model through 80 fp8 layers, and the native choice must be HF's on every step.
    gd = dict(np.load(G.PATH_DECISIVE))
    assert str(gd["variant"]) == "decisive" and [int(x) for x in gd["meta"]] == [int(x) for x in g["meta"]] and np.array_equal(gd["ids"], ids)
    arena.fill_tiled(seed=seed_w, variant="decisive")
    r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames, frames_layout="TCHW", repetition_penalty=1.0,
                        max_new_tokens=n_forced, min_new_tokens=n_forced, output_logits=True, do_sample=False, teacher_tokens=[int(t) for t in forced])
    lgd = r.logits.float().cpu().numpy()
    r.past_key_values.release()
    recd = dict(steps=n_forced, decided=0, decided_equal=0, argmax_equal_fp32=0, argmax_equal_bf16=0, rms_ratio=[], margin_over_rms_err=[])
    for k in range(n_forced):
        n, b16, t32 = lgd[k][sid].astype(np.float64), gd["t0_sample_vals_bf16"][k].astype(np.float64), gd["t0_sample_vals_fp32"][k].astype(np.float64)
        recd["rms_ratio"].append(float(np.sqrt(((n - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean())))
        top2 = gd["t0_fp32_top2_vals"][k]
        own, want = int(lgd[k].argmax()), int(gd["t0_fp32_top2_ids"][k][0])
        recd["argmax_equal_fp32"] += int(own == want)
        recd["argmax_equal_bf16"] += int(own == int(gd["hf_bf16_argmax"][k]))
        mor = float(top2[0] - top2[1]) / float(gd["t0_rms_err_bf16_full_vocab"][k])
        recd["margin_over_rms_err"].append(round(mor, 1))
        # Through 80 fp8 layers the reference's own bf16 error (rms 0.33-0.37 of a logit scale of 8.6-9.6) eats most of the margin the decisive
        # head has at 7B (13-60 x noise): here HF fp32's top-1 / top-2 margin is 1.6-5.1 x rms(HF_bf16 - fp32) on the four steps -- yet HF bf16,
        # HF fp32 and the id the head's permutation predicts for the current token agree on all four.  A step counts as decided at > 3 x rms
        # (~2 sigma of the difference of two logits: 3 of the 4 steps); the fourth (1.6 x) is reported, not asserted.
        if mor > 3.0:
            recd["decided"] += 1
            recd["decided_equal"] += int(own == want)
    assert np.array_equal(gd["hf_bf16_argmax"], gd["hf_fp32_argmax"]), "the fixture's two HF runs agree on every step's token"
    record("qwen2vl72b_fp8_decisive_vs_committed_golden", recd)
    print("72B fp8, 80 layers, decisive weights:", recd)
    assert recd["decided"] >= 3 and recd["decided_equal"] == recd["decided"], recd
    assert max(recd["rms_ratio"]) <= 1.25, recd
