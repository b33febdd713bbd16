"""-m gpu: the HIP path against the COMMITTED golden fixture (tests/golden/stream_tiny.npz, produced by oracle/make_golden.py
from the HF CPU oracle in the build container).  Same weights (seeded HF init), frames and ids; the fixture's tokens are the
bf16 oracle's free-running greedy tokens, so this also checks free-running token agreement, not only teacher-forced logits."""
import os

import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu

# ---- bounds of the committed-fixture comparisons (round 6, VERDICT r5 weak #1: "bounds to measured + 10 %") --------------------------------
# measured on the final trees of rounds 5 and 6 (profiles/r0*/parity_report.json; the kernels are deterministic: the same tree gives the same
# numbers on every box): per-step rms ratio 1.061 (two turns) / 1.033 (one-shot) / 1.144 (8 batched streams) / 1.110 (238-turn stream);
# all-steps ratio 0.996-1.002; |native - HF_bf16| at HF's top-64 ids 0.0198-0.0355 of the logit scale (0.0451 over the FULL vocabulary of the
# first token); tokens equal 30 / 32, 32 / 32, 56 / 64, 71 / 72, 8 / 8.
RMS_STEP_MAX = 1.15           # per-step rms(native - fp32) / rms(HF_bf16 - fp32); 1.20 where 64-72 steps are compared (batch8, long stream)
RMS_STEP_MAX_MANY = 1.20
RMS_ALL_MAX = 1.05            # over all steps of a fixture
GROSS_MAX = 4.5e-2            # |native - HF_bf16| at HF's top-64 ids, of the logit scale
TOKENS_SLACK = 3              # tokens_equal >= steps - 3 on the single-stream fixtures (a regression from 30 / 32 to 22 / 32 used to pass)


def _undecided_slack(st):
    """Token mismatches tolerated on a LONG fixture: every step HF's own margin decides must match (asserted separately: decided_equal ==
    decided); of the undecided steps -- flat synthetic logits, the winner changes with the rounding of any split merge -- at most a tenth may
    differ, never fewer than TOKENS_SLACK.  (Round 6: the 72-step stream went from 69-70 to 68 equal tokens when the chunk attention moved from
    4 to 5 key splits -- rms(native - fp32) / rms(HF_bf16 - fp32) stayed 1.003 over all steps -- so a fixed 3 of 72 measured the split count.)"""
    return max(TOKENS_SLACK, -(-(st["steps"] - st["decided"]) // 10))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_tiny.npz")


def test_native_stream_against_golden_fixture(dev):
    from livecc_amd import protocol
    from livecc_amd.config import tiny
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O
    g = np.load(GOLDEN)
    cfg = tiny()
    seed_w, seed_in, frames_n, H, W, max_new = (int(x) for x in g["meta"])
    hf16 = O.build_hf_model(cfg, torch.bfloat16, seed_w, 2.0)       # weights only; the oracle is not run here
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=2048, max_new_rows=1024,
                                                          max_patches=4096, max_history=16)
    frames = torch.from_numpy(protocol.synth_frames(frames_n, H, W, seed=seed_in, layout="TCHW"))
    state, past = None, None
    agree = total = 0
    worst = 0.0
    for ti, (a, b) in enumerate(protocol.split_clip(frames_n)[:int(g["n_turns"])]):
        new = g[f"t{ti}_ids"]
        ids = new if past is None else np.concatenate([past, new])
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], past_key_values=state,
                            repetition_penalty=1.05, max_new_tokens=max_new, min_new_tokens=max_new, output_logits=True)
        state = r.past_key_values
        toks = r.sequences[0, len(ids):].tolist()
        gold = g[f"t{ti}_tokens"].tolist()
        # ViT features of this chunk vs the fixture
        vit = native.get_video_features(frames=frames[a:b].to(dev)).float().cpu().numpy()
        v32, v16 = g[f"t{ti}_vit_fp32"], g[f"t{ti}_vit_bf16"]
        assert np.abs(vit - v32).max() <= 1.5 * np.abs(v16 - v32).max() + 1e-3 * np.abs(v32).max()
        # logits are comparable step by step as long as the histories agree
        lg = r.logits.float().cpu().numpy()
        l16, l32 = g[f"t{ti}_logits_bf16"], g[f"t{ti}_logits_fp32"]
        for k in range(max_new):
            total += 1
            if toks[:k] != gold[:k]:
                break
            scale = np.abs(l32[k]).max()
            d = np.abs(lg[k] - l16[k]).max()
            worst = max(worst, d / scale)
            assert d <= 6e-2 * scale, f"turn {ti} step {k}: {d} vs scale {scale}"
            assert np.abs(lg[k] - l32[k]).max() <= 1.5 * np.abs(l16[k] - l32[k]).max() + (1e-3 + 2.0 ** -7) * scale
            agree += int(toks[k] == gold[k])
        # continue along the GOLDEN history so that later turns stay comparable
        past = np.concatenate([ids, np.asarray(gold[:-1], dtype=np.int64)])
        if toks != gold:
            state.release()
            state = None
            break
    record("golden_stream_tiny", dict(agree=agree, total=total, worst_rel_dlogit=worst))
    assert agree >= 0.8 * total, f"only {agree}/{total} free-running greedy tokens equal the golden oracle tokens"


STAGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages_tiny.npz")


def test_native_stages_against_the_per_stage_golden_fixture(dev):
    """Every stage of the native engine (PatchEmbed, each vision block, merger, each decoder layer after attention / after the MLP,
    final norm + lm_head) on the COMMITTED bf16-oracle input of that stage against the committed fp32 truth of that stage
    (tests/golden/stages_tiny.npz from oracle/make_golden.py): rms error <= 1.25 x the bf16 oracle's own."""
    from livecc_amd import protocol
    from livecc_amd.config import tiny
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O
    from oracle.make_golden import load_stages
    from tests.test_gpu_layer_parity import _compare, native_probe
    a16, t32, g = load_stages(STAGES)
    cfg = tiny()
    seed_w, seed_in, n_frames, H, W = (int(x) for x in g["meta"])
    hf16 = O.build_hf_model(cfg, torch.bfloat16, seed_w, 2.0)       # weights only; the oracle is not run here
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=2048, max_new_rows=1024,
                                                          max_patches=4096, max_history=4)
    frames = torch.from_numpy(protocol.synth_frames(10, H, W, seed=seed_in, layout="TCHW"))[:n_frames]
    nat = native_probe(native, cfg, g["ids"], frames.to(dev), a16)
    _compare("per_stage_golden_tiny", nat, a16, t32, cfg, g["ids"])


def test_livecc7b_first_token_against_the_committed_hf_logits(dev):
    """REAL LiveCC-7B shapes against a COMMITTED golden vector (tests/golden/livecc7b_first_token.npz, oracle/make_golden_7b.py: the
    executed HF reference -- bf16 run and fp32 truth -- on the seeded `tiled:0` weights, 6-frame first turn of BASELINE configs[1],
    4,368 patches + a 1,131-token prefill, free-running).  No HF forward on the GPU box: the native first-token logits must be as
    close to the committed fp32 truth as the committed bf16 reference run is (rms over the 152,064 logits <= 1.25 x, worst logit
    <= 1.5 x + 1e-3 x scale), within 6e-2 x scale of the bf16 run, and pick the fp32 argmax (its margin, 0.49, is 2.5 x the worst error)."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b as G
    fx = G.load()
    seed_in, T, H, W, seed_w = (int(x) for x in fx["meta"])
    cfg = get_config("livecc-7b")
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=seed_in, layout="TCHW"))
    ids = protocol.TurnBuilder(cfg, seed=seed_in).turn_ids(0, protocol.num_video_tokens(tuple(int(x) for x in fx["grid"]), cfg))
    assert np.array_equal(np.asarray(ids), fx["ids"]), "the prompt of the fixture is rebuilt from its seeds"
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=2048, max_new_rows=1280, max_patches=4608, max_history=4)
    r = native.generate(input_ids=torch.from_numpy(np.asarray(ids)).view(1, -1), frames=frames.to(dev), frames_layout="TCHW", max_new_tokens=1,
                        output_logits=True, do_sample=False)
    n_log = r.logits[0].float().cpu().numpy().reshape(-1)
    r.past_key_values.release()
    rep = G.compare(n_log, fx)
    record("livecc7b_first_token_vs_committed_golden", rep)
    print("LiveCC-7B first token vs the committed HF logits:", rep)
    assert rep["rms_ratio"] <= 1.10, rep                                                 # measured 1.0045
    assert rep["max_err_native"] <= 1.5 * rep["max_err_ref16"] + 1e-3 * rep["scale"], rep
    assert rep["max_abs_native_vs_ref16"] <= 5.5e-2 * rep["scale"], rep                 # over all 152,064 logits (not the top-64 ids): measured 0.0451
    assert rep["argmax_native"] == rep["argmax_fp32"] == rep["argmax_ref16"], rep


def test_livecc7b_greedy_tokens_equal_the_committed_hf_tokens_on_decisive_weights(dev):
    """north_star: "token-id exact under greedy", at the REAL LiveCC-7B shapes against COMMITTED reference output
    (tests/golden/livecc7b_decisive_stream.npz, oracle/make_golden_7b.py --decisive: HF bf16, free-running, the benchmark protocol's
    6-frame + 2-frame turns, 16 greedy tokens each, repetition_penalty 1.05, on the `decisive:0` weights whose top-1 margin is 31 % of the
    logit scale at every step).  The HIP path, free-running on the same seeds, must emit the SAME 32 token ids; its raw logits at HF's
    top-8 ids stay within 6e-2 x scale of HF's.  No HF forward on the GPU box (the 97-s live version is
    test_greedy_tokens_are_exact_on_decisive_weights).  A token-identity test on margins this wide is NOT a sensitivity test of the
    layers (a several-x mid-network error would pass): that is what the rms-ratio fixtures below are for."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "livecc7b_decisive_stream.npz")))
    seed_in, n_frames, H, W, max_new, seed_w = (int(x) for x in g["meta"])
    cfg = get_config("livecc-7b")
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w, variant="decisive")
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=4096, max_new_rows=1280, max_patches=4608, max_history=16)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed_in, layout="TCHW"))
    builder = protocol.TurnBuilder(cfg, seed=seed_in)
    state, past, equal, total, worst = None, None, 0, 0, 0.0
    for ti, (a, b) in enumerate(protocol.split_clip(n_frames)[:int(g["n_turns"])]):
        new = g[f"t{ti}_ids"]
        assert np.array_equal(new, np.asarray(builder.turn_ids(ti, protocol.num_video_tokens(tuple(int(x) for x in g[f"t{ti}_grid"]), cfg))))
        ids = new if past is None else np.concatenate([past, new])
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], past_key_values=state, repetition_penalty=1.05,
                            max_new_tokens=max_new, min_new_tokens=max_new, output_logits=True, do_sample=False)
        state = r.past_key_values
        toks = r.sequences[0, len(ids):].tolist()
        gold = g[f"t{ti}_tokens"].tolist()
        lg = r.logits.float().cpu().numpy()
        for k in range(max_new):
            total += 1
            equal += int(toks[k] == gold[k])
            if toks[:k] == gold[:k]:           # same history: the raw logits are comparable
                d = np.abs(lg[k][g[f"t{ti}_top_ids"][k]] - g[f"t{ti}_top_vals"][k]).max() / float(g[f"t{ti}_scale"][k])
                worst = max(worst, float(d))
        past = np.concatenate([ids, np.asarray(gold[:-1], dtype=np.int64)])
        if toks != gold:
            break
    if state is not None:
        state.release()
    record("livecc7b_decisive_tokens_vs_committed_golden", dict(tokens_equal=equal, tokens_total=total, worst_rel_dlogit_at_top8=worst))
    assert (equal, total) == (32, 32), f"{equal}/{total} greedy tokens equal HF's committed tokens"
    assert worst <= 6e-2, worst


# ---------------------------------------------------------------------------------------------------------------------
# Fixture-based comparisons at the real shapes, TEACHER-FORCED along the committed reference stream (round 4): the first GPU run of the
# round-3 free-running versions showed what free running costs on random weights -- the native tokens left HF's at the first step whose
# top-1 margin lies inside bf16 noise (step 3 of 32) and nothing after it was comparable.  The engine now forces the committed tokens
# (lcc_debug_set_forced_tokens = what the HF oracle does with a forcing LogitsProcessor), so EVERY step is compared, and the token the
# native path would have chosen is recomputed on the host from its raw logits (repetition penalty over the ids seen so far, argmax).
# ---------------------------------------------------------------------------------------------------------------------
def own_choice(raw_logits: np.ndarray, seen_ids, penalty: float) -> int:
    """HF RepetitionPenaltyLogitsProcessor + argmax on one step's raw logits: score < 0 -> * penalty, else / penalty, for every id of
    the history (prompt + generated so far: ref demo/infer.py:160,169)."""
    sc = raw_logits.astype(np.float32).copy()
    idx = np.unique(np.asarray(list(seen_ids), dtype=np.int64))
    v = sc[idx]
    sc[idx] = np.where(v < 0, v * np.float32(penalty), v / np.float32(penalty))
    return int(sc.argmax())


def processed_top2(top_ids, top_vals, seen_ids, penalty):
    """The reference's own decision on one step, rebuilt from its top-K RAW logits: repetition penalty on the ids of the history, then the
    winner and its margin over the runner-up (both are among the raw top-K: the penalty moves a logit by < 5 % of its value)."""
    v = np.asarray(top_vals, dtype=np.float64).copy()
    hit = np.isin(np.asarray(top_ids), np.asarray(list(seen_ids), dtype=np.int64))
    v[hit] = np.where(v[hit] < 0, v[hit] * penalty, v[hit] / penalty)
    order = np.argsort(-v, kind="stable")
    return int(np.asarray(top_ids)[order[0]]), float(v[order[0]] - v[order[1]])


def follow_golden_stream(native, turns, frames_of_turn, max_new, top_key, noise_of_step, penalty=1.05, sample=None, past_rule=True):
    """The native path along a committed reference stream.  `turns`: per turn a dict(ids=new prompt ids, tokens=the reference's greedy
    tokens, top_ids / top_vals (its top-K raw logits per step), scale).  Every turn is teacher-forced along the reference's tokens; per
    step: |native - reference| at the reference's top-K ids relative to the scale, whether the native path's OWN choice (host-side
    penalty + argmax over its raw logits) equals the reference's token, and whether the reference's margin ON ITS PROCESSED SCORES (the
    penalty can cost the raw winner 5 % of its logit) exceeds the COMMITTED decision threshold `noise_of_step(turn, k)` (derived
    from the reference's own bf16-vs-fp32 noise, not from the native error) -- there the tokens must agree.
    `sample` = (ids, key16, key32): rms error ratio against the fp32 truth over fixed sample ids."""
    state, past = None, None
    st = dict(steps=0, tokens_equal=0, decided=0, decided_equal=0, worst_rel_dlogit_top=0.0, ratios=[], undecided_mismatch_steps=[])
    for ti, t in enumerate(turns):
        new = np.asarray(t["ids"], dtype=np.int64)
        ids = new if past is None else np.concatenate([past, new])
        gold = [int(x) for x in t["tokens"]]
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames_of_turn(ti), past_key_values=state,
                            repetition_penalty=penalty, max_new_tokens=max_new, min_new_tokens=max_new, output_logits=True, do_sample=False,
                            teacher_tokens=gold)
        state = r.past_key_values
        assert r.sequences[0, len(ids):].tolist() == gold, "teacher forcing: the history holds the committed tokens"
        lg = r.logits.float().cpu().numpy()
        seen = set(int(x) for x in ids)
        for k in range(max_new):
            st["steps"] += 1
            scale = float(t["scale"][k])
            top_ids, top_vals = np.asarray(t["top_ids"][k]), np.asarray(t[top_key][k], dtype=np.float64)
            d = float(np.abs(lg[k][top_ids] - top_vals).max())
            st["worst_rel_dlogit_top"] = max(st["worst_rel_dlogit_top"], d / scale)
            own = own_choice(lg[k], seen, penalty)
            st["tokens_equal"] += int(own == gold[k])
            winner, margin = processed_top2(top_ids, top_vals, seen, penalty)
            decided = margin > noise_of_step(ti, k) and winner == gold[k]
            if decided:
                st["decided"] += 1
                st["decided_equal"] += int(own == gold[k])
            elif own != gold[k]:
                st["undecided_mismatch_steps"].append((ti, k))
            if sample is not None:
                sid, k16, k32 = sample
                n, b16, t32 = lg[k][sid].astype(np.float64), t[k16][k].astype(np.float64), t[k32][k].astype(np.float64)
                st["ratios"].append(float(np.sqrt(((n - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean())))
            seen.add(gold[k])
        past = np.concatenate([ids, np.asarray(gold[:-1], dtype=np.int64)]) if past_rule else None
    if state is not None:
        state.release()
    return st


def _turns_of(g, n_turns, top_key):
    return [dict(ids=g[f"t{ti}_ids"], tokens=g[f"t{ti}_tokens"], top_ids=g[f"t{ti}_top_ids"], scale=g[f"t{ti}_scale"],
                 **{k[len(f"t{ti}_"):]: g[k] for k in g if k.startswith(f"t{ti}_") and ("vals" in k)}) for ti in range(n_turns)]


def test_livecc7b_two_turns_against_the_committed_hf_logits(dev):
    """BASELINE configs[1]'s first two turns at LiveCC-7B shapes (tiled:0 weights) against tests/golden/livecc7b_two_turns.npz (HF bf16
    free-running + HF fp32 teacher-forced, oracle/make_golden_7b.py --turns), teacher-forced along HF's tokens, all 32 steps: per step
    |native - HF_bf16| at HF's top-64 ids <= 6e-2 x scale; rms over the 4,096 sample ids of (native - fp32) <= 1.25 x rms(HF_bf16 - fp32)
    per step and <= 1.08 over all steps (their rms tracks the full vocabulary within 5 %); the native path's own choice equals HF's token
    wherever HF's margin on its processed scores exceeds 8 x the committed rms of (HF_bf16 - HF_fp32) of that step (4 sigma of the difference
    two equally accurate bf16 implementations show on a pair of logits; 11-12 of the 32 steps)."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b as G
    g = dict(np.load(G.PATH_TURNS))
    seed_in, n_frames, H, W, max_new, seed_w = (int(x) for x in g["meta"])
    cfg = get_config("livecc-7b")
    n_turns = int(g["n_turns"])
    builder = protocol.TurnBuilder(cfg, seed=seed_in)                                   # the fixture's prompts are rebuilt from the seed
    for ti in range(n_turns):
        assert np.array_equal(g[f"t{ti}_ids"], builder.turn_ids(ti, protocol.num_video_tokens(tuple(int(x) for x in g[f"t{ti}_grid"]), cfg)))
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=4096, max_new_rows=1280, max_patches=4608, max_history=16)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed_in, layout="TCHW"))
    chunks = protocol.split_clip(n_frames)
    turns = _turns_of(g, n_turns, "top_vals_bf16")

    def noise(ti, k):
        # decision threshold from the COMMITTED noise of the reference itself: rms over the vocabulary of (HF_bf16 - HF_fp32) on this step.  Two
        # equally accurate bf16 implementations differ per logit by ~sqrt(2) rms, the difference of two logits by ~2 rms: 8 rms = 4 sigma
        return 8.0 * float(g[f"t{ti}_rms_err_bf16_full_vocab"][k])
    st = follow_golden_stream(native, turns, lambda ti: frames[chunks[ti][0]:chunks[ti][1]], max_new, "top_vals_bf16", noise,
                              sample=(g["sample_ids"], "sample_vals_bf16", "sample_vals_fp32"))
    ratios = np.asarray(st.pop("ratios"))
    record("livecc7b_two_turns_vs_committed_golden", st | dict(worst_rms_ratio=float(ratios.max()), mean_rms_ratio=float(ratios.mean()),
                                                                rms_ratio_all_steps=float(np.sqrt((ratios ** 2).mean()))))
    assert st["steps"] == n_turns * max_new
    assert st["worst_rel_dlogit_top"] <= GROSS_MAX, st
    assert ratios.max() <= RMS_STEP_MAX and np.sqrt((ratios ** 2).mean()) <= RMS_ALL_MAX, (ratios.max(), ratios.mean())
    assert st["decided_equal"] == st["decided"], st
    assert st["tokens_equal"] >= st["steps"] - TOKENS_SLACK, st          # measured 30 of 32 (two undecided steps inside bf16 noise)


def test_qwen2vl2b_config0_against_the_committed_hf_stream(dev):
    """BASELINE configs[0] (Qwen2-VL-2B real shapes, 8-frame clip = 6 + 2 frames, greedy, 16 tokens per turn) against
    tests/golden/qwen2vl2b_config0_stream.npz (HF bf16 free-running, oracle/make_golden_2b.py) without HF's forward on the GPU box (HF
    builds the seeded weights only), teacher-forced along HF's tokens: raw logits at HF's top-64 ids within 6e-2 x scale on all 32 steps;
    the native path's own choice equals HF's wherever HF's margin on its processed scores exceeds 6 % of the logit scale (twice the bf16
    noise of these shapes: 2.9 % measured live against HF)."""
    from livecc_amd import protocol
    from livecc_amd.config import qwen2vl_2b
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O, make_golden_2b as G
    g = dict(np.load(G.PATH))
    seed_w, seed_in, n_frames, H, W, max_new = (int(x) for x in g["meta"])
    cfg = qwen2vl_2b()
    n_turns = int(g["n_turns"])
    builder = protocol.TurnBuilder(cfg, seed=seed_in)
    for ti in range(n_turns):
        assert np.array_equal(g[f"t{ti}_ids"], builder.turn_ids(ti, protocol.num_video_tokens(tuple(int(x) for x in g[f"t{ti}_grid"]), cfg)))
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=seed_w, init_scale=1.0)      # weights only
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=4096, max_new_rows=2048, max_patches=8192,
                                                          max_history=64)
    del hf16
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed_in, layout="TCHW"))
    chunks = protocol.split_clip(n_frames)
    turns = _turns_of(g, n_turns, "top_vals")
    st = follow_golden_stream(native, turns, lambda ti: frames[chunks[ti][0]:chunks[ti][1]], max_new, "top_vals",
                              lambda ti, k: 0.06 * float(g[f"t{ti}_scale"][k]))
    st.pop("ratios")
    record("qwen2vl2b_config0_vs_committed_golden", st)
    assert st["steps"] == n_turns * max_new
    assert st["worst_rel_dlogit_top"] <= GROSS_MAX, st
    assert st["decided_equal"] == st["decided"], st
    assert st["tokens_equal"] >= st["steps"] - TOKENS_SLACK, st


def test_livecc7b_oneshot480_against_the_committed_hf_logits(dev):
    """BASELINE configs[3] at the REAL shapes (VERDICT r3 missing #4): the reference's video_qa / MCQ first turn -- 480 frames 280x280 =
    96,000 patches in 240 temporal slices -> 24,000 visual tokens + a 24-id query in ONE generate call (a 24,058-row prefill served in
    4,096-row pieces over the carried KV), then 8 greedy tokens over the ~24k-key cache -- against tests/golden/livecc7b_oneshot480.npz
    (HF bf16 free-running + HF fp32 teacher-forced on the tiled:0 weights, oracle/make_golden_7b_long.py; HF 5.15 text-offset rule on both
    sides), teacher-forced along HF's tokens: |native - HF_bf16| at HF's top-64 ids <= 6e-2 x scale on every step (step 0 = the 24k-row
    prefill's token, steps 1-7 = decode steps at L ~ 24k), rms(native - fp32) <= 1.25 x rms(HF_bf16 - fp32) over the 4,096 sample ids, the
    native path's own choice equal to HF's wherever HF's processed-score margin exceeds 8 x its committed rms error."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b_long as L
    g = dict(np.load(L.PATH))
    seed_in, T, H, W, n_new, seed_w, qlen = (int(x) for x in g["meta"])
    assert (seed_in, T, H, W, n_new, qlen) == (L.SEED_IN, L.T, L.H, L.W, L.N_NEW, L.QUERY_LEN)
    cfg = get_config("livecc-7b")
    ids, grid = L.prompt_ids(cfg)
    assert len(ids) == int(g["ids_len"]) and np.array_equal(ids[:64], g["ids_head"]) and np.array_equal(ids[-64:], g["ids_tail"])
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=32 * ((len(ids) + n_new + 31) // 32) + 64, max_new_rows=4096,
                                            max_patches=40 * 400, max_history=16, text_offset_rule="hf5")
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=seed_in, layout="TCHW"))
    has32 = "t0_sample_vals_fp32" in g
    scale = g["t0_scale"] if has32 else g["t0_scale_bf16"]
    turn = dict(ids=ids, tokens=g["tokens"], top_ids=g["t0_top_ids"], scale=scale, top_vals_bf16=g["t0_top_vals_bf16"],
                sample_vals_bf16=g["t0_sample_vals_bf16"])
    if has32:
        turn["sample_vals_fp32"] = g["t0_sample_vals_fp32"]

    def noise(ti, k):          # decision threshold: 8 x the committed rms(HF_bf16 - HF_fp32) of the step (see the two-turn test)
        if has32:
            return 8.0 * float(g["t0_rms_err_bf16_full_vocab"][k])
        return 0.08 * float(scale[k])
    st = follow_golden_stream(native, [turn], lambda ti: frames, n_new, "top_vals_bf16", noise,
                              sample=(g["sample_ids"], "sample_vals_bf16", "sample_vals_fp32") if has32 else None, past_rule=False)
    ratios = np.asarray(st.pop("ratios")) if has32 else None
    rec = dict(st, prompt_rows=len(ids), patches=int(np.prod(grid)), fp32_truth=has32)
    if has32:
        rec.update(worst_rms_ratio=float(ratios.max()), mean_rms_ratio=float(ratios.mean()))
    record("livecc7b_oneshot480_vs_committed_golden", rec)
    assert st["steps"] == n_new
    assert st["worst_rel_dlogit_top"] <= GROSS_MAX, st
    if has32:
        assert ratios.max() <= RMS_STEP_MAX, ratios
    assert st["decided_equal"] == st["decided"], st


def test_livecc7b_batch_of_8_streams_against_the_committed_hf_logits(dev):
    """ONE GPU's share of BASELINE configs[2] at the REAL shapes (batched parity was at `small` / `tiny` only): 8 streams
    (the benchmark's seeds), the 6-frame turn + a 2-frame turn, ONE `generate_batch` call per turn -- batched vision tower over 34,944 /
    11,648 patches, packed prefill of 9,048 / 3,088 rows (the 256- and 192-row GEMM tiles, `attn_gqa32_kernel`), one weight pass per decode
    step for the 8 streams (`gemv_skinny_kernel` + `attn_decode_fused_kernel`) -- against tests/golden/livecc7b_batch8.npz: HF run once per
    stream (the reference has no batched path), bf16 free-running + fp32 teacher-forced (oracle/make_golden_7b_batch8.py).  Every stream
    follows its own HF tokens; per stream and step: |native - HF_bf16| at HF's top-64 ids <= 6e-2 x scale, rms(native - fp32) <= 1.25 x
    rms(HF_bf16 - fp32) over the 1,024 sample ids, the native path's own choice equal to HF's wherever HF's processed-score margin exceeds
    8 x its committed rms error."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b_batch8 as B
    g = dict(np.load(B.PATH))
    seed0, n_streams, n_frames, H, W, n_new, seed_w = (int(x) for x in g["meta"])
    cfg = get_config("livecc-7b")
    sid = g["sample_ids"]
    assert np.array_equal(sid, B.sample_ids(cfg.vocab_size))
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    n_tok = (H // 28) * (W // 28)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=n_streams, max_kv_len=4096, max_new_rows=n_streams * (3 * n_tok + 128),
                                            max_patches=n_streams * 12 * n_tok + 64, max_history=16)
    frames = [torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed0 + s, layout="TCHW")).to(dev) for s in range(n_streams)]
    builders = [protocol.TurnBuilder(cfg, seed=seed0 + s) for s in range(n_streams)]
    states, past = [None] * n_streams, [None] * n_streams
    st = dict(steps=0, tokens_equal=0, decided=0, decided_equal=0, worst_rel_dlogit_top=0.0)
    ratios = []
    for ti, (a, b) in enumerate(protocol.split_clip(n_frames)[:2]):
        reqs, rows, fulls = [], [], []
        for s in range(n_streams):
            new = g[f"s{s}_t{ti}_ids"]
            assert np.array_equal(new, builders[s].turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(b - a, H, W, cfg), cfg)))
            full = new if past[s] is None else np.concatenate([past[s], new])
            fulls.append(full)
            rows.append([int(x) for x in g[f"s{s}_t{ti}_tokens"]])
            reqs.append(dict(input_ids=torch.from_numpy(full), frames=frames[s][a:b], frames_layout="TCHW", state=states[s]))
        outs = native.generate_batch(reqs, repetition_penalty=B.PENALTY, max_new_tokens=n_new, force_length=True, output_logits=True,
                                     teacher_tokens=rows)
        for s, o in enumerate(outs):
            states[s] = o.past_key_values
            assert o.sequences[0, len(fulls[s]):].tolist() == rows[s]
            lg = o.logits.float().cpu().numpy()
            seen = set(int(x) for x in fulls[s])
            k0 = f"s{s}_t{ti}"
            for k in range(n_new):
                scale = float(g[f"{k0}_scale"][k])
                top_ids, top_vals = g[f"{k0}_top_ids"][k].astype(np.int64), g[f"{k0}_top_vals_bf16"][k].astype(np.float64)
                d = float(np.abs(lg[k][top_ids] - top_vals).max())
                st["steps"] += 1
                st["worst_rel_dlogit_top"] = max(st["worst_rel_dlogit_top"], d / scale)
                own = own_choice(lg[k], seen, B.PENALTY)
                st["tokens_equal"] += int(own == rows[s][k])
                winner, margin = processed_top2(top_ids, top_vals, seen, B.PENALTY)
                if margin > 8.0 * float(g[f"{k0}_rms_err_bf16_full_vocab"][k]) and winner == rows[s][k]:
                    st["decided"] += 1
                    st["decided_equal"] += int(own == rows[s][k])
                n_, b16, t32 = lg[k][sid].astype(np.float64), g[f"{k0}_sample_vals_bf16"][k].astype(np.float64), g[f"{k0}_sample_vals_fp32"][k].astype(np.float64)
                ratios.append(float(np.sqrt(((n_ - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean())))
                seen.add(rows[s][k])
            past[s] = np.concatenate([fulls[s], np.asarray(rows[s][:-1], dtype=np.int64)])
    for x in states:
        x.release()
    ratios = np.asarray(ratios)
    record("livecc7b_batch8_vs_committed_golden", st | dict(worst_rms_ratio=float(ratios.max()), mean_rms_ratio=float(ratios.mean()),
                                                             rms_ratio_all_steps=float(np.sqrt((ratios ** 2).mean()))))
    assert st["steps"] == n_streams * 2 * n_new
    assert st["worst_rel_dlogit_top"] <= GROSS_MAX, st
    assert ratios.max() <= RMS_STEP_MAX_MANY and np.sqrt((ratios ** 2).mean()) <= RMS_ALL_MAX, (ratios.max(), ratios.mean())
    assert st["decided_equal"] == st["decided"], st
    assert st["tokens_equal"] >= st["steps"] - 10, st          # 64 steps of 8 streams: measured 56 (the undecided steps of flat synthetic logits)


def test_livecc7b_long480_stream_against_the_committed_hf_stream(dev):
    """BASELINE configs[3] in its STREAMING form at the real shapes: one 480-frame 280x280 video through the reference protocol -- a
    6-frame turn + 237 two-frame turns, 12 tokens each, the KV cache growing to ~31.9k keys -- against
    tests/golden/livecc7b_long480_stream.npz (HF bf16 free-running on the tiled:0 weights, oracle/make_golden_7b_long_stream.py).  The native
    engine follows HF's tokens through ALL 238 turns (teacher forcing: both caches hold the same history); at the probe turns (60, 120, 180
    and the last three: 8k / 16k / 24k / 31.5k cached keys) every step's raw logits at HF's top-64 ids stay within 6e-2 x scale and the
    native path's own choice equals HF's wherever HF's processed-score margin exceeds 8 % of the scale.  Round 5: at the
    probe turns 60 and 120 (8.3k / 16.3k cached keys) tests/golden/livecc7b_long480_stream_fp32.npz holds the fp32 TRUTH of the same
    teacher-forced stream (HF fp32 with its own fp32 cache built over all 121 turns, oracle/make_golden_7b_long_stream_fp32.py) and HF
    bf16 at 4,096 sample ids: rms(native - fp32) <= 1.25 x rms(HF_bf16 - fp32) on each of those 24 steps -- the incremental path over a
    deep cache now has the sensitive bound, not only the gross-error guard."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b_long_stream as L
    from oracle import make_golden_7b_long_stream_fp32 as L32
    g = dict(np.load(L.PATH))
    g32 = dict(np.load(L32.PATH))
    probes32 = set(int(x) for x in g32["probe_turns"])
    assert all(f"t{ti}_sample_vals_fp32" in g32 for ti in probes32), "the fp32 leg of the fixture generator was interrupted"
    sid32 = g32["sample_ids"]
    ratios32 = []
    seed, n_frames, H, W, n_new, seed_w, n_turns = (int(x) for x in g["meta"])
    assert "final_kv" in g, "the fixture generator was interrupted"
    cfg = get_config("livecc-7b")
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=32768, max_new_rows=1024, max_patches=4096, max_history=16)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed, layout="TCHW")).to(dev)
    builder = protocol.TurnBuilder(cfg, seed=seed)
    chunks = protocol.split_clip(n_frames)
    assert len(chunks) == n_turns
    probes = set(int(x) for x in g["probe_turns"])
    state, past = None, None
    st = dict(steps=0, tokens_equal=0, decided=0, decided_equal=0, worst_rel_dlogit_top=0.0, kv_at_probes=[])
    for ti, (a, b) in enumerate(chunks):
        new = builder.turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(b - a, H, W, cfg), cfg))
        ids = new if past is None else np.concatenate([past, new])
        gold = [int(x) for x in g["tokens"][ti]]
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], past_key_values=state, repetition_penalty=L.PENALTY,
                            max_new_tokens=n_new, min_new_tokens=n_new, output_logits=ti in probes, do_sample=False, teacher_tokens=gold)
        state = r.past_key_values
        if ti in probes:
            assert r.sequences[0, len(ids):].tolist() == gold
            st["kv_at_probes"].append(len(ids))
            lg = r.logits.float().cpu().numpy()
            seen = set(int(x) for x in ids)
            for k in range(n_new):
                scale = float(g[f"t{ti}_scale"][k])
                top_ids, top_vals = g[f"t{ti}_top_ids"][k].astype(np.int64), g[f"t{ti}_top_vals"][k].astype(np.float64)
                d = float(np.abs(lg[k][top_ids] - top_vals).max())
                st["steps"] += 1
                st["worst_rel_dlogit_top"] = max(st["worst_rel_dlogit_top"], d / scale)
                own = own_choice(lg[k], seen, L.PENALTY)
                st["tokens_equal"] += int(own == gold[k])
                winner, margin = processed_top2(top_ids, top_vals, seen, L.PENALTY)
                if margin > 0.08 * scale and winner == gold[k]:
                    st["decided"] += 1
                    st["decided_equal"] += int(own == gold[k])
                if ti in probes32:
                    n_, b16, t32 = (lg[k][sid32].astype(np.float64), g32[f"t{ti}_sample_vals_bf16"][k].astype(np.float64),
                                    g32[f"t{ti}_sample_vals_fp32"][k].astype(np.float64))
                    ratios32.append(float(np.sqrt(((n_ - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean())))
                seen.add(gold[k])
        past = np.concatenate([ids, np.asarray(gold[:-1], dtype=np.int64)])
    assert state.get_seq_length() == int(g["final_kv"]) == len(past)
    state.release()
    ratios32 = np.asarray(ratios32)
    st.update(fp32_probe_turns=sorted(probes32), fp32_steps=int(ratios32.size), worst_rms_ratio_vs_fp32=float(ratios32.max()),
              rms_ratio_vs_fp32_all_steps=float(np.sqrt((ratios32 ** 2).mean())))
    record("livecc7b_long480_stream_vs_committed_golden", st)
    assert ratios32.size == len(probes32) * n_new and ratios32.max() <= RMS_STEP_MAX_MANY, ratios32
    assert st["steps"] == len(probes) * n_new
    assert st["worst_rel_dlogit_top"] <= GROSS_MAX, st
    assert st["decided_equal"] == st["decided"], st
    assert st["tokens_equal"] >= st["steps"] - _undecided_slack(st), st      # 72 steps, 4 decided: at most 7 undecided mismatches (measured 4)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own orchestrator, executed (oracle/ref_infer_harness.py): its committed call trace through the native engine
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_orchestrator_trace_through_the_native_engine(dev):
    """tests/golden/ref_infer_trace.json holds what `ref demo/infer.py` ITSELF (loaded from /root/reference in the build container, HF's
    Qwen2VLForConditionalGeneration behind it at tiny shapes, its own ThresholdLogitsProcessor inside HF's generate) did for
    `live_cc_once_for_evaluation` (ref :244-310) and for the demo/cli.py loop over `live_cc` (ref :61-180) on a seeded synthetic video:
    every generate call's ids and generated tokens, and the responses.  Here `livecc_amd.infer.LiveCCDemoInfer` drives the NATIVE engine
    (same seeded weights, same tokenizer, same video, GPU resize + patchify + fused sampler) through the same two flows: every call must
    carry identical ids, generate identical tokens (HF's top-1 margin is >= 8 % of the logit scale on every step) and return identical
    strings -- the drop-in claim, demonstrated end to end."""
    import json
    from livecc_amd import video as V
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.text import TextFrontEnd
    from oracle import ref_infer_harness as H
    with open(H.TRACE_PATH) as f:
        want = json.load(f)
    tk = H.load_tokenizer()
    cfg = H.model_config(tk)
    hf = H.hf_weights(cfg).to(torch.bfloat16)                     # weights only: HF computes nothing here
    native = LiveCCForConditionalGeneration.from_hf_model(hf, cfg, dev, max_streams=1, max_kv_len=4096, max_new_rows=1024, max_patches=4096,
                                                          max_history=32)
    del hf
    calls, inner = [], native.generate

    def recording_generate(**kw):
        out = inner(**kw)
        ids = kw["input_ids"][0].tolist()
        calls.append(dict(input_ids=ids, tokens=out.sequences[0, len(ids):].tolist()))
        return out
    native.generate = recording_generate
    infer = LiveCCDemoInfer(model=native, text=TextFrontEnd(tk, cfg))
    assert infer.streaming_eos_token_id == want["streaming_eos_token_id"] and infer.system_prompt_offset == want["system_prompt_offset"]
    V.register_video(H.VIDEO_NAME, H.make_video())
    try:
        got = {}
        responses = H.scenario_once(infer)
        got["once"] = dict(responses=[[float(a), float(b), t] for a, b, t in responses], calls=list(calls))
        calls.clear()
        live, _ = H.scenario_live(infer)
        got["live"] = dict(responses=live, calls=list(calls))
    finally:
        V.unregister_video(H.VIDEO_NAME)
    stats = {}
    for sc in ("once", "live"):
        w, g = want[sc], got[sc]
        assert len(g["calls"]) == len(w["calls"]), (sc, len(g["calls"]), len(w["calls"]))
        for i, (a, b) in enumerate(zip(w["calls"], g["calls"])):
            assert a["input_ids"] == b["input_ids"], f"{sc} call {i}: the prompt ids differ from the executed reference's"
            assert a["tokens"] == b["tokens"], f"{sc} call {i}: tokens {b['tokens']} vs the reference's {a['tokens']}"
        assert g["responses"] == w["responses"], sc
        stats[sc] = dict(calls=len(w["calls"]), tokens=sum(len(c["tokens"]) for c in w["calls"]))
    record("reference_orchestrator_trace_vs_native", stats)
