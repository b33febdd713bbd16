"""-m gpu: the HIP path against the COMMITTED golden fixture (tests/golden/stream_tiny.npz, produced by oracle/make_golden.py
from the HF CPU oracle in the build container).  Same weights (seeded HF init), frames and ids; the fixture's tokens are the
bf16 oracle's free-running greedy tokens, so this also checks free-running token agreement, not only teacher-forced logits."""
import os

import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu
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
    assert rep["rms_ratio"] <= 1.25, rep
    assert rep["max_err_native"] <= 1.5 * rep["max_err_ref16"] + 1e-3 * rep["scale"], rep
    assert rep["max_abs_native_vs_ref16"] <= 6e-2 * rep["scale"], rep
    assert rep["argmax_native"] == rep["argmax_fp32"] == rep["argmax_ref16"], rep


def test_livecc7b_greedy_tokens_equal_the_committed_hf_tokens_on_decisive_weights(dev):
    """north_star: "token-id exact under greedy", at the REAL LiveCC-7B shapes against COMMITTED reference output
    (tests/golden/livecc7b_decisive_stream.npz, oracle/make_golden_7b.py --decisive: HF bf16, free-running, the benchmark protocol's
    6-frame + 2-frame turns, 16 greedy tokens each, repetition_penalty 1.05, on the `decisive:0` weights whose top-1 margin is 31 % of the
    logit scale at every step).  The HIP path, free-running on the same seeds, must emit the SAME 32 token ids; its raw logits at HF's
    top-8 ids stay within 6e-2 x scale of HF's.  No HF forward on the GPU box (the 97-s live version is
    test_greedy_tokens_are_exact_on_decisive_weights)."""
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
# Two more fixture-based comparisons, WRITTEN at the end of round 3 when no GPU minutes were left to run them once: they are skipped
# unless LCC_UNVALIDATED_GOLDEN=1 so that an untested test cannot stop the tier; the first GPU call of the next round runs them with
# the variable set and removes the gate.  (The two tests above were run on the MI355X before they were committed.)
# ---------------------------------------------------------------------------------------------------------------------
_UNVALIDATED = pytest.mark.skipif(os.environ.get("LCC_UNVALIDATED_GOLDEN") != "1",
                                  reason="written without a GPU run (round 3's GPU minutes were spent); LCC_UNVALIDATED_GOLDEN=1 runs it")


def _follow_golden_stream(native, cfg, g, frames, n_turns, max_new, top_key, sample_key16=None, sample_key32=None):
    """The native path along a committed free-running reference stream: per turn the prompt is the golden history + the turn ids; tokens
    and raw logits are compared step by step while the histories agree, and every turn continues along the GOLDEN tokens."""
    from livecc_amd import protocol
    builder = protocol.TurnBuilder(cfg, seed=int(g["meta"][0]))            # callers put the prompt seed in meta[0]
    for ti in range(n_turns):                                               # the fixture's prompts are rebuilt from the seed
        assert np.array_equal(g[f"t{ti}_ids"], np.asarray(builder.turn_ids(ti, protocol.num_video_tokens(tuple(int(x) for x in g[f"t{ti}_grid"]), cfg))))
    state, past = None, None
    stats = dict(steps=0, tokens_equal=0, decided=0, decided_equal=0, worst_rel_dlogit_top=0.0, ratios=[])
    sid = g.get("sample_ids")
    for ti, (a, b) in enumerate(protocol.split_clip(frames.shape[0])[:n_turns]):
        new = g[f"t{ti}_ids"]
        ids = new if past is None else np.concatenate([past, new])
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], past_key_values=state, repetition_penalty=1.05,
                            max_new_tokens=max_new, min_new_tokens=max_new, output_logits=True, do_sample=False)
        state = r.past_key_values
        toks = r.sequences[0, len(ids):].tolist()
        gold = g[f"t{ti}_tokens"].tolist()
        lg = r.logits.float().cpu().numpy()
        for k in range(max_new):
            if toks[:k] != gold[:k]:
                break
            stats["steps"] += 1
            stats["tokens_equal"] += int(toks[k] == gold[k])
            scale = float(g[f"t{ti}_scale"][k])
            top_ids, top_vals = g[f"t{ti}_top_ids"][k], g[f"t{ti}_{top_key}"][k]
            d = float(np.abs(lg[k][top_ids] - top_vals).max())
            stats["worst_rel_dlogit_top"] = max(stats["worst_rel_dlogit_top"], d / scale)
            # the reference's raw top-1 margin against twice the measured difference: where it decides, the tokens must agree
            if (top_vals[0] - top_vals[1]) > 2.0 * d and gold[k] == int(top_ids[0]):
                stats["decided"] += 1
                stats["decided_equal"] += int(toks[k] == gold[k])
            if sample_key32 is not None:
                n, b16, t32 = lg[k][sid].astype(np.float64), g[f"t{ti}_{sample_key16}"][k].astype(np.float64), g[f"t{ti}_{sample_key32}"][k].astype(np.float64)
                stats["ratios"].append(float(np.sqrt(((n - t32) ** 2).mean()) / np.sqrt(((b16 - t32) ** 2).mean())))
        past = np.concatenate([ids, np.asarray(gold[:-1], dtype=np.int64)])
        if toks != gold:                       # the carried cache now holds a different history: later turns are not comparable
            break
    if state is not None:
        state.release()
    return stats


@_UNVALIDATED
def test_livecc7b_two_turns_against_the_committed_hf_logits(dev):
    """BASELINE configs[1]'s first two turns at LiveCC-7B shapes (tiled:0 weights) against tests/golden/livecc7b_two_turns.npz: while the
    native free-running tokens follow HF's, per step |native - HF_bf16| at HF's top-64 ids <= 6e-2 x scale, rms over the 4,096 sample
    ids of (native - fp32) <= 1.25 x rms(HF_bf16 - fp32) (their rms tracks the full vocabulary within 5 %), tokens equal wherever HF's
    raw top-1 margin exceeds twice the measured difference; at least the first 8 steps are compared."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    from oracle import make_golden_7b as G
    g = dict(np.load(G.PATH_TURNS))
    seed_in, n_frames, H, W, max_new, seed_w = (int(x) for x in g["meta"])
    cfg = get_config("livecc-7b")
    arena = WeightArena(cfg, dev).fill_tiled(seed=seed_w)
    native = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=4096, max_new_rows=1280, max_patches=4608, max_history=16)
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed_in, layout="TCHW"))
    g2 = dict(g)
    g2["meta"] = np.asarray([seed_in])          # _follow_golden_stream reads the prompt seed from meta[0]
    st = _follow_golden_stream(native, cfg, g2, frames, int(g["n_turns"]), max_new, "top_vals_bf16", "sample_vals_bf16", "sample_vals_fp32")
    record("livecc7b_two_turns_vs_committed_golden", {k: v for k, v in st.items() if k != "ratios"} | dict(worst_rms_ratio=max(st["ratios"]),
           mean_rms_ratio=float(np.mean(st["ratios"]))))
    assert st["steps"] >= 8, st
    assert st["worst_rel_dlogit_top"] <= 6e-2, st
    assert max(st["ratios"]) <= 1.25, st
    assert st["decided_equal"] == st["decided"], st


@_UNVALIDATED
def test_qwen2vl2b_config0_against_the_committed_hf_stream(dev):
    """BASELINE configs[0] (Qwen2-VL-2B real shapes, 8-frame clip = 6 + 2 frames, greedy, 16 tokens per turn) against
    tests/golden/qwen2vl2b_config0_stream.npz without HF's forward on the GPU box (HF builds the seeded weights only): raw logits at HF's
    top-64 ids within 6e-2 x scale while the histories agree, tokens equal wherever HF's margin decides."""
    from livecc_amd import protocol
    from livecc_amd.config import qwen2vl_2b
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O, make_golden_2b as G
    g = dict(np.load(G.PATH))
    seed_w, seed_in, n_frames, H, W, max_new = (int(x) for x in g["meta"])
    cfg = qwen2vl_2b()
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=seed_w, init_scale=1.0)      # weights only
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=4096, max_new_rows=2048, max_patches=8192,
                                                          max_history=64)
    del hf16
    frames = torch.from_numpy(protocol.synth_frames(n_frames, H, W, seed=seed_in, layout="TCHW"))
    g2 = dict(g)
    g2["meta"] = np.asarray([seed_in])
    st = _follow_golden_stream(native, cfg, g2, frames, int(g["n_turns"]), max_new, "top_vals")
    record("qwen2vl2b_config0_vs_committed_golden", {k: v for k, v in st.items() if k != "ratios"})
    assert st["steps"] >= 8, st
    assert st["worst_rel_dlogit_top"] <= 6e-2, st
    assert st["decided_equal"] == st["decided"], st
