"""Pins the oracle (no GPU): the numpy/torch restatement of the HF video preprocessing against HF's own `patchify` source,
and the oracle's streaming replay against the committed golden fixture."""
import ast
import os

import numpy as np
import pytest
import torch

from livecc_amd import protocol
from livecc_amd.config import tiny
from oracle import hf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_tiny.npz")


def _hf_patchify():
    """HF's Qwen2VLVideoProcessor.patchify, executed from its source file without importing the module (the module needs
    torchvision, which is not installed): HF models/qwen2_vl/video_processing_qwen2_vl.py:236-274."""
    import transformers
    path = os.path.join(os.path.dirname(transformers.__file__), "models", "qwen2_vl", "video_processing_qwen2_vl.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "patchify")
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch}
    exec(compile(ast.fix_missing_locations(mod), path, "exec"), ns)
    return ns["patchify"]


@pytest.mark.parametrize("T,H,W", [(2, 56, 84), (6, 28, 56), (3, 56, 56)])
def test_patchify_restatement_matches_hf_source(T, H, W):
    cfg = tiny()
    f = torch.from_numpy(protocol.synth_frames(T, H, W, seed=2, layout="TCHW"))
    mine, grid = O.patchify_normalize_ref(f, cfg)
    mean, std = O.fused_mean_std()
    x = (f.float() - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)        # torchvision normalize: sub_ then div_
    flat, gt, gh, gw = _hf_patchify()(None, x.unsqueeze(0), cfg.patch_size, cfg.spatial_merge_size, cfg.temporal_patch_size)
    assert (gt, gh, gw) == grid
    assert torch.equal(mine, flat[0])


def test_fused_mean_std_constants():
    mean, std = O.fused_mean_std()
    from livecc_amd.engine import fused_mean_std
    m2, s2 = fused_mean_std()
    assert np.array_equal(mean.numpy(), m2) and np.array_equal(std.numpy(), s2)
    assert abs(mean[0].item() - 0.48145466 * 255) < 1e-4


def test_oracle_reproduces_golden_fixture():
    """Re-run the bf16 oracle on the fixture's inputs: same tokens (teacher-forced logits within bf16 noise across hosts)."""
    g = np.load(GOLDEN)
    cfg = tiny()
    seed_w, seed_in, frames_n, H, W, max_new = (int(x) for x in g["meta"])
    hf16 = O.build_hf_model(cfg, torch.bfloat16, seed_w, 2.0)
    frames = torch.from_numpy(protocol.synth_frames(frames_n, H, W, seed=seed_in, layout="TCHW"))
    st = O.OracleStream(hf16, cfg)
    for ti, (a, b) in enumerate(protocol.split_clip(frames_n)[:int(g["n_turns"])]):
        pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
        assert list(grid) == g[f"t{ti}_grid"].tolist()
        toks = g[f"t{ti}_tokens"].tolist()
        r = st.turn(g[f"t{ti}_ids"], pv, grid, max_new_tokens=max_new, repetition_penalty=1.05, teacher_tokens=toks)
        lg = torch.stack(r["logits"]).numpy()
        ref = g[f"t{ti}_logits_bf16"]
        assert np.abs(lg - ref).max() <= 0.05 * np.abs(ref).max()
        own = [int(s.argmax()) for s in r["scores"]]
        assert sum(int(x == y) for x, y in zip(own, toks)) >= len(toks) - 1
    pos, delta = protocol.rope_index_first_turn(g["t0_ids"], [g["t0_grid"].tolist()], cfg)
    assert np.array_equal(pos, g["t0_pos3"]) and delta == int(g["rope_delta"])


def test_bf16_oracle_is_close_to_fp32_oracle_in_fixture():
    """Quantifies the reference's own bf16 error (the scale every parity tolerance is expressed in)."""
    g = np.load(GOLDEN)
    for ti in range(int(g["n_turns"])):
        a, b = g[f"t{ti}_logits_bf16"], g[f"t{ti}_logits_fp32"]
        rel = np.abs(a - b).max() / np.abs(b).max()
        assert 1e-4 < rel < 0.06, rel


STAGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages_tiny.npz")


def test_layer_probe_reproduces_the_per_stage_fixture():
    """oracle/layer_probe.py (HF with recording / replacing hooks) re-run on the fixture's inputs: the bf16 run's per-stage tensors
    within bf16 noise of the committed ones (bit-equal on the same host), the teacher-forced fp32 run's within fp32 noise; and the
    structure the per-layer tests rely on: with overrides, every stage's recorded input IS the override, and the un-overridden
    chain is self-consistent (llm_in[l+1] == llm_out[l])."""
    from oracle import layer_probe as P
    from oracle.make_golden import STAGE_KEYS16, STAGE_KEYS32, load_stages
    a_ref, t_ref, g = load_stages(STAGES)
    cfg = tiny()
    seed_w, seed_in, n_frames, H, W = (int(x) for x in g["meta"])
    hf16 = O.build_hf_model(cfg, torch.bfloat16, seed_w, 2.0)
    hf32 = O.build_hf_model(cfg, torch.float32, seed_w, 2.0)
    frames = torch.from_numpy(protocol.synth_frames(10, H, W, seed=seed_in, layout="TCHW"))[:n_frames]
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=seed_in).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    assert np.array_equal(np.asarray(ids), g["ids"]) and list(grid) == g["grid"].tolist()
    a16 = P.probe(hf16, cfg, ids, pv, grid)
    for k in STAGE_KEYS16:
        d = (a16[k].float() - a_ref[k].float()).abs().max().item()
        assert d <= 0.05 * a_ref[k].float().abs().max().item(), (k, d)
    assert torch.equal(a16["llm_in"][1:], a16["llm_out"][:-1]) and torch.equal(a16["vit_in"][1:], a16["vit_out"][:-1])
    assert torch.equal(a16["final_in"], a16["llm_out"][-1]) and torch.equal(a16["merger_in"], a16["vit_out"][-1])
    t32 = P.probe(hf32, cfg, ids, pv, grid, P.inputs_of(a_ref))
    assert torch.equal(t32["llm_in"], a_ref["llm_in"].float()) and torch.equal(t32["vit_in"], a_ref["vit_in"].float())
    for k in STAGE_KEYS32:
        d = (t32[k] - t_ref[k]).abs().max().item()
        assert d <= 1e-4 * t_ref[k].abs().max().item() + 1e-6, (k, d)
    # the bf16 reference's own per-layer error against the teacher-forced truth: the unit of the per-layer parity bound
    for r in P.layer_error_table(a_ref, a_ref, t_ref, ["vit_out", "llm_mid", "llm_out"]):
        assert 1e-4 < r["err_ref16_rms"] / r["scale_rms"] < 2e-2, r


def test_livecc7b_golden_fixture_is_selfconsistent_and_rebuildable_from_its_seeds():
    """tests/golden/livecc7b_first_token.npz (oracle/make_golden_7b.py: HF at the REAL LiveCC-7B shapes, bf16 run + fp32 truth): the
    prompt ids are rebuilt from the stored seeds, the bf16 run is stored exactly (bit patterns), both runs pick the same token with a
    margin well above their difference, and the comparison helper the GPU test uses reports the bf16 run against itself as ratio 1."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from oracle import make_golden_7b as G
    fx = G.load()
    cfg = get_config("livecc-7b")
    seed_in, T, H, W, seed_w = (int(x) for x in fx["meta"])
    assert (T, H, W, seed_w) == (6, 392, 728, 0)
    grid = tuple(int(x) for x in fx["grid"])
    assert grid == (3, 28, 52) and protocol.num_video_tokens(grid, cfg) == 1092
    ids = protocol.TurnBuilder(cfg, seed=seed_in).turn_ids(0, 1092)
    assert np.array_equal(np.asarray(ids), fx["ids"]) and len(ids) == 1131
    assert fx["logits_fp32"].shape == (cfg.vocab_size,) and fx["logits_bf16"].shape == (cfg.vocab_size,)
    assert np.array_equal(torch.from_numpy(fx["logits_bf16"]).to(torch.bfloat16).float().numpy(), fx["logits_bf16"])     # bf16 numbers
    rep = G.compare(fx["logits_bf16"], fx)
    assert rep["rms_ratio"] == 1.0 and rep["max_abs_native_vs_ref16"] == 0.0
    assert rep["argmax_ref16"] == rep["argmax_fp32"] and rep["fp32_top1_margin"] > 2.0 * rep["max_err_ref16"]
    assert 0.02 < rep["rms_err_ref16"] / 1.0 < 0.08 and 4.0 < rep["scale"] < 7.0          # 0.8 % of the logit scale: bf16 after 28 layers


def test_livecc7b_decisive_golden_stream_is_decided_by_margin_and_rebuildable():
    """tests/golden/livecc7b_decisive_stream.npz: HF's committed free-running greedy tokens on the decisive weights at LiveCC-7B shapes.
    Every step's raw top-1 margin is tens of times the bf16 logit noise (>= 25 % of the logit scale against ~4 %), every emitted token is
    the raw top-1 unless the repetition penalty demoted it, and the turn prompts are rebuilt from the seeds."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "livecc7b_decisive_stream.npz")))
    cfg = get_config("livecc-7b")
    seed_in, n_frames, H, W, max_new, seed_w = (int(x) for x in g["meta"])
    assert (n_frames, H, W, max_new, seed_w, int(g["n_turns"])) == (8, 392, 728, 16, 0, 2)
    b = protocol.TurnBuilder(cfg, seed=seed_in)
    seen_all = []
    for ti in range(2):
        grid = tuple(int(x) for x in g[f"t{ti}_grid"])
        assert np.array_equal(g[f"t{ti}_ids"], np.asarray(b.turn_ids(ti, protocol.num_video_tokens(grid, cfg))))
        vals, idx, toks, scale = g[f"t{ti}_top_vals"], g[f"t{ti}_top_ids"], g[f"t{ti}_tokens"], g[f"t{ti}_scale"]
        assert toks.shape == (16,) and vals.shape == (16, 8)
        assert float(((vals[:, 0] - vals[:, 1]) / scale).min()) >= 0.25
        seen_all += g[f"t{ti}_ids"].tolist()
        for k in range(16):
            assert toks[k] == idx[k, 0] or int(idx[k, 0]) in seen_all, "a token other than the raw top-1 needs the repetition penalty"
            seen_all.append(int(toks[k]))
    assert len(set(g["t0_tokens"].tolist() + g["t1_tokens"].tolist())) >= 30       # the permutation walk does not loop


def test_qwen2vl2b_config0_golden_stream_is_rebuildable_from_its_seeds():
    """tests/golden/qwen2vl2b_config0_stream.npz (oracle/make_golden_2b.py: HF bf16 at the real Qwen2-VL-2B shapes, BASELINE configs[0],
    free-running): prompts rebuilt from the seeds, per-step top-64 sorted, the emitted token is the raw top-1 unless the repetition
    penalty demoted a seen id, sample ids reproducible."""
    from livecc_amd import protocol
    from livecc_amd.config import qwen2vl_2b
    from oracle import make_golden_2b as G
    g = dict(np.load(G.PATH))
    cfg = qwen2vl_2b()
    seed_w, seed_in, n_frames, H, W, max_new = (int(x) for x in g["meta"])
    assert (seed_w, n_frames, H, W, max_new, int(g["n_turns"])) == (0, 8, 392, 728, 16, 2)
    assert np.array_equal(g["sample_ids"], G.sample_ids(cfg.vocab_size))
    b = protocol.TurnBuilder(cfg, seed=seed_in)
    seen = []
    for ti in range(2):
        grid = tuple(int(x) for x in g[f"t{ti}_grid"])
        assert np.array_equal(g[f"t{ti}_ids"], np.asarray(b.turn_ids(ti, protocol.num_video_tokens(grid, cfg))))
        vals, idx, toks = g[f"t{ti}_top_vals"], g[f"t{ti}_top_ids"], g[f"t{ti}_tokens"]
        assert vals.shape == (16, 64) and (np.diff(vals, axis=1) <= 0).all() and g[f"t{ti}_sample_vals"].shape == (16, 2048)
        assert (np.abs(vals[:, 0]) <= g[f"t{ti}_scale"] + 1e-6).all()
        seen += g[f"t{ti}_ids"].tolist()
        for k in range(16):
            assert toks[k] == idx[k, 0] or int(idx[k, 0]) in seen
            seen.append(int(toks[k]))


def test_livecc7b_two_turn_golden_is_consistent_with_the_first_token_fixture():
    """tests/golden/livecc7b_two_turns.npz (make_golden_7b.py --turns: HF bf16 free-running + HF fp32 teacher-forced, LiveCC-7B shapes,
    tiled:0 weights, the benchmark's first two turns): its first step agrees with the full-vocabulary first-token fixture, the 4,096
    sample ids reproduce, and the rms error over the sample ids tracks the stored full-vocabulary rms within 5 % at every step (so a
    GPU test may use the samples for the error ratio)."""
    from livecc_amd.config import get_config
    from oracle import make_golden_7b as G
    g = dict(np.load(G.PATH_TURNS))
    fx = G.load()
    cfg = get_config("livecc-7b")
    sid = g["sample_ids"]
    assert np.array_equal(sid, G.sample_ids(cfg.vocab_size)) and int(g["n_turns"]) == 2
    assert np.array_equal(g["t0_ids"], fx["ids"])
    # step 0 of turn 0 = the first-token fixture (same weights, same prompt; the bf16 run is deterministic on one host)
    assert np.array_equal(g["t0_sample_vals_bf16"][0], fx["logits_bf16"][sid])
    assert np.abs(g["t0_sample_vals_fp32"][0] - fx["logits_fp32"][sid]).max() <= 1e-4
    assert int(g["t0_tokens"][0]) == int(fx["logits_bf16"].argmax())
    for ti in range(2):
        e = np.sqrt(((g[f"t{ti}_sample_vals_bf16"] - g[f"t{ti}_sample_vals_fp32"]).astype(np.float64) ** 2).mean(axis=-1))
        full = g[f"t{ti}_rms_err_bf16_full_vocab"].astype(np.float64)
        assert e.shape == (16,) and (np.abs(e / full - 1.0) <= 0.05).all(), (e / full)
        assert (np.diff(g[f"t{ti}_top_vals_bf16"], axis=1) <= 0).all()


def test_the_fixture_following_helper_of_the_gpu_golden_tests_on_a_replaying_fake():
    """tests/test_gpu_golden.py::follow_golden_stream (used by the fixture-based 7B / 2B GPU tests) driven by a FAKE native model that
    replays the committed reference's logits and honours teacher forcing: 32 steps compared, every own choice equal, rms ratio exactly 1,
    zero logit difference; a fake whose logits prefer the reference's second choice on one step shows up as a token mismatch -- counted
    against the margin rule only when the reference's own margin decides that step."""
    import types
    from oracle import make_golden_7b as G
    from tests import test_gpu_golden as T
    g = dict(np.load(G.PATH_TURNS))
    seed_in, n_frames, H, W, max_new, _ = (int(x) for x in g["meta"])
    V, sid = 152064, g["sample_ids"]

    class Fake:
        def __init__(self, flip=None):
            self.turn, self.flip = 0, flip

        def generate(self, input_ids, frames, past_key_values, max_new_tokens, teacher_tokens=None, **kw):
            ti = self.turn
            self.turn += 1
            lg = np.full((max_new_tokens, V), -30.0, dtype=np.float32)
            for k in range(max_new_tokens):
                lg[k][sid] = g[f"t{ti}_sample_vals_bf16"][k]
                lg[k][g[f"t{ti}_top_ids"][k]] = g[f"t{ti}_top_vals_bf16"][k]
            if self.flip is not None and self.flip[0] == ti:
                k = self.flip[1]
                lg[k][g[f"t{ti}_top_ids"][k][1]] = g[f"t{ti}_top_vals_bf16"][k][0] + 1.0     # prefers the reference's second choice
            seq = torch.cat([input_ids[0], torch.as_tensor(teacher_tokens, dtype=torch.long)]).view(1, -1)
            return types.SimpleNamespace(sequences=seq, logits=torch.from_numpy(lg), past_key_values=types.SimpleNamespace(release=lambda: None))

    turns = T._turns_of(g, 2, "top_vals_bf16")

    def noise(ti, k):
        return 8.0 * float(g[f"t{ti}_rms_err_bf16_full_vocab"][k])
    sample = (sid, "sample_vals_bf16", "sample_vals_fp32")
    st = T.follow_golden_stream(Fake(), turns, lambda ti: None, max_new, "top_vals_bf16", noise, sample=sample)
    # the replayed RAW logits + the host-side penalty reproduce HF's own greedy choice on every step
    assert (st["steps"], st["tokens_equal"]) == (32, 32) and st["worst_rel_dlogit_top"] == 0.0
    assert max(st["ratios"]) == 1.0 and min(st["ratios"]) == 1.0 and st["decided_equal"] == st["decided"] > 0
    undecided = [(ti, k) for ti in range(2) for k in range(max_new)
                 if not (g[f"t{ti}_top_vals_bf16"][k][0] - g[f"t{ti}_top_vals_bf16"][k][1] > noise(ti, k))]
    assert undecided, "random weights leave undecided steps"
    st = T.follow_golden_stream(Fake(flip=undecided[0]), turns, lambda ti: None, max_new, "top_vals_bf16", noise, sample=sample)
    assert st["steps"] == 32 and st["tokens_equal"] == 31 and st["decided_equal"] == st["decided"] and st["undecided_mismatch_steps"] == [undecided[0]]


def test_livecc7b_oneshot480_fixture_is_selfconsistent_and_rebuildable_from_its_seeds():
    """tests/golden/livecc7b_oneshot480.npz (oracle/make_golden_7b_long.py: the executed HF reference at BASELINE configs[3]'s real shapes --
    480 frames, 24,058-row prompt, 8 tokens at L ~ 24k; bf16 free-running + fp32 teacher-forced): the prompt is rebuilt from the seeds, the
    bf16 run's own greedy tokens equal its top-1 ids after the repetition penalty, and the 4,096-id sample tracks the full-vocabulary rms
    of (bf16 - fp32) within 5 % on every step (the statistic the GPU test uses)."""
    from livecc_amd.config import get_config
    from oracle import make_golden_7b_long as L
    from tests.test_gpu_golden import own_choice
    g = dict(np.load(L.PATH))
    cfg = get_config("livecc-7b")
    ids, grid = L.prompt_ids(cfg)
    assert len(ids) == int(g["ids_len"]) == 24058 and tuple(grid) == (240, 20, 20) and int((ids == cfg.video_token_id).sum()) == 24000
    assert np.array_equal(ids[:64], g["ids_head"]) and np.array_equal(ids[-64:], g["ids_tail"])
    assert np.array_equal(g["sample_ids"], L.sample_ids(cfg.vocab_size))
    n = int(g["meta"][4])
    assert g["tokens"].shape == (n,) and g["t0_top_ids"].shape == (n, L.TOPK) and g["t0_sample_vals_fp32"].shape == (n, L.NSAMPLE)
    assert (np.diff(g["t0_top_vals_bf16"], axis=1) <= 0).all()
    e = np.sqrt(((g["t0_sample_vals_bf16"] - g["t0_sample_vals_fp32"]).astype(np.float64) ** 2).mean(axis=-1))
    full = g["t0_rms_err_bf16_full_vocab"].astype(np.float64)
    assert (np.abs(e / full - 1.0) <= 0.05).all(), e / full
    # the committed tokens are the bf16 run's own greedy choices: rebuild each step's decision from its top-64 raw logits + the penalty
    seen = set(int(x) for x in ids)
    for k in range(n):
        lg = np.full(cfg.vocab_size, -1e30, dtype=np.float32)
        lg[g["t0_top_ids"][k]] = g["t0_top_vals_bf16"][k]
        assert own_choice(lg, seen, L.PENALTY) == int(g["tokens"][k]), k
        seen.add(int(g["tokens"][k]))


def test_livecc7b_batch8_fixture_is_selfconsistent_and_rebuildable_from_its_seeds():
    """tests/golden/livecc7b_batch8.npz (oracle/make_golden_7b_batch8.py: HF once per stream for the 8 benchmark streams, two turns, 4 tokens
    each; bf16 free-running + fp32 teacher-forced at the real LiveCC-7B shapes): prompts rebuilt from the seeds, tokens = the bf16 run's own
    greedy choices, the 1,024-id sample tracks the full-vocabulary rms of (bf16 - fp32) within 10 %, streams differ from each other."""
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from oracle import make_golden_7b_batch8 as B
    from tests.test_gpu_golden import own_choice
    g = dict(np.load(B.PATH))
    seed0, n_streams, n_frames, H, W, n_new, _ = (int(x) for x in g["meta"])
    assert (seed0, n_streams, n_frames, H, W, n_new) == (B.SEED0, B.STREAMS, B.N_FRAMES, B.H, B.W, B.N_NEW)
    cfg = get_config("livecc-7b")
    assert np.array_equal(g["sample_ids"], B.sample_ids(cfg.vocab_size))
    first = []
    for s in range(n_streams):
        builder = protocol.TurnBuilder(cfg, seed=seed0 + s)
        history = []                                     # cat(past_ids, new_ids): what the repetition penalty sees (ref demo/infer.py:160,169)
        for ti, (a, b) in enumerate(protocol.split_clip(n_frames)[:2]):
            k0 = f"s{s}_t{ti}"
            ids = builder.turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(b - a, H, W, cfg), cfg))
            assert np.array_equal(g[f"{k0}_ids"], ids)
            history += [int(x) for x in ids]
            seen = set(history)
            e = np.sqrt(((g[f"{k0}_sample_vals_bf16"] - g[f"{k0}_sample_vals_fp32"]).astype(np.float64) ** 2).mean(axis=-1))
            full = g[f"{k0}_rms_err_bf16_full_vocab"].astype(np.float64)
            assert e.shape == (n_new,) and (np.abs(e / full - 1.0) <= 0.10).all(), (s, ti, e / full)
            for k in range(n_new):
                lg = np.full(cfg.vocab_size, -1e30, dtype=np.float32)
                lg[g[f"{k0}_top_ids"][k]] = g[f"{k0}_top_vals_bf16"][k]
                assert own_choice(lg, seen, B.PENALTY) == int(g[f"{k0}_tokens"][k]), (s, ti, k)
                seen.add(int(g[f"{k0}_tokens"][k]))
            history += [int(x) for x in g[f"{k0}_tokens"][:-1]]        # past_ids = sequences[:, :-1]: the last token never enters the history
        first.append(tuple(g[f"s{s}_t0_top_vals_bf16"][0][:4].round(3)))
    assert len(set(first)) == n_streams, "eight different streams"


def test_livecc7b_long480_stream_fixture_is_selfconsistent_and_rebuildable_from_its_seeds():
    """tests/golden/livecc7b_long480_stream.npz (oracle/make_golden_7b_long_stream.py: the executed HF bf16 reference for BASELINE
    configs[3] in its streaming form, 238 turns at the real LiveCC-7B shapes): complete (the generator writes `final_kv` last), the cache
    bookkeeping follows the reference's `past_ids = sequences[:, :-1]` rule (350 + 237 x 133 keys = 31,871 at the end), and at every probe turn the
    committed tokens are the run's own greedy choices after the repetition penalty over the whole history."""
    from livecc_amd.config import get_config
    from oracle import make_golden_7b_long_stream as L
    from tests.test_gpu_golden import own_choice
    g = dict(np.load(L.PATH))
    assert "final_kv" in g, "the fixture generator was interrupted: regenerate with python oracle/make_golden_7b_long_stream.py"
    seed, n_frames, H, W, n_new, seed_w, n_turns = (int(x) for x in g["meta"])
    assert (seed, n_frames, H, W, n_new, seed_w) == (L.SEED, L.N_FRAMES, L.H, L.W, L.N_NEW, 0)
    cfg = get_config("livecc-7b")
    chunks = protocol.split_clip(n_frames)
    assert len(chunks) == n_turns == 238 and g["tokens"].shape == (n_turns, n_new)
    probes = [int(x) for x in g["probe_turns"]]
    assert probes == L.probe_turns(n_turns) == [60, 120, 180, 235, 236, 237]
    builder = protocol.TurnBuilder(cfg, seed=seed)
    history, kv = [], 0
    for ti, (a, b) in enumerate(chunks):
        ids = builder.turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(b - a, H, W, cfg), cfg))
        if ti in probes:
            assert int(g[f"t{ti}_kv_before"]) == kv, (ti, kv)
            assert g[f"t{ti}_top_ids"].shape == (n_new, L.TOPK) and (np.diff(g[f"t{ti}_top_vals"], axis=1) <= 0).all()
            assert (g[f"t{ti}_scale"] >= np.abs(g[f"t{ti}_top_vals"]).max(axis=1) - 1e-6).all()
            seen = set(history) | set(int(x) for x in ids)
            for k in range(n_new):
                lg = np.full(cfg.vocab_size, -1e30, dtype=np.float32)
                lg[g[f"t{ti}_top_ids"][k]] = g[f"t{ti}_top_vals"][k]
                assert own_choice(lg, seen, L.PENALTY) == int(g["tokens"][ti][k]), (ti, k)
                seen.add(int(g["tokens"][ti][k]))
        history += [int(x) for x in ids] + [int(x) for x in g["tokens"][ti][:-1]]
        kv += len(ids) + n_new - 1
    assert kv == int(g["final_kv"]) == 350 + 237 * 133 == 31871
    assert len(set(map(tuple, g["tokens"]))) > n_turns // 2, "the stream's turns differ from each other"
