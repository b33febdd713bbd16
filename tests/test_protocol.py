"""Host logic (no GPU): chunking, turn layout, M-RoPE position ids against HF's own get_rope_index, smart_resize."""
import numpy as np
import pytest
import torch

from livecc_amd import protocol
from livecc_amd.config import LiveCCConfig, livecc_7b, qwen2vl_2b, qwen2vl_72b, tiny


def test_split_clip_matches_reference_protocol():
    # ref demo/infer.py:121-129: first 6 frames, then 2 by 2
    assert protocol.split_clip(60) == [(0, 6)] + [(6 + 2 * i, 8 + 2 * i) for i in range(27)]
    assert protocol.split_clip(8) == [(0, 6), (6, 8)]
    assert protocol.split_clip(7) == [(0, 6), (6, 7)]            # torch.split keeps a short tail
    assert protocol.split_clip(4) == [(0, 4)]
    assert protocol.split_clip(5, initialized=True) == [(0, 2), (2, 4), (4, 5)]
    assert protocol.split_clip(0) == []


def test_smart_resize_examples():
    # 1080p with max_pixels = 384*28*28 (ref demo/cli.py:17) -> 392 x 728 (SURVEY geometry G1)
    assert protocol.smart_resize(1080, 1920, max_pixels=384 * 28 * 28, min_pixels=100 * 28 * 28) == (392, 728)
    h, w = protocol.smart_resize(100, 100, min_pixels=56 * 56)
    assert h % 28 == 0 and w % 28 == 0
    with pytest.raises(ValueError):
        protocol.smart_resize(10, 3000)


def test_param_counts_match_survey():
    c = livecc_7b()
    assert c.kv_bytes_per_token == 57344
    assert c.llm_layer_params() == 233057792 + 0 or abs(c.llm_layer_params() - 233.06e6) < 0.02e6
    assert abs(c.decode_weight_bytes() / 1e9 - 14.14) < 0.02
    assert qwen2vl_2b().kv_bytes_per_token == 28672 and qwen2vl_72b().kv_bytes_per_token == 327680
    for cfg in (c, qwen2vl_2b(), qwen2vl_72b(), tiny()):
        cfg.validate()


def test_turn_builder_structure():
    cfg = tiny()
    b = protocol.TurnBuilder(cfg, seed=3)
    t0 = b.turn_ids(0, 18)
    t1 = b.turn_ids(1, 6)
    assert (t0 == cfg.video_token_id).sum() == 18 and (t1 == cfg.video_token_id).sum() == 6
    assert t1[0] == cfg.eos_token_id                            # '<|im_end|>' glue (ref infer.py:150)
    i = int(np.argmax(t0 == cfg.vision_start_token_id))
    assert (t0[i + 1:i + 19] == cfg.video_token_id).all() and t0[i + 19] == cfg.vision_end_token_id
    assert len(t0) == 14 + 3 + 10 + 1 + 18 + 1 + 5 + 5 and len(t1) == 2 + 3 + 10 + 1 + 6 + 1 + 5
    mm = protocol.mm_token_type_ids(t0, cfg)
    assert mm.sum() == 2 * 18


@pytest.mark.parametrize("grid", [(3, 4, 6), (1, 4, 6), (1, 28, 52), (3, 28, 52), (8, 4, 4)])
def test_rope_index_matches_hf_get_rope_index(grid):
    """protocol.rope_index_first_turn == HF Qwen2VLModel.get_rope_index (5.15) on the same ids."""
    from oracle import hf_oracle as O
    cfg = tiny()
    hf = _hf_tiny()
    ids = protocol.TurnBuilder(cfg, seed=1).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    pos, delta = protocol.rope_index_first_turn(ids, [grid], cfg, "hf5")
    t = torch.as_tensor(ids).view(1, -1)
    ref_pos, ref_delta = hf.model.get_rope_index(t, mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(t.numpy(), cfg)),
                                                 video_grid_thw=torch.as_tensor([list(grid)]))
    assert np.array_equal(pos, ref_pos[:, 0].numpy())
    assert delta == int(ref_delta.item())
    # 4.5x rule differs only when grid_t > max(h, w) / 2
    pos4, delta4 = protocol.rope_index_first_turn(ids, [grid], cfg, "hf4")
    same = grid[0] <= max(grid[1], grid[2]) // 2
    assert np.array_equal(pos, pos4) == same


_HF = {}


def _hf_tiny():
    if "m" not in _HF:
        from oracle import hf_oracle as O
        _HF["m"] = O.build_hf_model(tiny(), torch.float32, seed=0)
    return _HF["m"]


def test_positions_with_cache():
    p = protocol.positions_with_cache(100, 4, -15)
    assert p.shape == (3, 4) and (p[0] == [85, 86, 87, 88]).all() and (p[0] == p[1]).all() and (p[1] == p[2]).all()


def test_config_round_trip_through_hf():
    cfg = tiny()
    back = LiveCCConfig.from_hf(cfg.to_hf(), name="tiny")
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
              "num_key_value_heads", "vit_depth", "vit_embed_dim", "vit_num_heads", "video_token_id", "eos_token_id"):
        assert getattr(cfg, f) == getattr(back, f), f


def test_long_one_shot_clip_positions_under_the_two_text_offset_rules():
    """SURVEY 8c-3 / ADVICE r1: a one-shot 480-frame 280x280 clip (grid 240 x 20 x 20 -> 24,000 video tokens): the released
    checkpoints were trained with transformers 4.5x, where the text after a vision block continues at max(position so far) + 1
    ("hf4", the default of from_pretrained); transformers 5.15 (the oracle) advances by max(h, w) // merge instead ("hf5").
    The rules agree for every streaming chunk (grid_t <= max(h,w)/2) and differ here."""
    from livecc_amd.config import livecc_7b
    cfg = livecc_7b()
    grid = (240, 20, 20)
    n_vid = protocol.num_video_tokens(grid, cfg)
    assert n_vid == 24000
    ids = np.asarray([1, 2, 3, cfg.vision_start_token_id] + [cfg.video_token_id] * n_vid + [cfg.vision_end_token_id, 4, 5, 6], dtype=np.int64)
    p4, d4 = protocol.rope_index_first_turn(ids, [grid], cfg, "hf4")
    p5, d5 = protocol.rope_index_first_turn(ids, [grid], cfg, "hf5")
    # identical up to the end of the vision block: text 0..3, then (t, h, w) indices offset by 4
    assert np.array_equal(p4[:, :4 + n_vid], p5[:, :4 + n_vid])
    assert p4[0, 4 + n_vid - 1] == 4 + 239 and p4[1, 4 + n_vid - 1] == 4 + 9 and p4[2, 4 + n_vid - 1] == 4 + 9
    # hf4: the next text token sits right after the largest position used so far (temporal axis: 4 + 239)
    assert p4[:, 4 + n_vid].tolist() == [244, 244, 244] and p4[:, -1].tolist() == [247, 247, 247]
    # hf5: it restarts max(h, w) // 2 = 10 after the block's start
    assert p5[:, 4 + n_vid].tolist() == [14, 14, 14]
    # rope_delta = max position + 1 - length; under hf4 the last prompt row holds the maximum, so the in-call decode position
    # (last row + 1) equals the next call's kv_len + rope_delta -- the two generation paths agree (they do not under hf5)
    S = len(ids)
    assert d4 == 248 - S and d5 == 244 - S
    assert int(p4[:, -1].max()) + 1 == S + d4
    assert int(p5[:, -1].max()) + 1 != S + d5
    # a streaming chunk: no difference
    g2 = (1, 28, 52)
    ids2 = np.asarray([1, cfg.vision_start_token_id] + [cfg.video_token_id] * protocol.num_video_tokens(g2, cfg) + [cfg.vision_end_token_id, 7], dtype=np.int64)
    a, da = protocol.rope_index_first_turn(ids2, [g2], cfg, "hf4")
    b, db = protocol.rope_index_first_turn(ids2, [g2], cfg, "hf5")
    assert np.array_equal(a, b) and da == db


def test_model_constructors_pick_the_documented_text_offset_rule():
    import inspect
    from livecc_amd.modeling import LiveCCForConditionalGeneration as M
    assert inspect.signature(M.__init__).parameters["text_offset_rule"].default == "hf4"     # released checkpoints (4.5x training)
    src = inspect.getsource(M.from_hf_model)
    assert 'setdefault("text_offset_rule", "hf5")' in src                                    # oracle parity = the installed transformers


def _hf45_get_rope_index(input_ids, video_grid_thw, cfg, image_grid_thw=()):
    """INDEPENDENT restatement of transformers 4.5x `Qwen2VLForConditionalGeneration.get_rope_index` (the version the released LiveCC
    checkpoints were trained with, ref README.md:30 'trained on transformers 4.50.0'), written from the published 4.5x algorithm --
    the index-search structure of the original, not the run-length scan of `protocol.rope_index_first_turn`: every vision block is
    located with `list.index`, the text before it continues at `previous block's max + 1`, the block's (t, h/merge, w/merge) meshgrid is
    offset by `text_len + st_idx`, trailing text continues at max + 1, delta = max + 1 - len."""
    m = cfg.spatial_merge_size
    ids = np.asarray(input_ids).reshape(-1)
    toks = ids.tolist()
    starts = np.argwhere(ids == cfg.vision_start_token_id).reshape(-1)
    vision_tokens = ids[starts + 1]
    image_nums, video_nums = int((vision_tokens == cfg.image_token_id).sum()), int((vision_tokens == cfg.video_token_id).sum())
    blocks, st, image_index, video_index = [], 0, 0, 0
    remain_images, remain_videos = image_nums, video_nums
    for _ in range(image_nums + video_nums):
        ed_image = toks.index(cfg.image_token_id, st) if (cfg.image_token_id in toks and remain_images > 0) else len(toks) + 1
        ed_video = toks.index(cfg.video_token_id, st) if (cfg.video_token_id in toks and remain_videos > 0) else len(toks) + 1
        if ed_image < ed_video:
            t, h, w = image_grid_thw[image_index]
            image_index += 1; remain_images -= 1; ed = ed_image
        else:
            t, h, w = video_grid_thw[video_index]
            video_index += 1; remain_videos -= 1; ed = ed_video
        gt, gh, gw = int(t), int(h) // m, int(w) // m
        text_len = ed - st
        st_idx = int(blocks[-1].max()) + 1 if blocks else 0
        blocks.append(np.broadcast_to(np.arange(text_len)[None, :], (3, text_len)) + st_idx)
        t_index = np.repeat(np.arange(gt), gh * gw)
        h_index = np.tile(np.repeat(np.arange(gh), gw), gt)
        w_index = np.tile(np.arange(gw), gt * gh)
        blocks.append(np.stack([t_index, h_index, w_index]) + text_len + st_idx)
        st = ed + gt * gh * gw
    if st < len(toks):
        st_idx = int(blocks[-1].max()) + 1 if blocks else 0
        n = len(toks) - st
        blocks.append(np.broadcast_to(np.arange(n)[None, :], (3, n)) + st_idx)
    pos = np.concatenate(blocks, axis=1)
    return pos.astype(np.int64), int(pos.max()) + 1 - len(toks)


def test_hf4_rule_against_an_independent_restatement_of_the_4_5x_algorithm():
    """The product's "hf4" positions (`protocol.rope_index_first_turn`, the default of `from_pretrained`) against
    `_hf45_get_rope_index` above on (a) the one-shot 480-frame video_qa / MCQ prompt (grid_t = 240 >> max(h, w) / 2: the case where
    4.5x and 5.15 differ), (b) the TRAINING layout of ref data/lmm_dataset.py:177-183 -- the whole interleaved conversation in one
    sequence, so every 6- / 2-frame chunk gets 3-D ids and advances the text position by max(t, h, w) only,
    (c) random mixtures of video blocks and text."""
    cfg = tiny()
    rng = np.random.RandomState(5)

    def seq(grids, text_lens):
        ids = []
        for g, n in zip(grids, text_lens):
            ids += rng.randint(0, 1000, size=n).tolist()
            ids += [cfg.vision_start_token_id] + [cfg.video_token_id] * protocol.num_video_tokens(g, cfg) + [cfg.vision_end_token_id]
        ids += rng.randint(0, 1000, size=text_lens[-1]).tolist()
        return np.asarray(ids, dtype=np.int64)
    cases = [([(240, 20, 20)], [27, 34]),                                           # (a) one-shot 480 frames at 280x280
             ([(3, 28, 52)] + [(1, 28, 52)] * 27, [27] + [22] * 27 + [5]),          # (b) the training layout of a 60-frame stream
             ([(3, 4, 6), (1, 4, 6), (120, 2, 2), (1, 8, 2)], [5, 1, 9, 3, 2])]     # (c)
    for _ in range(20):
        k = int(rng.randint(1, 5))
        cases.append(([(int(rng.randint(1, 40)), 2 * int(rng.randint(1, 8)), 2 * int(rng.randint(1, 8))) for _ in range(k)],
                      [int(rng.randint(1, 30)) for _ in range(k + 1)]))
    differ_from_hf5 = 0
    for grids, text_lens in cases:
        ids = seq(grids, text_lens)
        want, dwant = _hf45_get_rope_index(ids, grids, cfg)
        got, dgot = protocol.rope_index_first_turn(ids, grids, cfg, "hf4")
        assert np.array_equal(got, want) and dgot == dwant, (grids, text_lens)
        p5, _ = protocol.rope_index_first_turn(ids, grids, cfg, "hf5")
        differ_from_hf5 += int(not np.array_equal(p5, want))
    assert differ_from_hf5 >= 2, "the sweep must contain clips where the 4.5x and 5.15 text offsets differ"
    # (b): every chunk of the training layout advances the running position by max(t, h/2, w/2) -- not by its token count
    ids = seq(*cases[1])
    pos, _ = protocol.rope_index_first_turn(ids, cases[1][0], cfg, "hf4")
    assert int(pos.max()) + 1 < len(ids) // 4


def _hf_smart_resize():
    """HF's own `smart_resize` (HF models/qwen2_vl/video_processing_qwen2_vl.py:40-66 -- the function the reference's processor and
    qwen_vl_utils call), executed from its SOURCE FILE: the module itself cannot be imported here (it needs torchvision)."""
    import ast
    import math
    import os
    import transformers
    path = os.path.join(os.path.dirname(transformers.__file__), "models", "qwen2_vl", "video_processing_qwen2_vl.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "smart_resize")
    fn.returns = None
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    ns = {"math": math}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[])), path, "exec"), ns)
    return ns["smart_resize"]


def test_smart_resize_equals_hf_source_over_a_random_sweep():
    """`protocol.smart_resize` feeds BOTH the native path and the oracle's inputs, so it is pinned against HF's own
    function (executed from source, like tests/test_oracle.py does for `patchify`) over 4,000 random (height, width, pixel budget)
    triples incl. the streaming budgets of the reference (384*28*28 per frame, the 480-frame 24k-token budget) and the error case."""
    hf = _hf_smart_resize()
    rng = np.random.RandomState(0)
    budgets = [(56 * 56, 14 * 14 * 4 * 1280), (100 * 28 * 28, 384 * 28 * 28), (int(1.05 * 100 * 28 * 28), 82320), (4 * 28 * 28, 16384 * 28 * 28)]
    n = 0
    for _ in range(4000):
        h, w = int(rng.randint(20, 2400)), int(rng.randint(20, 4000))
        mn, mx = budgets[int(rng.randint(len(budgets)))]
        if rng.rand() < 0.3:
            mx = int(rng.randint(mn, 4 * 1280 * 28 * 28))
        try:
            want = hf(h, w, 28, mn, mx)
        except ValueError:
            with pytest.raises(ValueError):
                protocol.smart_resize(h, w, 28, mn, mx)
            continue
        assert protocol.smart_resize(h, w, 28, mn, mx) == tuple(want), (h, w, mn, mx)
        n += 1
    assert n > 3500
    assert protocol.smart_resize(1080, 1920, 28, 100 * 28 * 28, 384 * 28 * 28) == tuple(hf(1080, 1920, 28, 100 * 28 * 28, 384 * 28 * 28)) == (392, 728)
