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
