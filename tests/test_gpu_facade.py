"""-m gpu: the L4 facade with the REFERENCE's signatures and the batched multi-stream call against HF itself (VERDICT r1 items
a3 / a4 / f2): `live_cc_once_for_evaluation(query, video, video_start, video_end)`, `video_qa(message, history, state)` and
`generate_batch` of three streams are run natively and then replayed through the HF CPU oracle on the same frames and prompt ids,
teacher-forced along the native tokens.  Frames reach the oracle through torch's own CPU antialias-bicubic resize (oracle/resize_ref.py)
of the frames the reference's frame selection picks -- the native path resizes on the GPU (bit-exact, tests/test_gpu_resize.py)."""
import dataclasses

import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu


def _tok_cfg(tmp_path):
    from livecc_amd.config import tiny
    from livecc_amd.text import TextFrontEnd
    from tests.test_text import make_tokenizer_dir
    fe0 = TextFrontEnd(make_tokenizer_dir(tmp_path))
    tk = fe0.tokenizer
    cfg = dataclasses.replace(tiny(), video_token_id=fe0.video_token_id, eos_token_id=fe0.eos_token_id,
                              image_token_id=tk.convert_tokens_to_ids("<|image_pad|>"),
                              vision_start_token_id=tk.convert_tokens_to_ids("<|vision_start|>"),
                              vision_end_token_id=tk.convert_tokens_to_ids("<|vision_end|>"),
                              bos_token_id=tk.convert_tokens_to_ids("<|endoftext|>"))
    return tk, cfg


def _check_tokens_against_oracle(name, calls, oracle_turn, margin_rel=0.06):
    """calls: [(prompt_ids, pixel_values or None, grid or None, native tokens)].  The oracle is teacher-forced; a native token must be
    the oracle's own argmax whenever the oracle's top-1/top-2 margin exceeds `margin_rel` of its logit scale (twice the 3 % worst
    native-vs-bf16 logit difference measured by the logit-level tests), and at least 80 % of all tokens must be identical."""
    n = eq = checked = 0
    for ids, pv, grid, toks in calls:
        r = oracle_turn(ids, pv, grid, toks)
        assert r["new_tokens"] == toks
        for k, t in enumerate(toks):
            sc = r["scores"][k]
            ranked = torch.where(torch.isfinite(sc), sc, torch.full_like(sc, -1e30))
            top2 = torch.topk(ranked, 2)
            scale = r["logits"][k].abs().max().item()
            n += 1
            eq += int(int(top2.indices[0]) == t)
            if (top2.values[0] - top2.values[1]).item() > margin_rel * scale:
                checked += 1
                assert int(top2.indices[0]) == t, f"{name}: token {t} != oracle argmax {int(top2.indices[0])} at a decided step"
    record(name, dict(tokens=n, equal=eq, margin_checked=checked))
    assert eq >= 0.8 * n, f"{name}: only {eq}/{n} tokens equal the HF oracle's"


def test_live_cc_once_for_evaluation_with_the_reference_signature_vs_hf(dev, tmp_path):
    """ref demo/infer.py:244-310 end to end: a decoded 30-fps video, [video_start, video_end] cut by pts, ~2 fps linspace sampling
    (`_read_video_decord_plus`), per-nframes pixel budget + antialias resize (`_spatial_resize_video`), chunks 6/2/2..., the
    query on the first turn only, the checkpoint's generation_config deciding the sampling (greedy here), 'Time=a-bs' texts."""
    from livecc_amd import video as V
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.text import TextFrontEnd
    from oracle import hf_oracle as O
    from oracle.resize_ref import resize_ref
    tk, cfg = _tok_cfg(tmp_path)
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=6, init_scale=2.0)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=4096, max_new_rows=1024,
                                                          max_patches=4096, max_history=32)
    infer = LiveCCDemoInfer(model=native, text=TextFrontEnd(tk, cfg))
    g = torch.Generator().manual_seed(3)
    vid = V.DecodedVideo(torch.randint(0, 256, (240, 90, 120, 3), dtype=torch.uint8, generator=g), np.arange(240) / 30.0, 30.0)   # 8 s
    query = "what is happening now?"
    responses = infer.live_cc_once_for_evaluation(query=query, video=vid, video_start=1.0, video_end=6.0, max_new_tokens=5,
                                                  repetition_penalty=1.15)
    r2 = infer.live_cc_once_for_evaluation(query=query, video=vid, video_start=1.0, video_end=6.0, max_new_tokens=5,
                                           repetition_penalty=1.15)     # the KV slot was released: a second call works and repeats
    assert [x[:2] for x in responses] == [[1.0, 4.0], [4.0, 5.0], [5.0, 6.0]] and all(isinstance(x[2], str) for x in responses)
    assert [x[2] for x in r2] == [x[2] for x in responses]
    gen = infer.last_generated[-3:]
    # the oracle's inputs, built independently: reference frame selection -> torch CPU resize -> HF patchify restatement
    idxs, _, _ = V.select_video_frames(vid.pts, len(vid), vid.avg_fps, dict(video_start=1.0, video_end=6.0))
    assert len(idxs) == 10
    clip = vid.frames[idxs].permute(0, 3, 1, 2)
    h, w = V.spatial_resize_hw(90, 120, len(idxs))
    clip = resize_ref(clip, h, w)
    s16 = O.OracleStream(hf16, cfg)
    calls = []
    from livecc_amd import protocol
    for (a, b), gcall in zip(protocol.split_clip(clip.shape[0]), gen):
        pv, grid = O.patchify_normalize_ref(clip[a:b], cfg)
        calls.append((gcall["prompt_ids"], pv, grid, gcall["tokens"]))
    text0 = tk.decode(gen[0]["prompt_ids"].tolist(), skip_special_tokens=False)
    assert "Time=0.0-3.0s" in text0 and query in text0 and query not in tk.decode(gen[1]["prompt_ids"].tolist(), skip_special_tokens=False)
    _check_tokens_against_oracle("live_cc_once_for_evaluation_vs_hf", calls,
                                 lambda ids, pv, grid, toks: s16.turn(ids, pv, grid, max_new_tokens=len(toks), repetition_penalty=1.15,
                                                                      force_length=False, teacher_tokens=toks))


def test_video_qa_with_the_reference_signature_vs_hf(dev, tmp_path):
    """ref demo/infer.py:182-242: first turn = the whole video (frame selection + resize as `process_vision_info` with the 'decord+'
    backend does) in ONE prefill + the question; second turn = text only on the carried KV; hf_spaces = stateless."""
    from livecc_amd import protocol, video as V
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.text import TextFrontEnd
    from oracle import hf_oracle as O
    from oracle.resize_ref import resize_ref
    tk, cfg = _tok_cfg(tmp_path)
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=7, init_scale=2.0)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=4096, max_new_rows=512,
                                                          max_patches=1024, max_history=64)       # small limits: chunked ViT + prefill
    infer = LiveCCDemoInfer(model=native, text=TextFrontEnd(tk, cfg))
    g = torch.Generator().manual_seed(4)
    vid = V.DecodedVideo(torch.randint(0, 256, (150, 90, 120, 3), dtype=torch.uint8, generator=g), np.arange(150) / 30.0, 30.0)   # 5 s
    state = {"video_path": vid}
    resp1, state = infer.video_qa("what is happening now?", [], state, max_new_tokens=6, force_length=True)
    resp2, state = infer.video_qa("Please describe the video.", [], state, max_new_tokens=6, force_length=True)
    assert isinstance(resp1, str) and isinstance(resp2, str) and state["past_ids"] is not None
    g1, g2 = infer.last_generated[-2:]
    assert (g2["prompt_ids"] == cfg.video_token_id).sum() == 0 and (g1["prompt_ids"] == cfg.video_token_id).sum() > 0
    idxs, _, _ = V.select_video_frames(vid.pts, len(vid), vid.avg_fps, {})
    clip = vid.frames[idxs].permute(0, 3, 1, 2)
    h, w = V.spatial_resize_hw(90, 120, len(idxs))
    clip = resize_ref(clip, h, w)
    pv, grid = O.patchify_normalize_ref(clip, cfg)
    assert protocol.num_video_tokens(grid, cfg) == int((g1["prompt_ids"] == cfg.video_token_id).sum())
    s16 = O.OracleStream(hf16, cfg)
    calls = [(g1["prompt_ids"], pv, grid, g1["tokens"]), (g2["prompt_ids"], None, None, g2["tokens"])]
    _check_tokens_against_oracle("video_qa_vs_hf", calls,
                                 lambda ids, pv_, grid_, toks: s16.turn(ids, pv_, grid_, max_new_tokens=len(toks), repetition_penalty=1.05,
                                                                        teacher_tokens=toks))
    state["past_key_values"].release()
    # stateless mode: nothing is kept, the slot is free again
    st2 = {"video_path": vid}
    _, st2 = infer.video_qa("what is happening now?", [], st2, hf_spaces=True, max_new_tokens=3)
    assert st2["past_key_values"] is None and st2["past_ids"] is None
    assert infer.last_generated[-1]["tokens"][:3] == g1["tokens"][:3]


def _batched_streams_vs_hf(dev, cfg, shapes, seed0, name, max_new_tokens=5, init_scale=1.5, check=None):
    from livecc_amd import protocol
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O
    from tests.test_gpu_e2e import _compare_stream
    n = len(shapes)
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=2, init_scale=init_scale)
    hf32 = O.build_hf_model(cfg, dtype=torch.float32, seed=2, init_scale=init_scale)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=n, max_kv_len=2048, max_new_rows=max(1024, 256 * n),
                                                          max_patches=max(4096, 512 * n), max_history=16)
    frames = [torch.from_numpy(protocol.synth_frames(t, h, w, seed=seed0 + i, layout="TCHW")) for i, (t, h, w) in enumerate(shapes)]
    builders = [protocol.TurnBuilder(cfg, seed=seed0 + i) for i in range(n)]
    states, past, turns = [None] * n, [None] * n, [[] for _ in range(n)]
    for ti in range(2):
        reqs, meta = [], []
        for i in range(n):
            a, b = protocol.split_clip(frames[i].shape[0])[ti]
            clip = frames[i][a:b]
            grid = protocol.grid_of(b - a, clip.shape[2], clip.shape[3], cfg)
            new = builders[i].turn_ids(ti, protocol.num_video_tokens(grid, cfg))
            ids = new if past[i] is None else np.concatenate([past[i], new])
            reqs.append(dict(input_ids=torch.from_numpy(ids), frames=clip, frames_layout="TCHW", state=states[i]))
            meta.append((new, grid, len(ids), (a, b)))
        outs = native.generate_batch(reqs, repetition_penalty=1.05, max_new_tokens=max_new_tokens, force_length=True, output_logits=True)
        for i, o in enumerate(outs):
            new, grid, n_in, ab = meta[i]
            states[i] = o.past_key_values
            seq = o.sequences[0].cpu().numpy()
            past[i] = seq[:-1]
            turns[i].append(dict(turn_ids=new, grid=grid, new_tokens=seq[n_in:].tolist(), logits=o.logits.float().cpu(), frames=ab))
    for st in states:
        st.release()
    # the hard bounds (logit error, error ratio vs fp32, margin-guaranteed token identity) are asserted per stream; the two RATES are
    # pooled over the streams (a rate over the ~8 decided steps of one short stream is one mismatch away from any threshold)
    tot = dict(steps=0, exact=0, strict=0, strict_eq=0)
    for i in (range(n) if check is None else check):
        r = _compare_stream(cfg, hf16, hf32, turns[i], frames[i], f"{name}[{i}]", 1.05, strict_rate=None)
        for k in tot:
            tot[k] += r[k]
    from tests.util import record
    record(name + "[pooled]", tot)
    assert tot["exact"] >= 0.8 * tot["steps"], tot
    assert tot["strict_eq"] >= 0.95 * tot["strict"], tot


def test_generate_batch_of_three_streams_matches_hf_per_stream(dev):
    """SURVEY 8f-2 / VERDICT weak #8: ONE `generate_batch` call per turn for three streams with different clips (batched ViT, packed
    multi-stream prefill tiles, B = 3 decode steps) -- every stream against its own HF run (bf16 + fp32, teacher-forced): same logit
    bounds, error ratio and margin-aware token identity as the single-stream tests."""
    from livecc_amd.config import small
    _batched_streams_vs_hf(dev, small(), [(8, 56, 84), (8, 84, 56), (10, 56, 56)], 60, "generate_batch_3_streams")


def test_generate_batch_of_twenty_streams_in_one_decode_pass_matches_hf(dev):
    """VERDICT r1 missing #6: more than 16 streams decode in ONE weight pass per step (17..64 rows go through the 64-row GEMM tiles +
    the decode attention; `LCC_MAX_DECODE_BATCH`).  20 streams with different clips, two turns, every stream against its own HF run."""
    from livecc_amd.config import tiny
    shapes = [((8, 56, 84), (8, 84, 56), (10, 56, 56), (8, 56, 56))[i % 4] for i in range(20)]
    _batched_streams_vs_hf(dev, tiny(), shapes, 200, "generate_batch_20_streams", max_new_tokens=4, init_scale=2.0)


def test_video_qa_one_shot_480_frames_24k_tokens_at_small_vs_hf(dev):
    """SURVEY 8f-3 / BASELINE configs[3] at the REAL token geometry: `video_qa`'s first turn (ref demo/infer.py:182-242) over a 250-second
    30-fps video -> `smart_nframes` picks 480 frames (FPS_MAX_FRAMES), the per-nframes pixel budget gives 280x280 (82,320 px per frame:
    100 tokens per frame pair), i.e. 96,000 patches in 240 temporal slices and a 24,000-visual-token prompt in ONE generate call --
    served in pieces (ViT groups of <= 16,000 patches, prefill launch sequences of <= 4,096 rows over the carried KV), then 6 decode
    tokens at L ~ 24k.  `small` model dims (GQA 7:1, head dims 128 / 80); HF bf16 on the CPU prefills the same prompt in one pass
    (5.x text-offset rule on both sides: grid_t = 240 >> max(h, w) / 2) and is teacher-forced along the native tokens."""
    from livecc_amd import protocol, video as V
    from livecc_amd.config import small
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O
    from oracle.resize_ref import resize_ref
    cfg = small()
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=11, init_scale=1.5)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=24576 + 256, max_new_rows=4096,
                                                          max_patches=16000, max_history=16)
    infer = LiveCCDemoInfer(model=native)
    g = torch.Generator().manual_seed(8)
    n_src = 7500                                        # 250 s at 30 fps
    vid = V.DecodedVideo(torch.randint(0, 256, (n_src, 64, 64, 3), dtype=torch.uint8, generator=g), np.arange(n_src) / 30.0, 30.0)
    state = {"video_path": vid}
    resp, state = infer.video_qa(12, [], state, max_new_tokens=6, force_length=True)       # synthetic 12-id query
    g1 = infer.last_generated[-1]
    n_vid = int((g1["prompt_ids"] == cfg.video_token_id).sum())
    assert n_vid == 24000 and state["past_key_values"].get_seq_length() == len(g1["prompt_ids"]) + 5
    idxs, _, _ = V.select_video_frames(vid.pts, len(vid), vid.avg_fps, {})
    assert len(idxs) == 480
    h, w = V.spatial_resize_hw(64, 64, len(idxs))
    assert (h, w) == (280, 280)
    clip = resize_ref(vid.frames[idxs].permute(0, 3, 1, 2), h, w)
    pv, grid = O.patchify_normalize_ref(clip, cfg)
    assert tuple(grid)[0] == 240 and protocol.num_video_tokens(grid, cfg) == n_vid
    s16 = O.OracleStream(hf16, cfg)
    _check_tokens_against_oracle("video_qa_one_shot_480_frames_small_vs_hf", [(g1["prompt_ids"], pv, grid, g1["tokens"])],
                                 lambda ids, pv_, grid_, toks: s16.turn(ids, pv_, grid_, max_new_tokens=len(toks), repetition_penalty=1.05,
                                                                        teacher_tokens=toks))
    state["past_key_values"].release()
