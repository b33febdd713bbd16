"""Host logic of the `LiveCCDemoInfer` facade with the REFERENCE's signatures (ref demo/infer.py:61-310), on a stand-in model
(no GPU): pacing from state['video_path'] / state['video_timestamp'], chunking, query-on-change, slot release, stateless
hf_spaces mode.  The arithmetic behind `generate` is covered by the -m gpu tests."""
import numpy as np
import pytest
import torch

from livecc_amd import infer as I, protocol, resize as R, video as V
from livecc_amd.config import tiny
from livecc_amd.modeling import GenerateOutput, StreamState


class _Engine:
    max_slots, max_history = 1, 512

    def __init__(self):
        self.len = {0: 0}

    def slot_length(self, slot):
        return self.len[slot], 0


class FakeModel:
    def __init__(self):
        self.cfg = tiny()
        self.device = torch.device("cpu")
        self.engine = _Engine()
        self._free_slots = [0]
        self.eos_token_ids = [self.cfg.eos_token_id]
        self.generation_config = {}
        self.calls = []

    def new_stream(self):
        if not self._free_slots:
            raise RuntimeError("all 1 stream slots are in use (raise max_streams)")
        st = StreamState(self, self._free_slots.pop())
        self.engine.len[st.slot] = 0
        return st

    def generate(self, input_ids=None, frames=None, past_key_values=None, max_new_tokens=16, **kw):
        st = past_key_values or self.new_stream()
        assert input_ids.shape[1] > st.get_seq_length()
        new = torch.arange(max_new_tokens) + 7
        seq = torch.cat([input_ids[0], new])
        self.engine.len[st.slot] = len(seq) - 1
        self.calls.append(dict(n_in=input_ids.shape[1], frames=None if frames is None else tuple(frames.shape), kw=kw,
                               max_new_tokens=max_new_tokens, had_state=past_key_values is not None))
        return GenerateOutput(sequences=seq.view(1, -1), past_key_values=st)


@pytest.fixture()
def infer(monkeypatch):
    def fake_clip(frames, h, w, ts, pts, index_from, layout="THWC", device=None):
        idxs, kept = R.select_clip_frames(ts, pts, index_from)
        return (torch.zeros(len(idxs), 3, h, w, dtype=torch.uint8) if idxs else None), kept, idxs
    monkeypatch.setattr(R, "get_smart_resized_clip", fake_clip)

    def fake_resize(video, nframes=None, device=None):
        n = nframes or video.shape[0]
        h, w = V.spatial_resize_hw(int(video.shape[2]), int(video.shape[3]), int(n))
        return torch.zeros(video.shape[0], 3, h, w, dtype=torch.uint8)
    monkeypatch.setattr(V, "spatial_resize_video", fake_resize)
    return I.LiveCCDemoInfer(model=FakeModel())


def _video(n, fps=30.0, h=60, w=90):
    return V.DecodedVideo(torch.zeros(n, h, w, 3, dtype=torch.uint8), np.arange(n) / fps, fps)


def test_live_cc_runs_the_reference_cli_loop_unmodified(infer):
    """ref demo/cli.py:10-24 verbatim, given a decoded video registered under the path."""
    V.register_video("demo/sources/x.mp4", _video(300))            # 10 s
    try:
        state = {"video_path": "demo/sources/x.mp4"}
        commentaries = []
        for t in range(31):
            state["video_timestamp"] = t
            for (start_t, stop_t), response, state in infer.live_cc(
                    message="Please describe the video.", state=state, max_pixels=4 * 28 * 28, repetition_penalty=1.05,
                    streaming_eos_base_threshold=0.0, streaming_eos_threshold_step=0):
                commentaries.append([start_t, stop_t, response])
            if state.get("video_end", False):
                break
    finally:
        V.unregister_video("demo/sources/x.mp4")
    spans = [(a, b) for a, b, _ in commentaries]
    assert spans[0] == (0.0, 3.0) and spans[1:] == [(3.0 + i, 4.0 + i) for i in range(len(spans) - 1)]
    assert spans[-1][1] <= 10.0 and state.get("video_end")
    calls = infer.model.calls
    assert calls[0]["frames"][0] == 6 and all(c["frames"][0] == 2 for c in calls[1:])
    assert calls[0]["kw"]["do_sample"] is True and calls[0]["max_new_tokens"] == 16          # ref defaults (:68, :170)
    assert not calls[0]["had_state"] and all(c["had_state"] for c in calls[1:])
    assert state["last_timestamp"] == pytest.approx(spans[-1][1] - 0.5)


def test_live_cc_without_a_video_or_before_the_next_frame_yields_nothing(infer):
    assert list(infer.live_cc("q", {})) == []
    vid = _video(300)
    state = {"video_path": vid, "video_timestamp": 0}
    first = list(infer.live_cc("q", state))
    assert [s for s, _, _ in first] == [(0.0, 3.0)]                 # the first call always takes the initial 3 s
    state["video_timestamp"] = 3.0
    assert list(infer.live_cc("q", state)) == []                    # video clock has not passed the next frame time yet
    state["video_timestamp"] = 3.2
    assert [s for s, _, _ in infer.live_cc("q", state)] == [(3.0, 4.0)]
    light = list(infer.live_cc("q", dict(state, video_timestamp=4.7), hf_spaces=True))[0][2]
    assert "past_key_values" not in light and "past_ids" not in light and "last_timestamp" in light


def test_live_cc_once_for_evaluation_twice_on_one_slot(infer):
    """the KV slot of an offline replay must be free again when the call returns -- the offline driver calls
    it once per record on a max_streams=1 model."""
    vid = _video(600)                                               # 20 s at 30 fps
    r1 = infer.live_cc_once_for_evaluation(query="commentate", video=vid, video_start=2.0, video_end=9.0, max_new_tokens=5,
                                           repetition_penalty=1.15)
    assert infer.model._free_slots == [0]
    r2 = infer.live_cc_once_for_evaluation(query="commentate", video=vid, video_start=0, video_end=None)
    assert infer.model._free_slots == [0]
    # 7 s at ~2 fps -> 14 frames -> chunks 6,2,2,2,2; times are offset by video_start (ref :305-308)
    assert [(a, b) for a, b, _ in r1] == [(2.0, 5.0), (5.0, 6.0), (6.0, 7.0), (7.0, 8.0), (8.0, 9.0)]
    assert r2[0][:2] == [0.0, 3.0] and len(r2) == 1 + (40 - 6) // 2
    calls = infer.model.calls
    assert calls[0]["max_new_tokens"] == 5 and calls[0]["kw"]["repetition_penalty"] == 1.15
    assert calls[0]["kw"]["do_sample"] is None                       # not passed by the reference: generation_config decides
    assert calls[len(r1)]["max_new_tokens"] == 32 and not calls[len(r1)]["had_state"]
    # an exception inside the loop must not leak the slot either
    def boom(*a, **k):
        raise RuntimeError("x")
    infer.model.generate = boom
    with pytest.raises(RuntimeError):
        infer.live_cc_once_for_evaluation(query="q", video=vid)
    assert infer.model._free_slots == [0]


def test_video_qa_reference_signature(infer):
    vid = _video(300)
    state = {"video_path": vid}
    resp, state = infer.video_qa(5, [], state, max_new_tokens=4)     # synthetic mode: message = query length
    c0 = infer.model.calls[0]
    assert c0["frames"] is not None and c0["frames"][0] == 20 and c0["kw"]["do_sample"] is False
    assert state["past_ids"] is not None and isinstance(resp, str)
    resp2, state = infer.video_qa(3, [], state, max_new_tokens=4)
    c1 = infer.model.calls[1]
    assert c1["frames"] is None and c1["had_state"] and c1["n_in"] > c0["n_in"]
    # default budget of the reference: 512 new tokens
    infer.video_qa(3, [], state)
    assert infer.model.calls[2]["max_new_tokens"] == 512
    # stateless mode keeps nothing and frees the slot
    st2 = {"video_path": vid}
    state["past_key_values"].release()
    _, st2 = infer.video_qa(4, [], st2, hf_spaces=True, max_new_tokens=2)
    assert st2["past_key_values"] is None and st2["past_ids"] is None and infer.model._free_slots == [0]


def test_clip_level_entry_point_keeps_the_stream_clock(infer):
    state = {}
    clip = torch.zeros(10, 3, 56, 84, dtype=torch.uint8)
    spans = [s for s, _, _ in infer.live_cc_clip(clip, state, max_new_tokens=3)]
    assert spans == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0)] and state["last_timestamp"] == 4.5
    spans = [s for s, _, _ in infer.live_cc_clip(clip[:4], state, max_new_tokens=3)]
    assert spans == [(5.0, 6.0), (6.0, 7.0)]


def test_prefill_packs_streams_into_groups_under_max_new_rows():
    """`_prefill` host logic: streams are packed into consecutive groups of <= max_new_rows rows, a stream longer than that is cut
    into pieces alone, ViT rows follow their streams, and step-0 logits / scores land in the rows of their streams."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from livecc_amd.modeling import LiveCCForConditionalGeneration as M
    VID = 9
    calls = []

    class Eng:
        max_new_rows = 100

        def prefill(self, slots, ids, pos3, vit, sp, scores_out=None, logits_out=None):
            nv = sum(int((np.asarray(x) == VID).sum()) for x in ids)
            assert (vit is None and nv == 0) or vit.shape[0] == nv
            assert sum(len(x) for x in ids) <= self.max_new_rows
            for x, p in zip(ids, pos3):
                assert p.shape == (3, len(x))
            calls.append(dict(slots=list(slots), rows=[len(x) for x in ids], vit=None if vit is None else vit[:, 0].tolist(),
                              sc=None if scores_out is None else tuple(scores_out.shape), lg=None if logits_out is None else tuple(logits_out.shape)))
            if logits_out is not None:
                logits_out[:, 0] = torch.tensor([float(s) for s in slots])

    m = M.__new__(M)
    m.cfg = SimpleNamespace(video_token_id=VID, image_token_id=-5)
    m.engine = Eng()
    lens = [40, 50, 30, 250, 60, 10]
    nvid = [10, 0, 5, 120, 20, 0]
    ids = [np.concatenate([np.full(v, VID), np.arange(n - v) + 20]).astype(np.int64) for n, v in zip(lens, nvid)]
    pos = [np.zeros((3, n), dtype=np.int64) for n in lens]
    vit = torch.arange(sum(nvid), dtype=torch.float32).view(-1, 1).repeat(1, 2)
    slots = [7, 3, 5, 1, 2, 0]
    logits = torch.full((4, 6, 8), -1.0)
    scores = torch.zeros(6, 8)
    m._prefill(slots, ids, pos, vit, None, scores, logits)
    assert [c["slots"] for c in calls] == [[7, 3], [5], [1], [1], [1], [2, 0]]
    assert [c["rows"] for c in calls] == [[40, 50], [30], [100], [100], [50], [60, 10]]
    assert calls[0]["vit"] == list(range(0, 10)) and calls[1]["vit"] == list(range(10, 15))
    assert calls[2]["vit"] == list(range(15, 115)) and calls[3]["vit"] == list(range(115, 135)) and calls[4]["vit"] is None
    assert calls[5]["vit"] == list(range(135, 155))
    assert [c["lg"] for c in calls] == [(2, 8), (1, 8), (1, 8), (1, 8), (1, 8), (2, 8)] and calls[0]["sc"] == (2, 8)
    assert logits[0, :, 0].tolist() == [7.0, 3.0, 5.0, 1.0, 2.0, 0.0] and (logits[1:] == -1).all()
    calls.clear()
    m._prefill(slots[:2], ids[:2], pos[:2], vit[:10], None, None, None)
    assert len(calls) == 1 and calls[0]["rows"] == [40, 50]


def test_sampling_arguments_resolve_like_hf_generate():
    """Explicit arguments override generation_config.json, which overrides HF's defaults (do_sample False, temperature 1.0, top_k 50,
    top_p 1.0); `do_sample=None` (what video_qa / live_cc_once_for_evaluation pass, ref demo/infer.py:236-241, 297-302) lets the
    checkpoint decide; invalid warper parameters are rejected with HF's messages."""
    import pytest
    from livecc_amd.modeling import LiveCCForConditionalGeneration as M
    m = M.__new__(M)
    m.generation_config = {}
    assert m._resolve_sampling(None, {}) == dict(do_sample=False)
    assert m._resolve_sampling(False, {"top_k": 5}) == dict(do_sample=False)
    assert m._resolve_sampling(True, {}) == dict(do_sample=True, temperature=1.0, top_k=50, top_p=1.0, seed=None)
    m.generation_config = {"do_sample": True, "top_k": 1, "top_p": 0.001, "temperature": 0.01}        # the released checkpoints'
    assert m._resolve_sampling(None, {}) == dict(do_sample=True, temperature=0.01, top_k=1, top_p=0.001, seed=None)
    assert m._resolve_sampling(None, {"top_k": 20, "seed": 7})["top_k"] == 20 and m._resolve_sampling(None, {"seed": 7})["seed"] == 7
    assert m._resolve_sampling(False, {}) == dict(do_sample=False)
    assert m._resolve_sampling(True, {"top_k": None})["top_k"] == 0                                     # HF: top_k=None disables the warper
    with pytest.raises(ValueError):
        m._resolve_sampling(True, {"temperature": 0.0})
    with pytest.raises(ValueError):
        m._resolve_sampling(True, {"top_p": 1.5})


def test_threshold_processor_parameters_continue_from_count():
    """ref demo/infer.py:10-23: threshold = base + step * count; a reused processor object continues where it stopped, and anything
    that is not the reference's processor is refused instead of silently ignored."""
    import pytest
    from livecc_amd.infer import ThresholdLogitsProcessor
    from livecc_amd.modeling import LiveCCForConditionalGeneration as M
    assert M._threshold_params(None) is None and M._threshold_params([]) is None
    p = ThresholdLogitsProcessor(77, 0.25, 0.05)
    assert M._threshold_params([p]) == (77, 0.25, 0.05)
    p.count = 4
    tok, base, step = M._threshold_params([p])
    assert tok == 77 and abs(base - 0.45) < 1e-12 and step == 0.05
    with pytest.raises(NotImplementedError):
        M._threshold_params([p, p])
    with pytest.raises(NotImplementedError):
        M._threshold_params([object()])


def test_live_cc_uploads_only_the_due_frames_and_evicts_old_readers(infer, monkeypatch):
    """the decoded video stays where the decoder put it (host); per call only the selected frame indices are fetched
    through `get_batch` (the reference's decord call), and the reader cache is bounded."""
    fetched = []

    def spy_clip(video, h, w, ts, pts, index_from, layout="THWC", device=None):
        idxs, kept = R.select_clip_frames(ts, pts, index_from)
        assert hasattr(video, "get_batch"), "live_cc must hand the READER over, not a resident copy of all frames"
        if idxs:
            fetched.append(tuple(video.get_batch(idxs).shape))
        return (torch.zeros(len(idxs), 3, h, w, dtype=torch.uint8) if idxs else None), kept, idxs
    monkeypatch.setattr(R, "get_smart_resized_clip", spy_clip)
    n = 40
    vids = [V.DecodedVideo(torch.zeros(n, 56, 84, 3, dtype=torch.uint8), np.arange(n) / 4.0) for _ in range(infer.MAX_CACHED_READERS + 3)]
    for v in vids:
        state = {"video_path": v, "video_timestamp": 3.0}
        list(infer.live_cc(message="q", state=state, max_new_tokens=2))
        assert not v.frames.is_cuda and v.frames.shape[0] == n
        kv = state.get("past_key_values")
        if kv is not None:
            kv.release()
    assert fetched and all(s[0] <= 8 for s in fetched), fetched
    assert len(infer._cached_video_readers_with_hw) <= infer.MAX_CACHED_READERS
    infer.close_video(vids[-1])
    assert id(vids[-1]) not in infer._cached_video_readers_with_hw
