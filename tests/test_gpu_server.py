"""-m gpu: the multi-stream scheduler -- continuous batching of the streams' due chunks with the reference's
pacing rule; every stream must see exactly the turns, and produce exactly the tokens, that it would alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _video(seed, n_frames, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n_frames, 60, 90, 3), dtype=torch.uint8, generator=g).to(dev), np.arange(n_frames) / 30.0


@pytest.fixture(scope="module")
def native(dev):
    from livecc_amd.config import tiny
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    return LiveCCForConditionalGeneration.from_config(tiny(), dev, seed=0, max_streams=4, max_kv_len=2048, max_new_rows=1024,
                                                      max_patches=4096, max_history=16)


def test_batched_streams_equal_each_stream_alone(dev, native):
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.server import StreamServer
    infer = LiveCCDemoInfer(model=native)
    vids = {"a": _video(1, 180, dev), "b": _video(2, 150, dev), "c": _video(3, 120, dev)}          # 6 s, 5 s, 4 s at 30 fps
    starts = {"a": 0.0, "b": 0.7, "c": 2.2}
    # each stream alone through the single-stream path (fresh turn builders with the same seed => same synthetic text ids)
    alone = {}
    for sid, (v, pts) in vids.items():
        inf1 = LiveCCDemoInfer(model=native)
        from livecc_amd.video import DecodedVideo
        state, outs = {"video_path": DecodedVideo(v, pts, 30.0)}, []
        for wall in np.arange(0.0, 9.0, 0.25):
            state["video_timestamp"] = float(wall)
            for (a, b), text, state in inf1.live_cc(None, state, max_pixels=4 * 28 * 28, max_new_tokens=4, force_length=True, do_sample=False):
                outs.append(((a, b), text))
        state["past_key_values"].release()
        alone[sid] = outs
    srv = StreamServer(infer, max_new_tokens=4, force_length=True)
    for sid, (v, pts) in vids.items():
        srv.streams.pop(sid, None)
    # every stream gets its OWN synthetic-text generator in the single-stream runs; give the server the same by running the
    # streams through per-stream LiveCCDemoInfer-equivalent builders: the TurnBuilder is stateful (seeded RNG), so use one
    # server per comparison stream set with fresh builders
    results = {}
    for sid, (v, pts) in vids.items():
        s1 = StreamServer(LiveCCDemoInfer(model=native), max_new_tokens=4, force_length=True)
        s1.add_stream(sid, v, pts, t_start=0.0, max_pixels=4 * 28 * 28)
        results[sid] = [(span, text) for (_, span, text, _) in s1.run(realtime=False)]
        s1.remove_stream(sid)
        assert results[sid] == alone[sid], f"stream {sid}: server {results[sid]} vs alone {alone[sid]}"
    # spans: first chunk 0-3 s, then 1-s chunks until the video ends
    assert [r[0] for r in results["c"]] == [(0.0, 3.0), (3.0, 4.0)]
    assert [r[0] for r in results["a"]][:3] == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0)]


def test_streams_with_different_clocks_are_batched_and_paced(dev, native):
    from livecc_amd import protocol
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.server import StreamServer

    class FixedIds(protocol.TurnBuilder):          # deterministic per-turn ids independent of call order (shared builder)
        def turn_ids(self, turn_index, n_video_tokens, with_query=None):
            self._rng = np.random.RandomState(1000 + turn_index)
            return super().turn_ids(turn_index, n_video_tokens, with_query)

    infer = LiveCCDemoInfer(model=native, turn_builder=FixedIds(native.cfg))
    srv = StreamServer(infer, max_new_tokens=3, force_length=True)
    va, vb = _video(7, 150, dev), _video(8, 150, dev)
    srv.add_stream("a", *va, t_start=0.0, max_pixels=4 * 28 * 28)
    srv.add_stream("b", *vb, t_start=1.0, max_pixels=4 * 28 * 28)
    log = []
    # offline clock: step at the due times
    now = 0.0
    for _ in range(40):
        res = srv.step(now)
        log += [(round(now, 3), sid, span) for (sid, span, _, _) in res]
        dues = [d for d in (srv.due_time(s) for s in ("a", "b")) if d is not None]
        if not dues:
            break
        now = max(now, min(dues)) if not res else now
    a = [(t, span) for (t, sid, span) in log if sid == "a"]
    b = [(t, span) for (t, sid, span) in log if sid == "b"]
    assert a[0] == (0.0, (0.0, 3.0)) and b[0][1] == (0.0, 3.0) and b[0][0] >= 1.0          # b's clock starts 1 s later
    assert [s for _, s in a] == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0)] and [s for _, s in b] == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0)]
    # a chunk is never served before the stream's video clock has passed its first frame
    for t, (s0, _) in a[1:]:
        assert t - 0.0 > s0 - 1e-9
    for t, (s0, _) in b[1:]:
        assert t - 1.0 > s0 - 1e-9
    # both streams' 3.0-4.0 chunks could only be batched if due at the same step; with the 1-s offset they are not
    assert all(srv.streams[s].ended or srv.due_time(s) is None for s in ("a", "b"))
    # same content (the ids are per-turn deterministic): the offset stream produces the same tokens as an un-offset copy
    srv2 = StreamServer(LiveCCDemoInfer(model=native, turn_builder=FixedIds(native.cfg)), max_new_tokens=3, force_length=True)
    srv2.add_stream("b0", *vb, t_start=0.0, max_pixels=4 * 28 * 28)
    r2 = srv2.run(realtime=False)
    texts_b = [text for (sid, span, text, _) in srv.run(realtime=False)]          # nothing left to serve
    assert texts_b == []
    srv.remove_stream("a"); srv.remove_stream("b"); srv2.remove_stream("b0")
    assert len(r2) == 3


def test_drop_policy_skips_stale_frames(dev, native):
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.server import StreamServer
    srv = StreamServer(LiveCCDemoInfer(model=native), max_new_tokens=2, force_length=True, lag_policy="drop", max_lag_s=2.0)
    v, pts = _video(9, 600, dev)                      # 20 s
    srv.add_stream("x", v, pts, t_start=0.0, max_pixels=4 * 28 * 28)
    r0 = srv.step(0.0)
    assert r0[0][1] == (0.0, 3.0)
    r1 = srv.step(10.3)                               # the consumer stalled for 10 s: 7.3 s behind -> skip to the newest due pair
    (sid, span, text, state) = r1[0]
    assert span == (10.0, 11.0) and state["dropped_s"] == 7.0
    r2 = srv.step(10.4)
    assert r2 == []                                   # 11.0 not yet reached
    srv.remove_stream("x")
