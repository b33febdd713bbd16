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


# ---- round 6: concurrent callers, end of the KV window, per-stream error isolation ---------------------
class _FixedIds:
    """Per-turn ids that depend on the turn index only (a shared builder is then independent of the order the streams call it in)."""

    def __new__(cls, cfg):
        from livecc_amd import protocol

        class B(protocol.TurnBuilder):
            def turn_ids(self, turn_index, n_video_tokens, with_query=None):
                self._rng = np.random.RandomState(1000 + turn_index)
                return super().turn_ids(turn_index, n_video_tokens, with_query)
        return B(cfg)


def _stream_turns(native, seed, n_turns=3, max_new_tokens=4):
    """One stream through `generate`, turn by turn over its carried KV (what a caller thread of ref demo/app.py does)."""
    from livecc_amd import protocol
    cfg = native.cfg
    frames = torch.from_numpy(protocol.synth_frames(6 + 2 * (n_turns - 1), 56, 84, seed=seed, layout="TCHW"))
    builder = protocol.TurnBuilder(cfg, seed=seed)
    state, past, toks = None, None, []
    for ti, (a, b) in enumerate(protocol.split_clip(frames.shape[0])):
        new = builder.turn_ids(ti, protocol.num_video_tokens(protocol.grid_of(b - a, 56, 84, cfg), cfg))
        ids = new if past is None else np.concatenate([past, new])
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b], past_key_values=state, repetition_penalty=1.05,
                            max_new_tokens=max_new_tokens, min_new_tokens=max_new_tokens)
        state = r.past_key_values
        seq = r.sequences[0].cpu().numpy()
        toks.append(seq[len(ids):].tolist())
        past = seq[:-1]
    state.release()
    return toks


def test_four_threads_on_one_model_give_the_tokens_of_the_serial_run(dev, native):
    """ref demo/app.py:178 runs `default_concurrency_limit=5` callers over ONE model object (and races on rope_deltas).  Here four threads,
    each with its own stream, interleave their turns on one engine (one workspace, one meta ring): the engine lock serialises whole
    generate calls, every stream keeps its own state, and every thread gets exactly the tokens of the serial run."""
    import threading
    serial = {s: _stream_turns(native, s) for s in (11, 12, 13, 14)}
    got, errs = {}, []

    def worker(seed):
        try:
            torch.cuda.set_device(dev)
            got[seed] = _stream_turns(native, seed)
        except Exception as e:          # noqa: BLE001
            errs.append((seed, repr(e)))
    for _ in range(2):
        got.clear()
        ths = [threading.Thread(target=worker, args=(s,)) for s in serial]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, errs
        assert got == serial
    assert len(native._free_slots) == native.engine.max_slots, "every stream slot came back"


def test_process_global_knobs_are_refused_while_a_call_is_in_flight(dev, native):
    """lcc_debug_set_* are process-global launch-routing state: while a host thread is inside a model-level call they answer LCC_ERR_STATE
    instead of changing the kernel family half-way through a forward pass."""
    import threading
    from livecc_amd import _lib
    lib = _lib.load()
    assert lib.lcc_debug_set_gemm_variant(2) == 0          # idle: accepted (2 = the default routing)
    stop, refused, done = threading.Event(), [], []

    def caller():
        torch.cuda.set_device(dev)
        while not stop.is_set():
            _stream_turns(native, 21, n_turns=2, max_new_tokens=16)
        done.append(1)
    th = threading.Thread(target=caller)
    th.start()
    try:
        import time
        t0 = time.time()
        while time.time() - t0 < 20.0 and len(refused) < 3:
            rc = lib.lcc_debug_set_gemm_variant(2)
            if rc != 0:
                refused.append((rc, lib.lcc_last_error().decode()))
    finally:
        stop.set()
        th.join()
    assert len(refused) >= 3 and all("in flight" in m for _, m in refused), refused[:2]
    assert lib.lcc_debug_set_gemm_variant(2) == 0          # idle again


def _small_window_model(dev):
    from livecc_amd.config import tiny
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    return LiveCCForConditionalGeneration.from_config(tiny(), dev, seed=0, max_streams=3, max_kv_len=192, max_new_rows=512,
                                                      max_patches=2048, max_history=8)


def test_a_stream_at_the_end_of_its_kv_window_is_ended_and_the_others_go_on(dev):
    """A 2-fps stream fills its KV slot sooner or later (ref README.md:77-79; the reference has no policy).  Before round 6 the engine
    refused the WHOLE batched prefill (engine_llm.hip: 'KV capacity exceeded') and `step` raised for every stream."""
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.server import StreamServer
    model = _small_window_model(dev)
    clean = StreamServer(LiveCCDemoInfer(model=model, turn_builder=_FixedIds(model.cfg)), max_new_tokens=3, force_length=True, prefetch=False)
    vs = _video(32, 150, dev)
    clean.add_stream("short", *vs, t_start=0.0, max_pixels=4 * 28 * 28)
    want_short = [(span, text) for _, span, text, _ in clean.run(realtime=False)]
    clean.remove_stream("short")
    srv = StreamServer(LiveCCDemoInfer(model=model, turn_builder=_FixedIds(model.cfg)), max_new_tokens=3, force_length=True, prefetch=False)
    srv.add_stream("long", *_video(31, 900, dev), t_start=0.0, max_pixels=4 * 28 * 28)       # 30 s of video: cannot fit 192 rows
    srv.add_stream("short", *vs, t_start=0.0, max_pixels=4 * 28 * 28)
    out = srv.run(realtime=False)
    long_res = [(span, text, st) for sid, span, text, st in out if sid == "long"]
    assert long_res[-1][2].get("window_full") and long_res[-1][2]["ended"] and srv.streams["long"].ended
    assert len(long_res) >= 3 and not any("error" in st for _, _, st in long_res)
    assert [(span, text) for sid, span, text, _ in out if sid == "short"] == want_short, "the other stream is served as if alone"
    assert len(model._free_slots) == model.engine.max_slots - 1, "the ended stream gave its KV slot back; only `short` still holds one"
    srv.remove_stream("long"); srv.remove_stream("short")
    # "restart": the slot is recycled, the chunk becomes the first turn of a fresh conversation, the stream is served to its end
    srv = StreamServer(LiveCCDemoInfer(model=model, turn_builder=_FixedIds(model.cfg)), max_new_tokens=3, force_length=True, prefetch=False,
                       window_policy="restart")
    srv.add_stream("long", *_video(31, 450, dev), t_start=0.0, max_pixels=4 * 28 * 28)       # 15 s
    out = srv.run(realtime=False)
    spans = [span for _, span, _, _ in out]
    assert spans[0] == (0.0, 3.0) and spans[-1][1] >= 14.0 and all(b - a == 1.0 for a, b in spans[1:])
    assert sum(1 for *_, st in out if st.get("restarted")) == srv.streams["long"].restarts >= 1
    assert not any(st.get("window_full") or "error" in st for *_, st in out)
    srv.remove_stream("long")


def test_an_error_in_one_stream_is_isolated_by_per_stream_retries(dev, native):
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.server import StreamServer
    vids = {"a": _video(41, 150, dev), "b": _video(42, 150, dev), "c": _video(43, 150, dev)}

    def serve(corrupt):
        srv = StreamServer(LiveCCDemoInfer(model=native, turn_builder=_FixedIds(native.cfg)), max_new_tokens=3, force_length=True, prefetch=False)
        for sid, (v, pts) in vids.items():
            srv.add_stream(sid, v, pts, t_start=0.0, max_pixels=4 * 28 * 28)
        out = list(srv.step(0.0))
        stale = None
        if corrupt:
            stale = srv.streams["b"].kv
            stale.released = True        # a stale handle: generate_batch refuses the whole batched call ("stream state was released")
        out += srv.run(realtime=False)
        if stale is not None:            # give the slot back for real (the flag above only faked the release)
            stale.released = False
            stale.release()
        for sid in list(srv.streams):
            srv.remove_stream(sid)
        return out
    clean, broken = serve(False), serve(True)
    for sid in ("a", "c"):
        assert [(s, t) for i, s, t, _ in broken if i == sid] == [(s, t) for i, s, t, _ in clean if i == sid], sid
    b = [(s, t, st) for i, s, t, st in broken if i == "b"]
    assert len(b) == 2 and "released" in b[1][2]["error"] and b[1][2]["ended"]
    assert len(native._free_slots) == native.engine.max_slots


def test_a_failed_generate_batch_gives_its_fresh_slots_back_at_once(dev, native):
    """A call that allocated stream slots for requests without a state and then failed must not keep them until the caller lets go of the
    exception (its traceback holds the call's frame): StreamServer's per-stream retries need the slots while they handle the error."""
    from livecc_amd import protocol
    cfg = native.cfg
    frames = torch.from_numpy(protocol.synth_frames(6, 56, 84, seed=3, layout="TCHW")).to(dev)
    ids = protocol.TurnBuilder(cfg, seed=3).turn_ids(0, protocol.num_video_tokens(protocol.grid_of(6, 56, 84, cfg), cfg))
    good = dict(input_ids=torch.from_numpy(ids), frames=frames, frames_layout="TCHW", state=None)
    bad = dict(input_ids=torch.from_numpy(ids), state=None)                      # <|video_pad|> ids without frames: refused
    free0 = len(native._free_slots)
    held = None
    try:
        native.generate_batch([good, good, bad], max_new_tokens=2, force_length=True)
    except ValueError as e:
        held = e                                                                  # keep the exception (and its traceback) alive
        assert len(native._free_slots) == free0, "the two slots allocated before the bad request are free again"
    assert held is not None and "no frames" in str(held)
    outs = native.generate_batch([good, good], max_new_tokens=2, force_length=True)       # and usable
    assert len(outs) == 2
    for o in outs:
        o.past_key_values.release()
    assert len(native._free_slots) == free0
