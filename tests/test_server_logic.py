"""Scheduler logic of the multi-stream loop (livecc_amd/server.py) on the CPU with a stand-in model: the reference's pacing rule
(first 6-frame chunk at once, then one 2-frame chunk whenever the stream's video clock passes the chunk's first frame), one
batched generate per step, catch-up vs drop back-pressure, end of stream."""
import numpy as np
import pytest
import torch

from livecc_amd import protocol, server
from livecc_amd.config import tiny


class _KV:
    def release(self):
        pass


class FakeModel:
    def __init__(self, cfg):
        self.cfg = cfg
        self.engine = type("E", (), {"max_slots": 4})()
        self.batches = []

    def generate_batch(self, reqs, **kw):
        self.batches.append(len(reqs))
        self.prefetched = getattr(self, "prefetched", []) + [len(kw.get("prefetch") or [])]
        self.req_clips = getattr(self, "req_clips", []) + [[id(r["frames"]) for r in reqs]]
        self.pf_clips = getattr(self, "pf_clips", []) + [[id(c["frames"]) for c in (kw.get("prefetch") or [])]]
        outs = []
        for r in reqs:
            seq = torch.cat([r["input_ids"].view(-1), torch.tensor([7, 8, 9])]).view(1, -1)
            outs.append(type("O", (), {"sequences": seq, "past_key_values": r["state"] or _KV()})())
        return outs


class FakeInfer:
    def __init__(self):
        self.cfg = tiny()
        self.model = FakeModel(self.cfg)
        self.text = None
        self.turn_builder = protocol.TurnBuilder(self.cfg)
        self.streaming_eos_token_id = None
        self.decode = lambda ids: " ".join(str(i) for i in ids)


@pytest.fixture()
def srv(monkeypatch):
    def fake_clip(frames, h, w, ts, pts, index_from, layout="THWC"):
        from livecc_amd import resize as R
        idxs, kept = R.select_clip_frames(ts, pts, index_from)
        return (torch.zeros(len(idxs), 3, h, w, dtype=torch.uint8) if idxs else None), kept, idxs
    monkeypatch.setattr(server.R, "get_smart_resized_clip", fake_clip)
    return server.StreamServer(FakeInfer(), max_new_tokens=3)


def _video(n):
    return torch.zeros(n, 60, 90, 3, dtype=torch.uint8), np.arange(n) / 30.0


def test_pacing_batching_and_end(srv):
    srv.add_stream("a", *_video(180), t_start=0.0, max_pixels=4 * 28 * 28)      # 6 s
    srv.add_stream("b", *_video(180), t_start=0.0, max_pixels=4 * 28 * 28)
    srv.add_stream("c", *_video(120), t_start=2.0, max_pixels=4 * 28 * 28)      # starts later, 4 s long
    r = srv.step(0.0)
    assert sorted(x[0] for x in r) == ["a", "b"] and all(x[1] == (0.0, 3.0) for x in r) and srv.model.batches == [2]
    r = srv.step(2.9)                                                    # c started at wall 2.0: its first chunk is taken at once
    assert [x[0] for x in r] == ["c"] and r[0][1] == (0.0, 3.0)
    # at wall 3.2: a and b have passed video time 3.0 -> their (3.0, 4.0) chunks run in ONE batch
    r = srv.step(3.2)
    assert sorted(x[0] for x in r) == ["a", "b"] and all(x[1] == (3.0, 4.0) for x in r) and srv.model.batches[-1] == 2
    assert srv.step(3.3) == []
    assert srv.due_time("a") == pytest.approx(4.0, abs=1e-3) and srv.due_time("c") == pytest.approx(2.0 + 3.0, abs=1e-3)
    out = srv.run(realtime=False)
    spans = {}
    for sid, span, text, state in out:
        spans.setdefault(sid, []).append(span)
    assert spans["a"] == [(4.0, 5.0), (5.0, 6.0)] and spans["c"] == [(3.0, 4.0)]
    assert all(st.ended for st in srv.streams.values())
    assert "7 8" in out[0][2]                                           # decoded text of the generated ids (minus the dropped last)


def test_catch_up_and_drop(srv):
    srv.add_stream("x", *_video(600), t_start=0.0, max_pixels=4 * 28 * 28)      # 20 s
    assert srv.step(0.0)[0][1] == (0.0, 3.0)
    # a stalled consumer under the reference's policy: one pending chunk per step until it has caught up
    got = [srv.step(8.2)[0][1] for _ in range(3)]
    assert got == [(3.0, 4.0), (4.0, 5.0), (5.0, 6.0)]
    drop = server.StreamServer(FakeInfer(), max_new_tokens=3, lag_policy="drop", max_lag_s=2.0)
    drop.add_stream("y", *_video(600), t_start=0.0, max_pixels=4 * 28 * 28)
    assert drop.step(0.0)[0][1] == (0.0, 3.0)
    sid, span, _, st = drop.step(10.3)[0]
    assert span == (10.0, 11.0) and st["dropped_s"] == 7.0
    with pytest.raises(ValueError):
        server.StreamServer(FakeInfer(), lag_policy="nope")
    full = server.StreamServer(FakeInfer())
    for i in range(4):
        full.add_stream(i, *_video(60))
    with pytest.raises(RuntimeError):
        full.add_stream(9, *_video(60))


@pytest.mark.parametrize("n_frames", [121, 91, 106, 151])
@pytest.mark.timeout(20)
def test_run_terminates_when_the_last_pts_sits_on_a_half_second(srv, n_frames):
    """pts[-1] == last_timestamp + 0.5 exactly (121 frames at 30 fps -> 4.0 s) used to spin forever: the video clock is
    clamped to pts[-1], the end test was a strict `>`, and the loop neither slept nor advanced."""
    srv.add_stream("v", *_video(n_frames), t_start=0.0, max_pixels=4 * 28 * 28)
    out = srv.run(realtime=False)
    assert srv.streams["v"].ended
    spans = [span for _, span, _, _ in out]
    assert spans[0] == (0.0, 3.0) and all(b - a == 1.0 for a, b in spans[1:])
    # every full 2-frame pair whose frames exist was served, nothing beyond the video
    assert spans[-1][1] <= (n_frames - 1) / 30.0 + 1.0


@pytest.mark.timeout(20)
def test_run_realtime_never_spins(srv):
    srv.add_stream("v", *_video(121), t_start=0.0, max_pixels=4 * 28 * 28)
    t = [0.0]
    sleeps = []

    def sleep(dt):
        sleeps.append(dt)
        t[0] += dt
    out = srv.run(realtime=True, clock=lambda: t[0], sleep=sleep, t0=0.0)
    assert srv.streams["v"].ended and len(out) >= 1 and len(sleeps) < 200


def test_next_chunk_is_fetched_ahead_and_reused(srv):
    """The chunk after the current one is fetched + resized while the current one is generated (its ViT is handed to
    generate_batch(prefetch=...)); the next step passes the SAME clip object, so the model finds the prefetched embeddings."""
    srv.add_stream("a", *_video(240), t_start=0.0, max_pixels=4 * 28 * 28)      # 8 s
    out = srv.run(realtime=False)
    m = srv.model
    assert [span for _, span, _, _ in out] == [(0.0, 3.0)] + [(3.0 + i, 4.0 + i) for i in range(5)]
    assert m.prefetched[:-1] == [1] * (len(m.prefetched) - 1) and m.prefetched[-1] == 0      # nothing to fetch ahead of the last chunk
    for k in range(1, len(m.req_clips)):
        assert m.req_clips[k] == m.pf_clips[k - 1], "the prefetched clip object is the one generated next"
    quiet = server.StreamServer(FakeInfer(), max_new_tokens=3, prefetch=False)
    quiet.add_stream("a", *_video(240), t_start=0.0, max_pixels=4 * 28 * 28)
    assert [span for _, span, _, _ in quiet.run(realtime=False)] == [span for _, span, _, _ in out]
    assert set(quiet.model.prefetched) == {0}


# ---- round 6: end of the KV window, per-stream error isolation, locking -------------------------------
class _Slot:
    def __init__(self, slot):
        self.slot, self.released = slot, False

    def release(self):
        self.released = True


class CapModel(FakeModel):
    """Stand-in with the engine's capacity rule (cached + new + max_history <= max_kv_len) and a switchable failure."""

    def __init__(self, cfg, max_kv_len, max_history=3):
        super().__init__(cfg)
        self.engine = type("E", (), {"max_slots": 4, "max_kv_len": max_kv_len, "max_history": max_history})()
        self.lengths, self.rolled_back, self.fail_ids, self.next_slot = {}, [], set(), 0
        self.engine.slot_length = lambda slot: (self.lengths.get(slot, 0), self.lengths.get(slot, 0))
        self.engine.set_slot_length = lambda slot, kv, pos: (self.rolled_back.append((slot, kv)), self.lengths.__setitem__(slot, kv))

    def generate_batch(self, reqs, **kw):
        self.batches.append(len(reqs))
        for r in reqs:          # like lcc_llm_prefill: the WHOLE batched call fails on one bad stream
            n = int(r["input_ids"].numel())
            if n + self.engine.max_history > self.engine.max_kv_len:
                raise RuntimeError(f"KV capacity {self.engine.max_kv_len} exceeded")
            if r["state"] is not None and r["state"].slot in self.fail_ids:
                # a half-done call: pretend the prefill of every stream of the batch already advanced its KV
                for q in reqs:
                    if q["state"] is not None:
                        self.lengths[q["state"].slot] = int(q["input_ids"].numel())
                raise RuntimeError("injected failure")
        outs = []
        for r in reqs:
            st = r["state"]
            if st is None:
                st = _Slot(self.next_slot)
                self.next_slot += 1
            seq = torch.cat([r["input_ids"].view(-1), torch.tensor([7, 8, 9])]).view(1, -1)
            self.lengths[st.slot] = seq.numel() - 1
            outs.append(type("O", (), {"sequences": seq, "past_key_values": st})())
        return outs


def _cap_server(monkeypatch, max_kv_len, **kw):
    def fake_clip(frames, h, w, ts, pts, index_from, layout="THWC"):
        from livecc_amd import resize as R
        idxs, kept = R.select_clip_frames(ts, pts, index_from)
        return (torch.zeros(len(idxs), 3, h, w, dtype=torch.uint8) if idxs else None), kept, idxs
    monkeypatch.setattr(server.R, "get_smart_resized_clip", fake_clip)
    inf = FakeInfer()
    inf.model = CapModel(inf.cfg, max_kv_len)
    return server.StreamServer(inf, max_new_tokens=3, **kw)


def test_a_full_kv_window_ends_that_stream_only(monkeypatch):
    srv = _cap_server(monkeypatch, max_kv_len=160)
    srv.add_stream("long", *_video(600), t_start=0.0, max_pixels=4 * 28 * 28)     # 20 s: outgrows 160 rows
    srv.add_stream("short", *_video(150), t_start=0.0, max_pixels=4 * 28 * 28)    # 5 s
    out = srv.run(realtime=False)
    by = {}
    for sid, span, text, state in out:
        by.setdefault(sid, []).append((span, text, state))
    assert by["long"][-1][2].get("window_full") is True and by["long"][-1][2]["ended"] and by["long"][-1][1] == ""
    assert srv.streams["long"].window_full and srv.streams["long"].ended and srv.streams["long"].kv is None
    assert [s for s, _, _ in by["short"]] == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0)]          # served to its end, untouched
    assert not any(st.get("window_full") for _, _, st in by["short"])
    assert not any("error" in st for _, _, st in by["long"] + by["short"]), "the engine was never handed a stream it would refuse"
    assert len(by["long"]) >= 3 and all(t != "" for _, t, _ in by["long"][:-1])


def test_restart_policy_recycles_the_slot_and_continues(monkeypatch):
    srv = _cap_server(monkeypatch, max_kv_len=160, window_policy="restart")
    srv.add_stream("v", *_video(600), t_start=0.0, max_pixels=4 * 28 * 28)
    out = srv.run(realtime=False)
    spans = [span for _, span, _, _ in out]
    assert spans[0] == (0.0, 3.0) and spans[-1][1] >= 19.0, "every chunk of the 20-s video was served"
    assert all(b - a == 1.0 for a, b in spans[1:])
    flagged = [st for _, _, _, st in out if st.get("restarted")]
    assert len(flagged) >= 2 and srv.streams["v"].restarts == len(flagged)
    assert not any(st.get("window_full") or "error" in st for _, _, _, st in out)
    with pytest.raises(ValueError):
        server.StreamServer(FakeInfer(), window_policy="nope")


def test_one_failing_stream_does_not_fail_the_batch(monkeypatch):
    srv = _cap_server(monkeypatch, max_kv_len=100000)
    for sid in ("a", "b", "c"):
        srv.add_stream(sid, *_video(180), t_start=0.0, max_pixels=4 * 28 * 28)
    assert len(srv.step(0.0)) == 3
    m = srv.model
    bad = srv.streams["b"].kv.slot
    before = {sid: m.lengths[srv.streams[sid].kv.slot] for sid in ("a", "c")}
    m.fail_ids.add(bad)
    n_batches = len(m.batches)
    res = {sid: (span, text, st) for sid, span, text, st in srv.step(3.2)}
    assert m.batches[n_batches:] == [3, 1, 1, 1], "one batched attempt, then one call per stream"
    assert "injected failure" in res["b"][2]["error"] and res["b"][2]["ended"] and srv.streams["b"].ended and srv.streams["b"].kv is None
    for sid in ("a", "c"):
        assert res[sid][0] == (3.0, 4.0) and res[sid][1] != "" and "error" not in res[sid][2]
        assert (srv.streams[sid].kv.slot, before[sid]) in m.rolled_back, "the half-done batched call was rolled back before the retry"
    later = srv.run(realtime=False)
    assert sorted({sid for sid, *_ in later}) == ["a", "c"]


def test_server_steps_hold_the_engine_lock(monkeypatch):
    import threading
    srv = _cap_server(monkeypatch, max_kv_len=100000)
    held = []

    class Probe:
        def __enter__(self):
            held.append("in")

        def __exit__(self, *a):
            held.append("out")
    srv.model._lock = Probe()
    srv.add_stream("a", *_video(120), t_start=0.0, max_pixels=4 * 28 * 28)
    srv.step(0.0)
    assert held == ["in", "out"]
    srv.model._lock = threading.RLock()
    ths = [threading.Thread(target=srv.step, args=(3.2,)) for _ in range(4)]     # concurrent steps: the chunk is served exactly once
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert srv.streams["a"].turn_index == 2
