"""Multi-stream real-time loop: what replaces `LiveCCDemoInfer.live_cc`'s per-stream generator and its Gradio /
CLI callers (ref demo/infer.py:61-180, demo/app.py:108-146, demo/cli.py) when MANY 2-fps video streams share one GPU.

The reference serves one stream per blocking `generate` call; with concurrent callers it is not even safe (`rope_deltas` is
module state, HF modeling_qwen2_vl.py:857).  Here every stream keeps its own state (KV slot, rope_delta, past_ids, timestamps)
and one scheduler thread owns the device:

  * `step(now)`: for every stream, decide with the reference's rule which frames became due at its video time (`demo/infer.py`
    steps 1-2), take its OLDEST pending chunk (6 frames first, then 2), fetch + resize it on the GPU (step 3), build the turn
    ids (step 5) -- and run ALL the streams' chunks as ONE `generate_batch` call: one batched ViT, one packed prefill, decode
    steps that stream the weights once for the whole batch (continuous batching at chunk granularity);
  * pacing (the reference's): the first chunk (6 frames = 3 s) is taken at once, every later 2-frame chunk becomes due when
    the stream's video clock passes its first frame time; `run()` sleeps until the next due time (real-time mode) or jumps
    its clock there (offline replay);
  * back-pressure: a stream that falls more than `max_lag_s` behind either keeps catching up one chunk per step ("catch_up",
    the reference's behaviour: it processes every pending chunk) or skips the stale frames ("drop": the skipped interval is
    never shown to the model, the next chunk starts at the newest due frame pair);
  * stateless mode (`hf_spaces`, ref demo/infer.py:176-178): results carry a light state without KV / past_ids;
  * end of the KV window (round 6; the reference has no policy: ref demo/infer.py:61-180 grows `past_key_values` until HF runs out
    of trained positions, README.md:77-79 quotes 24k visual + 8k text): BEFORE a batch is formed, a stream whose next chunk would
    not fit its KV slot (cached + new rows + generation headroom > max_kv_len) is either ended with a result that carries
    `window_full=True` (`window_policy="end"`, default) or restarted from its system prompt with the current chunk as a first
    turn (`"restart"`: the KV slot is recycled, the result carries `restarted=True`) -- one full stream can no longer fail the
    batched call of every other stream;
  * error isolation: when the batched `generate_batch` raises, every stream's KV length is rolled back to what it was before
    the call and the streams are retried ONE BY ONE; a stream that fails alone is ended with `error=...` in its result state,
    the others are served as if the bad stream had not been in the batch;
  * one lock: a step holds the model's engine lock (modeling.LiveCCForConditionalGeneration._lock) from batch formation to the
    last token read, so `step` / `add_stream` / `remove_stream` may be called from several threads (ref demo/app.py:178 runs
    five concurrent callers over one model object).

Video decoding stays external: a stream is a GPU-resident uint8 frame tensor plus its pts (or any object with the same two
attributes that a decoder thread appends to).
"""
from __future__ import annotations

import contextlib
import dataclasses
import threading
import time
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import protocol
from . import resize as R
from .infer import LiveCCDemoInfer, ThresholdLogitsProcessor


@dataclasses.dataclass
class _Stream:
    sid: object
    frames: torch.Tensor                 # uint8 [N,H,W,3] (THWC) on the GPU
    pts: np.ndarray                      # seconds
    query: Optional[str]
    layout: str
    t_start: float                       # wall-clock time at which the stream's video time is 0
    resized_hw: Tuple[int, int]
    last_timestamp: float = -1.0 / protocol.FPS
    last_pts_index: int = -1
    turn_index: int = 0
    past_ids: Optional[np.ndarray] = None
    kv: object = None
    sent_query: Optional[str] = None
    ended: bool = False
    dropped_s: float = 0.0
    ahead: Optional[dict] = None         # the chunk after the current one, fetched + resized ahead (its ViT runs under this step's decode)
    window_full: bool = False            # ended because the next chunk would not fit the KV slot (window_policy="end")
    restarts: int = 0                    # times the stream was restarted from its system prompt (window_policy="restart")
    error: Optional[str] = None          # ended because its own generate call failed


class StreamServer:
    def __init__(self, infer: LiveCCDemoInfer, max_new_tokens: int = 16, repetition_penalty: float = 1.05,
                 streaming_eos_base_threshold: Optional[float] = None, streaming_eos_threshold_step: float = 0.0,
                 default_query: str = "Please describe the video.", max_lag_s: float = 4.0, lag_policy: str = "catch_up",
                 force_length: bool = False, prefetch: bool = True, window_policy: str = "end"):
        if lag_policy not in ("catch_up", "drop"):
            raise ValueError(lag_policy)
        if window_policy not in ("end", "restart"):
            raise ValueError(window_policy)
        self.window_policy = window_policy
        self._mu = threading.RLock()          # the stream table; the engine itself is guarded by the model's own lock
        self.infer, self.model, self.cfg = infer, infer.model, infer.cfg
        self.max_new_tokens, self.repetition_penalty = max_new_tokens, repetition_penalty
        self.thr = (streaming_eos_base_threshold, streaming_eos_threshold_step)
        self.default_query, self.max_lag_s, self.lag_policy, self.force_length = default_query, max_lag_s, lag_policy, force_length
        self.streams: Dict[object, _Stream] = {}
        # prefetch: when the frames of a stream's NEXT chunk are already in its buffer, they are fetched + resized now and their
        # vision tower runs on the model's side stream under this step's decode steps (modeling.generate_batch(prefetch=...))
        self.prefetch = prefetch

    # ---- stream management ----
    def _engine_lock(self):
        """The model's engine lock (re-entrant: generate_batch takes it again); stand-in models of the CPU tests have none."""
        return getattr(self.model, "_lock", None) or contextlib.nullcontext()

    def add_stream(self, sid, video_frames: torch.Tensor, video_pts, query: Optional[str] = None, t_start: float = 0.0,
                   max_pixels: int = 384 * 28 * 28, layout: str = "THWC") -> None:
        with self._mu:
            if len(self.streams) >= self.model.engine.max_slots:
                raise RuntimeError(f"all {self.model.engine.max_slots} stream slots of this GPU are in use")
            hw = video_frames.shape[1:3] if layout == "THWC" else video_frames.shape[2:4]
            rh, rw = R.smart_resized_hw(int(hw[0]), int(hw[1]), int(video_frames.shape[0]), max_pixels)
            self.streams[sid] = _Stream(sid, video_frames, np.asarray(video_pts, dtype=np.float64), query, layout, float(t_start), (rh, rw))

    def remove_stream(self, sid) -> None:
        with self._mu, self._engine_lock():
            st = self.streams.pop(sid)
            if st.kv is not None:
                st.kv.release()
                st.kv = None

    # ---- which chunk of a stream is due (ref demo/infer.py steps 1-2, one chunk at a time) ----
    def _next_chunk_timestamps(self, st: _Stream, now: float) -> Optional[List[float]]:
        fti = protocol.FRAME_TIME_INTERVAL
        if now < st.t_start:                              # the stream has not started yet
            return None
        video_time = min(now - st.t_start, float(st.pts[-1]))
        if st.last_timestamp + fti > st.pts[-1]:
            st.ended = True
            return None
        first = st.last_timestamp < 0
        n = protocol.INITIAL_FPS_FRAMES if first else protocol.STREAMING_FPS_FRAMES
        start = st.last_timestamp + fti
        if first:
            return [start + i * fti for i in range(n)]   # ref :107-110: the first call always takes the initial 3 s (6 frames)
        # ref :111-113: a chunk is due as soon as the video clock has passed its FIRST frame time; its second frame is fetched
        # with it (timestamps are padded to an even count, video_process_patch.py:134-135)
        if video_time <= start:
            if start >= float(st.pts[-1]):
                # The video clock is clamped to pts[-1], so a chunk whose first frame time is >= pts[-1] can never become due
                # (pts[-1] on an exact half second, e.g. 121 frames at 30 fps: pts[-1] = 4.0 = last_timestamp + 0.5).  The
                # reference has the same dead zone (ref demo/infer.py:96-111: neither `video_end` nor a chunk) and leaves it
                # to its polling UI; a scheduler loop must terminate, so the stream ends here.
                st.ended = True
            return None
        if self.lag_policy == "drop" and video_time - start > self.max_lag_s:
            skip = float(int((video_time - start) / (n * fti))) * n * fti      # skip stale pairs: restart at the newest due pair
            st.dropped_s += skip
            start += skip
        return [start + i * fti for i in range(n)]

    def due_time(self, sid) -> Optional[float]:
        """Wall-clock time at which the stream's next chunk becomes due (None when the stream has ended)."""
        st = self.streams[sid]
        if st.ended or st.last_timestamp + protocol.FRAME_TIME_INTERVAL >= st.pts[-1]:
            st.ended = True          # (>=: see the dead zone note in _next_chunk_timestamps)
            return None
        if st.last_timestamp < 0:
            return st.t_start
        return st.t_start + st.last_timestamp + protocol.FRAME_TIME_INTERVAL + 1e-6

    # ---- one scheduler step: at most one chunk per stream, all chunks in one batched generate ----
    def _turn_ids(self, st: _Stream, start: float, stop: float, grid) -> np.ndarray:
        if self.infer.text is not None:
            msg = st.query or self.default_query
            q = msg if st.sent_query != msg else None
            st.sent_query = msg
            return self.infer.text.turn_ids(start, stop, grid, q, continuing=st.past_ids is not None)
        return self.infer.turn_builder.turn_ids(st.turn_index, protocol.num_video_tokens(grid, self.cfg))

    def _fits_window(self, n_ids_total: int) -> bool:
        """lcc_llm_prefill's capacity rule (engine_llm.hip: cached + new + max_history <= max_kv_len) on the host, BEFORE the batch exists."""
        eng = self.model.engine
        cap, head = getattr(eng, "max_kv_len", None), getattr(eng, "max_history", 0)
        return cap is None or n_ids_total + head <= cap

    def _light_state(self, st: _Stream, hf_spaces: bool, **flags) -> dict:
        state = dict(last_timestamp=st.last_timestamp, turn_index=st.turn_index, dropped_s=st.dropped_s, restarts=st.restarts, **flags)
        if not hf_spaces:
            state.update(past_ids=st.past_ids, past_key_values=st.kv)
        return state

    def _end_stream(self, st: _Stream) -> None:
        st.ended, st.ahead = True, None
        if st.kv is not None:
            st.kv.release()
            st.kv = None

    @torch.inference_mode()
    def step(self, now: float, hf_spaces: bool = False):
        with self._mu, self._engine_lock():
            return self._step(now, hf_spaces)

    def _step(self, now: float, hf_spaces: bool):
        reqs, metas, results = [], [], []
        for st in list(self.streams.values()):
            if st.ended:
                continue
            ts = self._next_chunk_timestamps(st, now)
            if ts is None:
                continue
            if st.ahead is not None and st.ahead["ts"] == ts and st.ahead["index_from"] == st.last_pts_index + 1:
                clip, clip_ts, idxs = st.ahead["clip"], st.ahead["clip_ts"], st.ahead["idxs"]      # the clip whose ViT was prefetched
            else:
                clip, clip_ts, idxs = R.get_smart_resized_clip(st.frames, st.resized_hw[0], st.resized_hw[1], ts, st.pts, st.last_pts_index + 1,
                                                               st.layout)
            st.ahead = None
            if len(idxs) == 0:
                st.ended = True
                continue
            start, stop = clip_ts[0], clip_ts[len(idxs) - 1] + protocol.FRAME_TIME_INTERVAL
            grid = protocol.grid_of(clip.shape[0], clip.shape[2], clip.shape[3], self.cfg)
            # end of the KV window, decided here so that the batch never contains a stream the engine would refuse.  The turn ids are
            # only built once the chunk is known to run (the synthetic TurnBuilder is a seeded stream: an unused draw would shift it).
            restarted = False
            n_new_est = protocol.num_video_tokens(grid, self.cfg) + 64           # text of a turn is a few dozen ids (checked exactly below)
            if st.past_ids is not None and not self._fits_window(len(st.past_ids) + n_new_est):
                if self.window_policy == "end":
                    st.window_full = True
                    self._end_stream(st)
                    results.append((st.sid, (start, stop), "", self._light_state(st, hf_spaces, window_full=True, ended=True)))
                    continue
                # "restart": recycle the KV slot, the current chunk becomes the first turn of a fresh conversation (system prompt + query)
                if st.kv is not None:
                    st.kv.release()
                st.kv, st.past_ids, st.sent_query, st.turn_index = None, None, None, 0
                st.restarts += 1
                restarted = True
            new_ids = self._turn_ids(st, start, stop, grid)
            ids = new_ids if st.past_ids is None else np.concatenate([st.past_ids, new_ids])
            if not self._fits_window(len(ids)):          # a single turn larger than the whole slot (or the estimate above was short)
                st.window_full = True
                self._end_stream(st)
                results.append((st.sid, (start, stop), "", self._light_state(st, hf_spaces, window_full=True, ended=True)))
                continue
            reqs.append(dict(input_ids=torch.from_numpy(ids), frames=clip, frames_layout="TCHW", state=st.kv))
            metas.append((st, start, stop, len(ids), idxs[-1], clip_ts[-1], restarted))     # ref :117-118 keeps timestamps[-1]
        if not reqs:
            return results
        procs = None
        if self.thr[0] is not None and self.infer.streaming_eos_token_id is not None:
            procs = [ThresholdLogitsProcessor(self.infer.streaming_eos_token_id, self.thr[0], self.thr[1] or 0.0)]
        ahead = []
        if self.prefetch:
            fti = protocol.FRAME_TIME_INTERVAL
            for (st, start, stop, n_in, last_idx, last_ts, _) in metas:
                nts = [last_ts + fti + i * fti for i in range(protocol.STREAMING_FPS_FRAMES)]
                if nts[-1] > float(st.pts[-1]):
                    continue                                   # the next pair is not (completely) in the buffer yet
                clip, clip_ts, idxs = R.get_smart_resized_clip(st.frames, st.resized_hw[0], st.resized_hw[1], nts, st.pts, last_idx + 1, st.layout)
                if len(idxs) == protocol.STREAMING_FPS_FRAMES:
                    st.ahead = dict(ts=nts, index_from=last_idx + 1, clip=clip, clip_ts=clip_ts, idxs=idxs)
                    ahead.append(dict(frames=clip, frames_layout="TCHW"))
        gen_kw = dict(repetition_penalty=self.repetition_penalty, max_new_tokens=self.max_new_tokens, force_length=self.force_length)
        # KV lengths before the call: what a failed batched call is rolled back to before the per-stream retries
        eng = self.model.engine
        snap = {}
        if hasattr(eng, "slot_length"):
            for (st, *_rest) in metas:
                if st.kv is not None:
                    snap[id(st)] = eng.slot_length(st.kv.slot)
        try:
            outs = self.model.generate_batch(reqs, logits_processor=procs, **gen_kw, **({"prefetch": ahead} if ahead else {}))
        except Exception as exc:          # LccError / ValueError / RuntimeError: isolate the stream(s) that cause it
            if len(reqs) == 1:
                outs = [exc]
            else:
                outs = []
                for rq, (st, *_rest) in zip(reqs, metas):
                    if id(st) in snap:
                        eng.set_slot_length(st.kv.slot, *snap[id(st)])
                    p1 = None
                    if procs is not None:
                        p1 = [ThresholdLogitsProcessor(self.infer.streaming_eos_token_id, self.thr[0], self.thr[1] or 0.0)]
                    try:
                        outs.append(self.model.generate_batch([rq], logits_processor=p1, **gen_kw)[0])
                    except Exception as exc1:
                        outs.append(exc1)
        for (st, start, stop, n_in, last_idx, last_ts, restarted), o in zip(metas, outs):
            if isinstance(o, Exception):
                st.error = f"{type(o).__name__}: {o}"
                self._end_stream(st)
                results.append((st.sid, (start, stop), "", self._light_state(st, hf_spaces, error=st.error, ended=True)))
                continue
            seq = o.sequences[0].cpu().numpy()
            st.kv, st.past_ids = o.past_key_values, seq[:-1]            # ref demo/infer.py:173-174
            st.turn_index += 1
            st.last_pts_index, st.last_timestamp = last_idx, last_ts
            eos = set(getattr(self.model, "eos_token_ids", None) or (self.cfg.eos_token_id,))     # <|im_end|> and <|endoftext|>
            toks = [int(t) for t in seq[n_in:] if int(t) not in eos]
            flags = dict(restarted=True) if restarted else {}
            results.append((st.sid, (start, stop), self.infer.decode(toks), self._light_state(st, hf_spaces, **flags)))
        return results

    # ---- driver loops ----
    def run(self, until: Optional[float] = None, realtime: bool = True, clock: Callable[[], float] = time.monotonic,
            sleep: Callable[[float], None] = time.sleep, on_result: Optional[Callable] = None, t0: Optional[float] = None):
        """Serve until every stream has ended (or wall time `until`).  realtime=False replays back to back: the scheduler's
        clock jumps to the next due time instead of sleeping (offline evaluation / benchmark)."""
        out = []
        base = clock() if t0 is None else t0
        now = 0.0
        stalls = 0
        while True:
            now = (clock() - base) if realtime else now
            res = self.step(now)
            for r in res:
                out.append(r)
                if on_result:
                    on_result(*r)
            dues = [d for d in (self.due_time(s) for s in list(self.streams)) if d is not None]
            if not dues or (until is not None and now >= until):
                break
            nxt = min(dues)
            if not res:
                if nxt > now:
                    if realtime:
                        sleep(min(nxt - now, 0.25))
                    else:
                        now = nxt
                else:
                    # Nothing ran although a stream claims to be due (it cannot happen with the end-of-video rules above; kept
                    # as a guard): never spin -- wait a tick in real time, step the offline clock, and after a few fruitless
                    # rounds end the streams that claim to be due.
                    stalls += 1
                    if stalls > 8:
                        for sid in list(self.streams):
                            d = self.due_time(sid)
                            if d is not None and d <= now:
                                self.streams[sid].ended = True
                        stalls = 0
                    elif realtime:
                        sleep(0.01)
                    else:
                        now += protocol.FRAME_TIME_INTERVAL
            else:
                stalls = 0
        return out
