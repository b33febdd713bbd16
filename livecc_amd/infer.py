"""Stream orchestrator with the surface of the reference's `LiveCCDemoInfer` (ref demo/infer.py:25-310), driving the
native model instead of HF.  Video decode / resize (decord, torchvision) are outside this round's scope:
clips arrive as uint8 frame tensors already at the model resolution (what `get_smart_resized_clip`,
ref livecc_utils/video_process_patch.py:126-156, returns).  Text: with tokenizer files (a checkpoint directory, or a
`text.TextFrontEnd` passed in) the turns are built from real strings exactly as the reference does -- chat template,
'Time=a-bs' prefix, query appended on the first turn or when it changes, '<|im_end|>\n' glue, `processor.decode` of the
answer (ref demo/infer.py:134-157, 175); without them the turn ids are synthetic (`protocol.TurnBuilder`, benchmarks).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch

from . import protocol
from .modeling import LiveCCForConditionalGeneration


class ThresholdLogitsProcessor:
    """ref demo/infer.py:10-23.  Executed natively inside the fused sampler kernel; this object only carries the
    parameters (and the `count` the reference keeps)."""

    def __init__(self, token_id: int, base_threshold: float, step: float):
        self.token_id, self.base_threshold, self.step, self.count = token_id, base_threshold, step, 0


class LiveCCDemoInfer:
    fps = protocol.FPS
    initial_fps_frames = protocol.INITIAL_FPS_FRAMES
    streaming_fps_frames = protocol.STREAMING_FPS_FRAMES
    initial_time_interval = protocol.INITIAL_TIME_INTERVAL
    streaming_time_interval = protocol.STREAMING_TIME_INTERVAL
    frame_time_interval = protocol.FRAME_TIME_INTERVAL

    def __init__(self, model: LiveCCForConditionalGeneration = None, model_path: str = None, device: str = None,
                 turn_builder: Optional[protocol.TurnBuilder] = None, decode: Optional[Callable[[List[int]], str]] = None,
                 streaming_eos_token_id: Optional[int] = None, text=None):
        """`text`: a `text.TextFrontEnd` (real tokenizer).  With `model_path` it is created from the checkpoint directory's
        tokenizer files when they exist (ref demo/infer.py:48-58)."""
        if model is None:
            device = device or "cuda"
            model = LiveCCForConditionalGeneration.from_pretrained(model_path, torch_dtype="auto", device_map=device)
        self.model = model
        self.cfg = model.cfg
        if text is None and model_path is not None:
            import os
            if any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.json", "vocab.json", "tokenizer_config.json")):
                from .text import TextFrontEnd
                text = TextFrontEnd(model_path, self.cfg)
        self.text = text
        self.turn_builder = turn_builder or protocol.TurnBuilder(self.cfg)
        if text is not None:
            self.decode = decode or (lambda ids: text.decode(ids, skip_special_tokens=True))
            self.streaming_eos_token_id = text.streaming_eos_token_id if streaming_eos_token_id is None else streaming_eos_token_id
        else:
            self.decode = decode or (lambda ids: " ".join(str(i) for i in ids))
            # ref infer.py:49: tokenizer(' ...').input_ids[-1]; without tokenizer files the caller supplies it
            self.streaming_eos_token_id = streaming_eos_token_id

    @torch.inference_mode()
    def live_cc(self, clip: torch.Tensor, state: dict, frames_layout: str = "TCHW", do_sample: bool = False,
                repetition_penalty: float = 1.05, streaming_eos_base_threshold: float = None,
                streaming_eos_threshold_step: float = None, max_new_tokens: int = 16, force_length: bool = False,
                message: Optional[str] = None, default_query: str = "Please describe the video.", hf_spaces: bool = False):
        """One call = the frames that became due since the last call (ref infer.py:61-180, steps 4-5).
        `clip`: uint8 frames [T,3,H,W] (or THWC).  Yields ((start, stop), text, state) per chunk.
        `message` / `default_query` (real-tokenizer mode): the user query, appended to the turn when it is new or changed
        (ref infer.py:141-146)."""
        initialized = state.get("last_timestamp", -1.0) >= 0
        t0 = state.get("last_timestamp", -self.frame_time_interval) + self.frame_time_interval
        for a, b in protocol.split_clip(clip.shape[0], initialized):
            frames = clip[a:b]
            start = t0 + a * self.frame_time_interval
            stop = t0 + b * self.frame_time_interval
            turn = state.get("turn_index", 0)
            grid = protocol.grid_of(frames.shape[0], *(frames.shape[2:] if frames_layout == "TCHW" else frames.shape[1:3]), self.cfg)
            past_ids = state.get("past_ids")
            if self.text is not None:
                if not message and not state.get("message"):
                    message = default_query                       # ref infer.py:141-143
                query = None
                if message and state.get("message") != message:   # ref infer.py:144-146
                    query = message
                    state["message"] = message
                new_ids = self.text.turn_ids(start, stop, grid, query, continuing=past_ids is not None)
            else:
                new_ids = self.turn_builder.turn_ids(turn, protocol.num_video_tokens(grid, self.cfg))
            ids = new_ids if past_ids is None else np.concatenate([past_ids, new_ids])
            procs = None
            if streaming_eos_base_threshold is not None and self.streaming_eos_token_id is not None:
                procs = [ThresholdLogitsProcessor(self.streaming_eos_token_id, streaming_eos_base_threshold,
                                                  streaming_eos_threshold_step or 0.0)]
            out = self.model.generate(
                input_ids=torch.from_numpy(ids).view(1, -1), frames=frames, frames_layout=frames_layout,
                past_key_values=state.get("past_key_values"), return_dict_in_generate=True, do_sample=do_sample,
                repetition_penalty=repetition_penalty, logits_processor=procs, max_new_tokens=max_new_tokens,
                min_new_tokens=max_new_tokens if force_length else None, pad_token_id=self.cfg.eos_token_id)
            seq = out.sequences[0].cpu().numpy()
            state["past_key_values"] = out.past_key_values
            state["past_ids"] = seq[:-1]                      # ref infer.py:174
            state["turn_index"] = turn + 1
            state["last_timestamp"] = stop - self.frame_time_interval
            new_tokens = seq[len(ids):].tolist()
            text_out = self.decode([t for t in new_tokens if t != self.cfg.eos_token_id])
            if hf_spaces:   # ref infer.py:176-178: the caller gets a light copy without the device-side state
                yield (start, stop), text_out, {k: v for k, v in state.items() if k not in ("past_ids", "past_key_values")}
            else:
                yield (start, stop), text_out, state

    @torch.inference_mode()
    def live_cc_from_video(self, video_frames: torch.Tensor, video_pts, state: dict, video_timestamp: float,
                           max_pixels: int = 384 * 28 * 28, frames_layout: str = "THWC", **kw):
        """The whole `live_cc` of the reference (ref demo/infer.py:61-180) on a decoded video that is resident on the GPU
        (`video_frames` uint8 [N,H,W,3], `video_pts` seconds per frame; decoding itself stays external): decide which frames
        became due at wall-clock `video_timestamp` (steps 1-2), fetch and resize them on the GPU (step 3,
        `resize.get_smart_resized_clip`), then chunk + generate (steps 4-5, `live_cc`).  `state` carries `last_timestamp`,
        `last_video_pts_index`, `resized_hw`, `video_end` as the reference's does.  Yields ((start, stop), text, state)."""
        from . import resize as R
        pts = np.asarray(video_pts, dtype=np.float64)
        last_timestamp = state.get("last_timestamp", -1 / self.fps)
        if "resized_hw" not in state:                                  # get_smart_resized_video_reader, once per video
            hw = video_frames.shape[1:3] if frames_layout == "THWC" else video_frames.shape[2:4]
            state["resized_hw"] = R.smart_resized_hw(int(hw[0]), int(hw[1]), int(video_frames.shape[0]), max_pixels)
            state["last_video_pts_index"] = -1
        video_timestamp = min(float(video_timestamp), float(pts[-1]))
        if last_timestamp + self.frame_time_interval > pts[-1]:
            state["video_end"] = True
            return
        initialized = last_timestamp >= 0
        if not initialized:
            video_timestamp = max(video_timestamp, self.initial_time_interval)
        if video_timestamp <= last_timestamp + self.frame_time_interval:
            return
        timestamps = torch.arange(last_timestamp + self.frame_time_interval, video_timestamp, self.frame_time_interval).tolist()
        h, w = state["resized_hw"]
        clip, clip_ts, idxs = R.get_smart_resized_clip(video_frames, h, w, timestamps, pts, state["last_video_pts_index"] + 1, frames_layout)
        if len(idxs) == 0:
            return
        state["last_video_pts_index"] = idxs[-1]
        for out in self.live_cc(clip, state, frames_layout="TCHW", **kw):
            yield out
        state["last_timestamp"] = clip_ts[-1]                          # ref infer.py:118 (set from the fetched timestamps)

    @torch.inference_mode()
    def video_qa(self, query_len, state: dict, clip: Optional[torch.Tensor] = None, frames_layout: str = "TCHW",
                 repetition_penalty: float = 1.05, max_new_tokens: int = 512, force_length: bool = False):
        """Multi-turn QA with KV reuse (ref demo/infer.py:182-242): the first turn prefills the WHOLE clip in one shot
        (up to 480 frames / 24k visual tokens), later turns are text only.  `query_len`: the query STRING in real-tokenizer
        mode (the reference's `message`), or the number of synthetic query ids.  Returns (generated ids, state) -- with a
        tokenizer, `self.decode(ids)` is the reference's `response`."""
        turn = state.get("turn_index", 0)
        past_ids = state.get("past_ids")
        n_vid = 0
        grid = None
        if past_ids is None and clip is not None:       # "only use once" (infer.py:213-214)
            grid = protocol.grid_of(clip.shape[0], *(clip.shape[2:] if frames_layout == "TCHW" else clip.shape[1:3]), self.cfg)
            n_vid = protocol.num_video_tokens(grid, self.cfg)
        if isinstance(query_len, str):
            if self.text is None:
                raise ValueError("a query string needs tokenizer files (pass text=TextFrontEnd(...) or a checkpoint directory)")
            new_ids = self.text.encode(self.text.qa_text(query_len, continuing=past_ids is not None, with_video=grid is not None),
                                       [grid] if grid is not None else [])
        else:
            self.turn_builder.query_len = query_len
            new_ids = self.turn_builder.turn_ids(turn, n_vid, with_query=True)
        ids = new_ids if past_ids is None else np.concatenate([past_ids, new_ids])
        out = self.model.generate(
            input_ids=torch.from_numpy(ids).view(1, -1), frames=clip if n_vid else None, frames_layout=frames_layout,
            past_key_values=state.get("past_key_values"), return_dict_in_generate=True, do_sample=False,
            repetition_penalty=repetition_penalty, max_new_tokens=max_new_tokens,
            min_new_tokens=max_new_tokens if force_length else None, pad_token_id=self.cfg.eos_token_id)
        seq = out.sequences[0].cpu().numpy()
        state["past_key_values"] = out.past_key_values
        state["past_ids"] = seq[:-1]
        state["turn_index"] = turn + 1
        return seq[len(ids):].tolist(), state

    @torch.inference_mode()
    def mcq_predict(self, clip: Optional[torch.Tensor], question: str, options: List[str], letters=("A", "B", "C", "D"),
                    frames_layout: str = "TCHW", question_prefix: str = "", question_postfix: str = "\nPlease select the correct answer.",
                    answer_prefix: str = "Answer:", abcd_previous_str: str = ": ", subtitles: Optional[str] = None):
        """Prefill-only multiple-choice scoring (ref evaluation/distributed_mcq_predictor.py:33-105): one-shot
        prefill of [video] + question + options + 'Answer:', logits of the LAST prompt position restricted to the option-letter
        ids (`tokenizer(': A').input_ids[-1]`, ...), argmax.  Returns (index of the chosen option, restricted logits).
        Needs tokenizer files (`self.text`)."""
        if self.text is None:
            raise ValueError("mcq_predict needs tokenizer files (pass text=TextFrontEnd(...) or a checkpoint directory)")
        tok = self.text.tokenizer
        letter_ids = [tok(f"{abcd_previous_str}{x}").input_ids[-1] for x in letters]             # ref :91
        query = question_prefix + question + "\n" + "\n".join(options) + question_postfix       # ref :37
        if subtitles is not None:                                                                # ref :50-52
            query = f"This video's subtitles are listed below:\n{subtitles}\nAccording to the video and subtitles, " + query
        grid = None
        if clip is not None:
            grid = protocol.grid_of(clip.shape[0], *(clip.shape[2:] if frames_layout == "TCHW" else clip.shape[1:3]), self.cfg)
        text = self.text.qa_text(query, continuing=False, with_video=grid is not None) + answer_prefix   # ref :56-57
        ids = self.text.encode(text, [grid] if grid is not None else [])
        out = self.model.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=clip, frames_layout=frames_layout,
                                  return_dict_in_generate=True, do_sample=False, repetition_penalty=1.0, max_new_tokens=1,
                                  output_logits=True, pad_token_id=self.cfg.eos_token_id)
        out.past_key_values.release()
        logits = out.logits[0].float().view(-1)[torch.as_tensor(letter_ids, device=out.logits.device)]
        return int(torch.argmax(logits)), logits.cpu()

    @torch.inference_mode()
    def live_cc_once_for_evaluation(self, clip: torch.Tensor, frames_layout: str = "TCHW", max_new_tokens: int = 32,
                                    repetition_penalty: float = 1.05, video_start: float = 0.0, force_length: bool = False,
                                    query: Optional[str] = None):
        """Offline replay of a whole clip (ref infer.py:244-310): chunks 6,2,2,...; returns [[t0, t1, text], ...]."""
        state: dict = {}
        responses = []
        for (a, b), text, state in self.live_cc(clip, state, frames_layout=frames_layout, repetition_penalty=repetition_penalty,
                                                max_new_tokens=max_new_tokens, force_length=force_length, message=query):
            responses.append([video_start + a, video_start + b, text])
        self.last_state = state
        return responses
