"""Decoded-video side of the L4/L5 surface: what the reference does between a video FILE and the uint8 clip tensor that
enters the hot path (ref livecc-utils/src/livecc_utils/video_process_patch.py:24-124 + the `qwen_vl_utils.vision_process`
helpers it imports).  Decoding itself (decord / ffmpeg) is external here as it is a C++ dependency in the reference: a video
is a `DecodedVideo` = frames (uint8, decoder order [N,H,W,3]) + per-frame pts + average fps, registered under the path string
that `state['video_path']` / `video=` carries, or produced on demand by a caller-supplied loader.  Everything after the decoder
is restated here as host integer logic (index selection) plus the GPU resize of `livecc_amd.resize`:

  * `smart_nframes`               qwen_vl_utils.vision_process (third-party, not under /root/reference, no pinned version in
                                  the reference's README:26; restated from the 0.0.10/0.0.11 source the reference imports at
                                  video_process_patch.py:16-19 -- FPS=2.0, FRAME_FACTOR=2, FPS_MIN_FRAMES=4, and
                                  FPS_MAX_FRAMES as the reference overrides it, :14)
  * `read_video_decord_plus`      ref video_process_patch.py:24-83: [video_start, video_end] cut by pts, then either
                                  `smart_nframes` + linspace sampling (default) or the strict-fps branch (first frame with
                                  pts >= every expected 1/FPS timestamp, FPS_MAX_FRAMES cap, padding to an even count)
  * `spatial_resize_video`        ref :88-107 (per-nframes pixel budget -> smart_resize -> antialias bicubic uint8 resize)
  * `get_smart_resized_video_reader`  ref :109-124 (target size decided once per video)

The index-selection functions are pure numpy and are pinned against the reference's own functions executed with stubbed
decoders (oracle/make_frame_selection_golden.py -> tests/golden/frame_selection.json).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import protocol
from . import resize as R

FPS_MIN_FRAMES = 4          # qwen_vl_utils.vision_process.FPS_MIN_FRAMES


@dataclasses.dataclass
class DecodedVideo:
    """What a decoder delivers: `frames` uint8 [N,H,W,3] (THWC, host or GPU), `pts` seconds per frame (decord's
    `_frame_pts[:, 1]`, ref demo/infer.py:93), `avg_fps` (decord `get_avg_fps()`)."""
    frames: torch.Tensor
    pts: np.ndarray
    avg_fps: Optional[float] = None
    layout: str = "THWC"

    def __post_init__(self):
        self.pts = np.asarray(self.pts, dtype=np.float64)
        if self.frames.dtype != torch.uint8 or self.frames.dim() != 4:
            raise ValueError("frames must be uint8 [N,H,W,3] (THWC) or [N,3,H,W] (TCHW)")
        if len(self.pts) != self.frames.shape[0]:
            raise ValueError(f"{self.frames.shape[0]} frames but {len(self.pts)} pts")
        if self.avg_fps is None:
            span = float(self.pts[-1] - self.pts[0]) if len(self.pts) > 1 else 0.0
            self.avg_fps = (len(self.pts) - 1) / span if span > 0 else float(R.FPS)

    def __len__(self) -> int:
        return int(self.frames.shape[0])

    @property
    def height_width(self) -> Tuple[int, int]:
        s = self.frames.shape
        return (int(s[1]), int(s[2])) if self.layout == "THWC" else (int(s[2]), int(s[3]))

    def get_batch(self, idxs: Sequence[int], device=None) -> torch.Tensor:
        """decord `VideoReader.get_batch(idxs)` (repeated indices allowed), moved to `device`."""
        idx = torch.as_tensor(list(idxs), dtype=torch.long, device=self.frames.device)
        out = self.frames.index_select(0, idx)
        return out.to(device) if device is not None else out


# ---- the registry that stands where the reference opens files with decord ----
_VIDEOS: Dict[str, DecodedVideo] = {}
_LOADER: Optional[Callable[[str], DecodedVideo]] = None


def register_video(path: str, video: DecodedVideo) -> None:
    _VIDEOS[str(path)] = video


def unregister_video(path: str) -> None:
    _VIDEOS.pop(str(path), None)


def set_video_loader(loader: Optional[Callable[[str], DecodedVideo]]) -> None:
    """`loader(path) -> DecodedVideo`: plug a decoder (decord, PyAV, a hardware decoder) in; called for unregistered paths."""
    global _LOADER
    _LOADER = loader


def open_video(video, remote_loader: Optional[Callable] = None) -> DecodedVideo:
    """ref video_process_patch.py:38-44: `video` is a path; here also a DecodedVideo.  Unknown paths raise like the
    reference's `ValueError(video_path not found)`."""
    if isinstance(video, DecodedVideo):
        return video
    key = str(video)
    if key in _VIDEOS:
        return _VIDEOS[key]
    if remote_loader is not None:
        got = remote_loader(key)
        if isinstance(got, DecodedVideo):
            return got
    if _LOADER is not None:
        return _LOADER(key)
    raise ValueError(f"video_path {key} not found: register a DecodedVideo for it (livecc_amd.video.register_video) or install "
                     f"a decoder with set_video_loader -- video decoding is external to this library")


# ---- qwen_vl_utils.vision_process helpers the reference imports ----
def _round_by_factor(x, f):
    return round(x / f) * f


def _ceil_by_factor(x, f):
    return math.ceil(x / f) * f


def _floor_by_factor(x, f):
    return math.floor(x / f) * f


def smart_nframes(ele: dict, total_frames: int, video_fps: float) -> int:
    """Number of frames to sample (qwen_vl_utils.vision_process.smart_nframes): `nframes` rounded to FRAME_FACTOR, or
    total_frames / video_fps * fps clamped to [min_frames, max_frames, total_frames] and floored to FRAME_FACTOR."""
    if "fps" in ele and "nframes" in ele:
        raise ValueError("Only accept either `fps` or `nframes`")
    if "nframes" in ele:
        nframes = _round_by_factor(ele["nframes"], R.FRAME_FACTOR)
    else:
        fps = ele.get("fps", R.FPS)
        min_frames = _ceil_by_factor(ele.get("min_frames", FPS_MIN_FRAMES), R.FRAME_FACTOR)
        max_frames = _floor_by_factor(ele.get("max_frames", min(R.FPS_MAX_FRAMES, total_frames)), R.FRAME_FACTOR)
        nframes = total_frames / video_fps * fps
        nframes = min(min(max(nframes, min_frames), max_frames), total_frames)
        nframes = _floor_by_factor(nframes, R.FRAME_FACTOR)
    if not (R.FRAME_FACTOR <= nframes <= total_frames):
        raise ValueError(f"nframes should in interval [{R.FRAME_FACTOR}, {total_frames}], but got {nframes}.")
    return int(nframes)


def select_video_frames(video_pts: np.ndarray, n_total: int, video_fps: float, ele: dict, strict_fps: bool = False,
                        drop_last: bool = True):
    """The index arithmetic of `_read_video_decord_plus` (ref video_process_patch.py:45-80) without the decoder.
    Returns (clip_idxs list[int], clip_pts list[float] or None, total_frames)."""
    video_start, video_end = ele.get("video_start"), ele.get("video_end")
    clip_idxs, clip_pts = None, None
    if video_start is not None or video_end is not None:                      # ref :49-56
        video_pts = np.asarray(video_pts, dtype=np.float64)
        vs = video_pts[0] if not video_start else video_start
        ve = video_pts[-1] if not video_end else video_end
        clip_idxs = ((vs <= video_pts) & (video_pts <= ve)).nonzero()[0]
        clip_pts = video_pts[clip_idxs]
        total_frames = len(clip_idxs)
    else:
        total_frames = int(n_total)
    if not strict_fps:                                                        # ref :59-62
        nframes = smart_nframes(ele, total_frames=total_frames, video_fps=video_fps)
        nf = np.linspace(0, total_frames - 1, nframes).round().astype(int)
        clip_idxs = nf if clip_idxs is None else clip_idxs[nf]
        # NOTE the reference does not re-index clip_pts here: with return_pts it hands back the pts of the WHOLE cut range
        return [int(i) for i in clip_idxs], (None if clip_pts is None else [float(p) for p in clip_pts]), total_frames
    if clip_pts is None:                                                      # ref :64-67
        clip_pts = np.asarray(video_pts, dtype=np.float64)
        clip_idxs = np.arange(len(clip_pts))
    expected = np.arange(clip_pts[0], clip_pts[-1] + 1e-6, 1 / R.FPS)         # ref :68
    if len(expected) > R.FPS_MAX_FRAMES:                                      # ref :69-73
        if drop_last:
            expected = expected[:R.FPS_MAX_FRAMES]
        else:
            expected = expected[np.linspace(0, len(expected) - 1, R.FPS_MAX_FRAMES).round().astype(int)]
    first_ge = (expected[:, None] <= clip_pts).argmax(axis=1)                 # ref :74
    pts_l, idx_l = clip_pts[first_ge].tolist(), clip_idxs[first_ge].tolist()
    while len(idx_l) % R.FRAME_FACTOR != 0:                                   # ref :76-78
        idx_l.append(idx_l[-1])
        pts_l.append(pts_l[-1])
    return [int(i) for i in idx_l], [float(p) for p in pts_l], total_frames


def read_video_decord_plus(ele: dict, strict_fps: bool = False, drop_last: bool = True, return_pts: bool = False, device=None):
    """ref video_process_patch.py:24-83 on a decoded video: returns (clip uint8 [T,3,H,W], sample_fps[, clip_pts]).  The
    reference's note applies here too: without video_start/video_end and strict_fps=False, `clip_pts` is None (:47,82)."""
    vr = open_video(ele["video"], ele.get("remote_loader"))
    idxs, pts, total = select_video_frames(vr.pts, len(vr), vr.avg_fps, ele, strict_fps, drop_last)
    clip = vr.get_batch(idxs, device)
    if vr.layout == "THWC":
        clip = clip.permute(0, 3, 1, 2)                                      # ref :79 (a view, like the reference's)
    sample_fps = len(idxs) / max(total, 1e-6) * vr.avg_fps                    # ref :80
    if return_pts:
        return clip, sample_fps, pts
    return clip, sample_fps


def spatial_resize_hw(height: int, width: int, nframes: int) -> Tuple[int, int]:
    """ref video_process_patch.py:93-100: per-frame budget max(min(VIDEO_MAX_PIXELS, TOTAL / nframes * 2), 1.05 * MIN)."""
    max_pixels = max(min(R.VIDEO_MAX_PIXELS, R.VIDEO_TOTAL_PIXELS / nframes * R.FRAME_FACTOR), int(R.VIDEO_MIN_PIXELS * 1.05))
    return protocol.smart_resize(height, width, factor=R.IMAGE_FACTOR, min_pixels=R.VIDEO_MIN_PIXELS, max_pixels=max_pixels)


def spatial_resize_video(video: torch.Tensor, nframes: Optional[int] = None, device=None) -> torch.Tensor:
    """ref video_process_patch.py:88-107: uint8 [T,3,H,W] -> uint8 [T,3,h,w] on the GPU.  (The reference appends `.float()`;
    the values are the rounded uint8 ones either way and the patchify kernel consumes uint8.)"""
    if not nframes:
        nframes, _, height, width = video.shape
    else:
        height, width = video.shape[2:]
    h, w = spatial_resize_hw(int(height), int(width), int(nframes))
    if device is not None and not video.is_cuda:
        video = video.to(device)
    if video.stride(-1) != 1 or not video.is_contiguous():
        # a permuted THWC view (decoder order): resize straight from that layout, no transposed copy
        thwc = video.permute(0, 2, 3, 1)
        if thwc.is_contiguous():
            return R.resize_bicubic_aa(thwc, h, w, layout="THWC")
        video = video.contiguous()
    return R.resize_bicubic_aa(video, h, w, layout="TCHW")


def get_smart_resized_video_reader(video, max_pixels: Optional[int] = None):
    """ref video_process_patch.py:109-124: (reader, resized_height, resized_width)."""
    vr = open_video(video)
    h, w = vr.height_width
    rh, rw = R.smart_resized_hw(h, w, len(vr), max_pixels)
    return vr, rh, rw
