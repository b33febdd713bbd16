"""Frame fetch -> resize front end (the step immediately before the hot path): what the reference does on the
CPU in `livecc_utils/video_process_patch.py` --

  * `get_smart_resized_video_reader` (109-124): target size from `smart_resize` under the LiveCC pixel budgets;
  * `get_smart_resized_clip` (126-156): map each requested timestamp to the first frame with pts >= t, keep an even number
    of frames, then `torchvision.transforms.functional.resize(clip_uint8, [h, w], BICUBIC, antialias=True)`.

The resize runs on the GPU (`lcc_resize_bicubic_aa_u8`).  torchvision's tensor path casts uint8 -> float32, calls
`torch.nn.functional.interpolate(mode="bicubic", antialias=True, align_corners=False)`, clamps to [0, 255], rounds half to
even and casts back; ATen's kernel (aten/src/ATen/native/cpu/UpSampleKernel.cpp, `_compute_indices_min_size_weights_aa`,
`interpolate_aa_single_dim`) is separable: width pass, then height pass, fp32, taps accumulated in order with fused
multiply-adds.  The per-output-pixel tap tables are built HERE on the host with exactly ATen's float arithmetic (including its
FMA contractions -- verified bit for bit against torch in tests/test_resize.py), the device only accumulates.  Video decode
itself (decord / ffmpeg) stays external: frames arrive as a uint8 tensor.
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import protocol

# qwen_vl_utils.vision_process constants as the reference configures them (video_process_patch.py:10-14; env-overridable)
IMAGE_FACTOR = 28
FRAME_FACTOR = 2
FPS = 2.0
VIDEO_MIN_PIXELS = int(os.environ.get("VIDEO_MIN_PIXELS", 100 * 28 * 28))
VIDEO_MAX_PIXELS = 768 * 28 * 28                                              # qwen_vl_utils module constant
VIDEO_TOTAL_PIXELS = int(float(os.environ.get("VIDEO_MAX_PIXELS", 24576 * 28 * 28)))   # the env var feeds TOTAL (sic)
FPS_MAX_FRAMES = int(os.environ.get("FPS_MAX_FRAMES", 480))

_f32 = np.float32


def smart_resized_hw(height: int, width: int, n_frames: int, max_pixels: int = None) -> Tuple[int, int]:
    """Target (h, w) of a video (ref video_process_patch.py:109-124): per-frame pixel budget
    max(min(VIDEO_MAX_PIXELS, TOTAL / nframes * 2), 1.05 * MIN), then `smart_resize` to multiples of 28."""
    nframes = min(n_frames, FPS_MAX_FRAMES)
    if max_pixels is None:
        max_pixels = max(min(VIDEO_MAX_PIXELS, VIDEO_TOTAL_PIXELS / nframes * FRAME_FACTOR), int(VIDEO_MIN_PIXELS * 1.05))
    return protocol.smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=VIDEO_MIN_PIXELS, max_pixels=max_pixels)


def select_clip_frames(timestamps: Sequence[float], video_pts: Sequence[float], index_from: int = 0):
    """ref video_process_patch.py:134-145: pad the timestamps to an even count (+1/FPS), map each to the first frame whose pts
    is >= the timestamp (monotone scan from `index_from`), stop at the end of the video, drop a trailing odd frame.
    Returns (frame indices, timestamps) exactly as the reference leaves them: when the video ends early the timestamp list
    is NOT cut to the number of frames found (only the odd-frame trim removes one from both) -- `live_cc` then stores
    `timestamps[-1]` as `last_timestamp` (ref demo/infer.py:118), so the quirk is part of the protocol."""
    ts = [float(t) for t in timestamps]
    while len(ts) % FRAME_FACTOR != 0:
        ts.append(ts[-1] + 1 / FPS)
    idxs: List[int] = []
    i = int(index_from)
    n = len(video_pts)
    for t in ts:
        while i < n and video_pts[i] < t:
            i += 1
        if i >= n:
            break
        idxs.append(i)
    while len(idxs) % FRAME_FACTOR != 0:
        idxs = idxs[:-1]
        ts = ts[:-1]
    return idxs, ts


# ---- ATen's antialias bicubic tap tables, bit for bit ----
def _fma(a, b, c):
    return _f32(np.float64(a) * np.float64(b) + np.float64(c))


def _aa_cubic(x):
    """`aa_filter` of HelperInterpCubic (A = -0.5) with the Horner steps fused as the compiled ATen kernel does."""
    x = _f32(abs(x))
    A = _f32(-0.5)
    if x < _f32(1):
        s = _fma(_f32(A + _f32(2)), x, -_f32(A + _f32(3)))
        return _fma(_f32(s * x), x, _f32(1))
    if x < _f32(2):
        s = _fma(A, x, -_f32(_f32(5) * A))
        s = _fma(s, x, _f32(_f32(8) * A))
        return _fma(s, x, -_f32(_f32(4) * A))
    return _f32(0)


def aa_bicubic_taps(in_size: int, out_size: int):
    """(xmin int32 [out], xsize int32 [out], weights fp32 [out, K]) of `_compute_indices_min_size_weights_aa` for
    align_corners=False: support 2*scale when downscaling (the antialias widening), taps normalised to sum 1."""
    scale = _f32(_f32(in_size) / _f32(out_size))
    support = _f32(2.0 * float(scale)) if scale >= 1.0 else _f32(2.0)
    invscale = _f32(1.0 / float(scale)) if scale >= 1.0 else _f32(1.0)
    K = int(math.ceil(float(support))) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    W = np.zeros((out_size, K), np.float32)
    half = _f32(_f32(0.5) * invscale)
    for i in range(out_size):
        center = _f32(float(scale) * (i + 0.5))
        lo = max(int(float(center) - float(support) + 0.5), 0)
        hi = min(int(float(center) + float(support) + 0.5), in_size)
        n = hi - lo
        xmin[i], xsize[i] = lo, n
        tot = _f32(0)
        for j in range(n):
            W[i, j] = _aa_cubic(_fma(_f32(_f32(j + lo) - center), invscale, half))
            tot = _f32(tot + W[i, j])
        if tot != 0:
            for j in range(n):
                W[i, j] = _f32(W[i, j] / tot)
    return xmin, xsize, W


_TAPS_CACHE = {}


def _taps_on(device, in_size, out_size, tap_major=False):
    """tap_major: weights as [K][out] (the width pass reads them coalesced across output columns), else [out][K]."""
    key = (str(device), in_size, out_size, tap_major)
    if key not in _TAPS_CACHE:
        xmin, xsize, W = aa_bicubic_taps(in_size, out_size)
        Wd = np.ascontiguousarray(W.T if tap_major else W)
        _TAPS_CACHE[key] = (torch.from_numpy(xmin).to(device), torch.from_numpy(xsize).to(device), torch.from_numpy(Wd).to(device), W.shape[1])
    return _TAPS_CACHE[key]


def resize_bicubic_aa(frames: torch.Tensor, height: int, width: int, layout: str = "THWC") -> torch.Tensor:
    """uint8 frames on the GPU ([T,H,W,3] as a decoder delivers them, or [T,3,H,W]) -> uint8 [T,3,height,width], the values
    `torchvision.transforms.functional.resize(clip, [h, w], BICUBIC, antialias=True)` gives (ref video_process_patch.py:150-155)."""
    from . import _lib
    if frames.dtype != torch.uint8 or frames.dim() != 4 or not frames.is_cuda:
        raise ValueError("frames must be a uint8 [T,H,W,C] / [T,C,H,W] tensor on the GPU (livecc_amd has no CPU path)")
    frames = frames.contiguous()
    if layout == "THWC":
        T, Hin, Win, C = frames.shape
    elif layout == "TCHW":
        T, C, Hin, Win = frames.shape
    else:
        raise ValueError(layout)
    if C != 3:
        raise ValueError("3 channels expected")
    dev = frames.device
    xmin, xsize, wx, kx = _taps_on(dev, Win, width, tap_major=True)
    ymin, ysize, wy, ky = _taps_on(dev, Hin, height)
    from . import torch_ops as _T
    t = _T.op("resize_bicubic_aa")          # dispatcher path (torch.ops.livecc_amd.resize_bicubic_aa) when the registration library is present
    if t is not None:
        return t(frames, 0 if layout == "THWC" else 1, int(height), int(width), xmin, xsize, wx, int(kx), ymin, ysize, wy, int(ky))
    out = torch.empty(T, 3, height, width, dtype=torch.uint8, device=dev)
    tmp = torch.empty(T * 3 * Hin * width, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().lcc_resize_bicubic_aa_u8(
        frames.data_ptr(), 0 if layout == "THWC" else 1, T, Hin, Win, out.data_ptr(), height, width, xmin.data_ptr(), xsize.data_ptr(),
        wx.data_ptr(), kx, ymin.data_ptr(), ysize.data_ptr(), wy.data_ptr(), ky, tmp.data_ptr(),
        torch.cuda.current_stream(dev).cuda_stream), "lcc_resize_bicubic_aa_u8")
    return out


def get_smart_resized_clip(video, resized_height: int, resized_width: int, timestamps, video_pts,
                           video_pts_index_from: int = 0, layout: str = "THWC", device=None):
    """ref video_process_patch.py:126-156: returns (clip uint8 [n,3,h,w], kept timestamps, frame indices) -- the tensor `live_cc`
    feeds to the processor.  `video`: a decoded-video reader with `get_batch(idxs, device)` (decord's interface: ONLY the selected
    frames travel to `device`, as in the reference) or a uint8 tensor of all frames that already lives on the GPU."""
    idxs, ts = select_clip_frames(timestamps, video_pts, video_pts_index_from)
    if not idxs:
        return None, ts, idxs
    if hasattr(video, "get_batch"):
        sel = video.get_batch(idxs, device=device)
    else:
        sel = video.index_select(0, torch.as_tensor(idxs, device=video.device))
    return resize_bicubic_aa(sel, resized_height, resized_width, layout), ts, idxs
