"""TEST INFRASTRUCTURE ONLY (never imported by livecc_amd/): CPU reference of the frame fetch -> resize step in front of the hot
path.

`resize_ref` is what `torchvision.transforms.functional.resize(uint8_clip, [h, w], InterpolationMode.BICUBIC, antialias=True)`
computes (ref livecc_utils/video_process_patch.py:150-155).  torchvision is not installed here; its tensor implementation
(`torchvision/transforms/_functional_tensor.py: resize`) is four torch calls, restated below: cast uint8 -> float32,
`torch.nn.functional.interpolate(mode="bicubic", align_corners=False, antialias=True)`, clamp to [0, 255] (bicubic overshoot),
`torch.round` + cast back to uint8.  The interpolation itself is torch's own CPU kernel -- the real thing, not a restatement.

`select_clip_frames_ref` restates ref video_process_patch.py:134-145 line by line.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def resize_ref(clip_tchw_u8: torch.Tensor, height: int, width: int) -> torch.Tensor:
    assert clip_tchw_u8.dtype == torch.uint8 and clip_tchw_u8.dim() == 4
    x = clip_tchw_u8.cpu().to(torch.float32)                                  # _cast_squeeze_in: uint8 -> float32
    y = F.interpolate(x, size=[height, width], mode="bicubic", align_corners=False, antialias=True)
    y = y.clamp(min=0, max=255)                                               # "if interpolation == bicubic and out_dtype == uint8"
    return torch.round(y).to(torch.uint8)                                     # _cast_squeeze_out: round, then cast


def resize_float_ref(clip_tchw_u8: torch.Tensor, height: int, width: int) -> torch.Tensor:
    return F.interpolate(clip_tchw_u8.cpu().to(torch.float32), size=[height, width], mode="bicubic", align_corners=False, antialias=True)


def select_clip_frames_ref(timestamps, video_pts, video_pts_index_from=0, frame_factor=2, fps=2.0):
    timestamps = torch.as_tensor(timestamps, dtype=torch.float64)
    while len(timestamps) % frame_factor != 0:
        timestamps = torch.cat([timestamps, timestamps[-1:] + 1 / fps])
    clip_idxs = []
    for timestamp in timestamps:
        while video_pts_index_from < len(video_pts) and video_pts[video_pts_index_from] < timestamp:
            video_pts_index_from += 1
        if video_pts_index_from >= len(video_pts):
            break
        clip_idxs.append(video_pts_index_from)
    while len(clip_idxs) % frame_factor != 0:
        clip_idxs = clip_idxs[:-1]
        timestamps = timestamps[:-1]
    return clip_idxs, timestamps.tolist()
