"""Host-side streaming protocol of the LiveCC hot path (pure Python / integers, no device code).

Everything here restates *orchestration* of the reference, citing the lines it follows:

  * chunking of a clip into a first 6-frame chunk and 2-frame chunks  -> ref demo/infer.py:28-33,121-129
  * turn text layout (system prompt only on turn 1, '<|im_end|>\n' glue) -> ref demo/infer.py:134-150
  * number of <|video_pad|> placeholders N = t*h*w / merge^2            -> HF processing_qwen2_vl.py:63-66
  * M-RoPE position ids of the first turn and `rope_deltas`              -> HF modeling_qwen2_vl.py:914-1016
  * positions of every later turn / decode step = past_len + i + delta   -> HF modeling_qwen2_vl.py:1349-1351
  * smart_resize policy                                                  -> HF video_processing_qwen2_vl.py:40-66
                                                                           and ref livecc_utils/video_process_patch.py:88-124

There are no tokenizer files offline, so turn ids are *synthesised* with the real structure
(special ids at the real places, text ids drawn from a seeded generator); with a checkpoint directory the HF
processor/tokenizer produce the real ids and are passed to `generate()` unchanged.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .config import LiveCCConfig

# ref demo/infer.py:28-33
FPS = 2
INITIAL_FPS_FRAMES = 6
STREAMING_FPS_FRAMES = 2
INITIAL_TIME_INTERVAL = INITIAL_FPS_FRAMES / FPS
STREAMING_TIME_INTERVAL = STREAMING_FPS_FRAMES / FPS
FRAME_TIME_INTERVAL = 1 / FPS


def split_clip(num_frames: int, initialized: bool = False) -> List[Tuple[int, int]]:
    """Frame ranges [(start, stop), ...] of the interleaved chunks (ref demo/infer.py:121-129, 258-264).

    First chunk is 6 frames unless the stream is already initialised; the rest are chunks of 2
    (torch.split semantics: the last chunk may be shorter).
    """
    chunks, pos = [], 0
    if num_frames <= 0:
        return chunks
    if not initialized:
        stop = min(INITIAL_FPS_FRAMES, num_frames)
        chunks.append((0, stop))
        pos = stop
    while pos < num_frames:
        stop = min(pos + STREAMING_FPS_FRAMES, num_frames)
        chunks.append((pos, stop))
        pos = stop
    return chunks


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """HF video_processing_qwen2_vl.py:40-66 (identical to qwen_vl_utils.smart_resize)."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def grid_of(num_frames: int, height: int, width: int, cfg: LiveCCConfig) -> Tuple[int, int, int]:
    """(grid_t, grid_h, grid_w) of a clip; odd frame counts are padded by repeating the last frame
    (HF video_processing_qwen2_vl.py:246-250)."""
    t = (num_frames + cfg.temporal_patch_size - 1) // cfg.temporal_patch_size
    assert height % (cfg.patch_size * cfg.spatial_merge_size) == 0, "height must be a multiple of 28"
    assert width % (cfg.patch_size * cfg.spatial_merge_size) == 0, "width must be a multiple of 28"
    return t, height // cfg.patch_size, width // cfg.patch_size


def num_video_tokens(grid: Sequence[int], cfg: LiveCCConfig) -> int:
    return int(grid[0]) * int(grid[1]) * int(grid[2]) // cfg.spatial_merge_size ** 2


def rope_index_first_turn(ids: np.ndarray, grids: Sequence[Sequence[int]], cfg: LiveCCConfig,
                          text_offset_rule: str = "hf5") -> Tuple[np.ndarray, int]:
    """3-D M-RoPE position ids [3, S] and rope_delta for a sequence processed without a KV cache.

    Follows HF modeling_qwen2_vl.py:914-1016 (`get_rope_index`): text runs get 1-D positions on all
    three axes; a vision run of grid (t,h,w) gets (t, h/2, w/2) meshgrid indices offset by the running
    position.  After a vision run the running position advances by
      * "hf5":  max(h, w) // merge               (transformers 5.15, line 1008 -- what the oracle runs)
      * "hf4":  max(position so far) + 1          (transformers 4.5x rule the checkpoints were trained with)
    The two agree whenever grid_t <= max(h,w)/2, i.e. for every streaming chunk.
    """
    ids = np.asarray(ids).reshape(-1)
    S = ids.shape[0]
    pos = np.zeros((3, S), dtype=np.int64)
    is_vis = (ids == cfg.video_token_id) | (ids == cfg.image_token_id)
    m = cfg.spatial_merge_size
    cur, i, gi = 0, 0, 0
    maxpos = -1
    while i < S:
        j = i
        if not is_vis[i]:
            while j < S and not is_vis[j]:
                j += 1
            n = j - i
            pos[:, i:j] = cur + np.arange(n)[None, :]
            maxpos = max(maxpos, cur + n - 1)
            cur += n
        else:
            while j < S and is_vis[j] and ids[j] == ids[i]:
                j += 1
            t, h, w = (int(x) for x in grids[gi])
            gi += 1
            lt, lh, lw = t, h // m, w // m
            n = lt * lh * lw
            if n != j - i:
                raise ValueError(f"vision run of {j - i} placeholder tokens does not match grid {t, h, w} -> {n}")
            tt, hh, ww = np.meshgrid(np.arange(lt), np.arange(lh), np.arange(lw), indexing="ij")
            pos[0, i:j] = tt.reshape(-1) + cur
            pos[1, i:j] = hh.reshape(-1) + cur
            pos[2, i:j] = ww.reshape(-1) + cur
            maxpos = max(maxpos, cur + max(lt, lh, lw) - 1)
            if text_offset_rule == "hf5":
                cur += max(h, w) // m
            elif text_offset_rule == "hf4":
                cur = maxpos + 1
            else:
                raise ValueError(text_offset_rule)
        i = j
    delta = int(pos.max()) + 1 - S
    return pos, delta


def positions_with_cache(past_len: int, n_new: int, rope_delta: int) -> np.ndarray:
    """[3, n_new] positions of tokens appended to a non-empty cache (HF modeling_qwen2_vl.py:1349-1351):
    plain 1-D positions past_len + i + rope_delta broadcast on the three axes."""
    p = past_len + np.arange(n_new, dtype=np.int64) + rope_delta
    return np.broadcast_to(p[None, :], (3, n_new)).copy()


@dataclasses.dataclass
class TurnBuilder:
    """Builds the token ids of one streaming turn with the reference's layout (ref demo/infer.py:134-150):

      turn 1 : <|im_start|>system\\n{sys}<|im_end|>\\n<|im_start|>user\\nTime=a-bs<|vision_start|>
               <|video_pad|>*N<|vision_end|>{query}<|im_end|>\\n<|im_start|>assistant\\n
      turn k : <|im_end|>\\n<|im_start|>user\\nTime=a-bs<|vision_start|><|video_pad|>*N<|vision_end|>
               <|im_end|>\\n<|im_start|>assistant\\n

    Text pieces are synthetic ids (seeded) with realistic lengths: system 14 ids, user header 3,
    'Time=a-bs' 10, trailer 5; the query is `query_len` ids.  Special ids sit at their real places.
    """
    cfg: LiveCCConfig
    seed: int = 1234
    query_len: int = 5
    text_id_hi: Optional[int] = None     # exclusive upper bound for synthetic text ids

    def __post_init__(self):
        hi = self.text_id_hi
        if hi is None:
            hi = min(self.cfg.bos_token_id, self.cfg.vision_start_token_id, self.cfg.image_token_id,
                     self.cfg.video_token_id, self.cfg.eos_token_id) - 1
        self._hi = int(hi)
        self._rng = np.random.RandomState(self.seed)

    def _text(self, n: int) -> List[int]:
        return [int(x) for x in self._rng.randint(0, self._hi, size=n)]

    def turn_ids(self, turn_index: int, n_video_tokens: int, with_query: Optional[bool] = None) -> np.ndarray:
        c = self.cfg
        with_query = (turn_index == 0) if with_query is None else with_query
        ids: List[int] = []
        if turn_index == 0:
            ids += self._text(14)                 # <|im_start|>system\n...<|im_end|>\n
        else:
            ids += [c.eos_token_id] + self._text(1)   # '<|im_end|>' + '\n'   (ref demo/infer.py:150)
        ids += self._text(3)                      # <|im_start|>user\n
        ids += self._text(10)                     # Time=a-bs
        if n_video_tokens > 0:
            ids += [c.vision_start_token_id] + [c.video_token_id] * n_video_tokens + [c.vision_end_token_id]
        if with_query:
            ids += self._text(self.query_len)
        ids += [c.eos_token_id] + self._text(4)   # <|im_end|>\n<|im_start|>assistant\n
        return np.asarray(ids, dtype=np.int64)


def mm_token_type_ids(ids: np.ndarray, cfg: LiveCCConfig) -> np.ndarray:
    """text 0 / image 1 / video 2 (HF processing_qwen2_vl.py, `mm_token_type_ids`)."""
    ids = np.asarray(ids)
    out = np.zeros_like(ids, dtype=np.int32)
    out[ids == cfg.image_token_id] = 1
    out[ids == cfg.video_token_id] = 2
    return out


def synth_frames(num_frames: int, height: int, width: int, seed: int = 1234, layout: str = "THWC") -> np.ndarray:
    """Seeded uniform 0..255 uint8 frames (`manual_seed(1234 + stream_id)`)."""
    rng = np.random.RandomState(seed)
    f = rng.randint(0, 256, size=(num_frames, height, width, 3), dtype=np.uint8)
    if layout == "THWC":
        return f
    if layout == "TCHW":
        return np.ascontiguousarray(f.transpose(0, 3, 1, 2))
    raise ValueError(layout)
