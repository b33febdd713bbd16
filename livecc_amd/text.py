"""Text side of the streaming turn with a REAL tokenizer: what the reference does with
`processor.apply_chat_template` + `processor(text=..., videos=[clip])` + `processor.decode` (ref demo/infer.py:49, 51-58,
134-157, 175), without the HF *video* processor (it needs torchvision; normalise + patchify run on the GPU in this library).

Only the tokenizer files of the checkpoint directory are needed (`tokenizer.json` / `vocab.json` + `merges.txt`,
`tokenizer_config.json`); everything runs on the host and produces the `input_ids` that `generate()` consumes.

  * chat template : the tokenizer's own `chat_template` when it has one, else the Qwen2-VL template restated here
                    (system prompt only before the first message, `<|vision_start|><|video_pad|><|vision_end|>` per video,
                    generation prompt `<|im_start|>assistant\\n`);
  * placeholder expansion : one `<|video_pad|>` -> N = t*h*w / merge^2 copies (HF processing_qwen2_vl.py:63-66);
  * `system_prompt_offset` : index of '<|im_start|>user' in the templated probe message (ref demo/infer.py:51-58) -- later
                    turns are `'<|im_end|>\\n' + text[offset:]` (ref demo/infer.py:149-150);
  * `streaming_eos_token_id` : last id of tokenizer(' ...') (ref demo/infer.py:49).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

DEFAULT_SYSTEM = "You are a helpful assistant."
VISION_START, VISION_END, VIDEO_PAD, IMAGE_PAD = "<|vision_start|>", "<|vision_end|>", "<|video_pad|>", "<|image_pad|>"
IM_START, IM_END = "<|im_start|>", "<|im_end|>"


def qwen2vl_chat_template(conversation: Sequence[dict], add_generation_prompt: bool = True) -> str:
    """The Qwen2-VL chat template for the message shapes the reference builds (text / video / image content items)."""
    out = []
    for i, msg in enumerate(conversation):
        if i == 0 and msg["role"] != "system":
            out.append(f"{IM_START}system\n{DEFAULT_SYSTEM}{IM_END}\n")
        out.append(f"{IM_START}{msg['role']}\n")
        content = msg["content"]
        if isinstance(content, str):
            out.append(content)
        else:
            for item in content:
                kind = item.get("type")
                if kind == "video" or "video" in item:
                    out.append(f"{VISION_START}{VIDEO_PAD}{VISION_END}")
                elif kind == "image" or "image" in item or "image_url" in item:
                    out.append(f"{VISION_START}{IMAGE_PAD}{VISION_END}")
                elif "text" in item:
                    out.append(item["text"])
        out.append(f"{IM_END}\n")
    if add_generation_prompt:
        out.append(f"{IM_START}assistant\n")
    return "".join(out)


class TextFrontEnd:
    """Tokenizer-backed builder of streaming-turn ids; the real-text counterpart of `protocol.TurnBuilder`."""

    def __init__(self, tokenizer_or_path, cfg=None):
        if isinstance(tokenizer_or_path, str):
            from transformers import AutoTokenizer
            tokenizer_or_path = AutoTokenizer.from_pretrained(tokenizer_or_path)
        self.tokenizer = tok = tokenizer_or_path
        self.video_token_id = tok.convert_tokens_to_ids(VIDEO_PAD)
        self.eos_token_id = tok.convert_tokens_to_ids(IM_END)
        self.endoftext_token_id = tok.convert_tokens_to_ids("<|endoftext|>")
        if cfg is not None:
            # config.json's eos is <|im_end|> for the instruct checkpoints and <|endoftext|> for base-style ones
            if cfg.video_token_id != self.video_token_id or cfg.eos_token_id not in (self.eos_token_id, self.endoftext_token_id):
                raise ValueError(f"tokenizer special ids (video_pad {self.video_token_id}, im_end {self.eos_token_id}, endoftext "
                                 f"{self.endoftext_token_id}) do not match the model config ({cfg.video_token_id}, {cfg.eos_token_id})")
        self.merge = 2 if cfg is None else cfg.spatial_merge_size
        self.streaming_eos_token_id = tok(" ...").input_ids[-1]                        # ref demo/infer.py:49
        probe = self.apply_chat_template([{"role": "user", "content": [{"type": "text", "text": "livecc"}]}], add_generation_prompt=False)
        self.system_prompt_offset = probe.index(f"{IM_START}user")                     # ref demo/infer.py:51-58

    # ---- the three processor calls of the reference ----
    def apply_chat_template(self, conversation, add_generation_prompt: bool = True) -> str:
        if getattr(self.tokenizer, "chat_template", None):
            n_vid = sum(1 for m in conversation if not isinstance(m["content"], str)
                        for it in m["content"] if it.get("type") == "video" or "video" in it)
            try:
                out = self.tokenizer.apply_chat_template(conversation, tokenize=False, add_generation_prompt=add_generation_prompt)
                if out.count(VIDEO_PAD) == n_vid and "{'type'" not in out:
                    return out
            except Exception:
                pass
            # a text-only template cannot render the video items (it prints the content list): use the Qwen2-VL restatement
        return qwen2vl_chat_template(conversation, add_generation_prompt)

    def encode(self, text: str, video_grids: Sequence[Sequence[int]] = ()) -> np.ndarray:
        """`processor(text=..., videos=[clip])`: expand the i-th `<|video_pad|>` to t*h*w/merge^2 copies, then tokenize."""
        parts = text.split(VIDEO_PAD)
        if len(parts) - 1 != len(video_grids):
            raise ValueError(f"{len(parts) - 1} video placeholders in the text but {len(video_grids)} clips")
        out = [parts[0]]
        for g, rest in zip(video_grids, parts[1:]):
            t, h, w = (int(x) for x in g)
            out.append(VIDEO_PAD * (t * h * w // (self.merge * self.merge)))
            out.append(rest)
        ids = self.tokenizer("".join(out), add_special_tokens=False).input_ids
        return np.asarray(ids, dtype=np.int64)

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        return self.tokenizer.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens)

    # ---- one streaming turn (ref demo/infer.py:134-150) ----
    def turn_text(self, start: float, stop: float, query: Optional[str], continuing: bool, with_video: bool = True) -> str:
        content: List[dict] = [{"type": "text", "text": f"Time={start:.1f}-{stop:.1f}s"}]
        if with_video:
            content.append({"type": "video", "video": None})
        if query:
            content.append({"type": "text", "text": query})
        text = self.apply_chat_template([{"role": "user", "content": content}], add_generation_prompt=True)
        if continuing:
            text = f"{IM_END}\n" + text[self.system_prompt_offset:]
        return text

    def turn_ids(self, start: float, stop: float, grid, query: Optional[str], continuing: bool) -> np.ndarray:
        return self.encode(self.turn_text(start, stop, query, continuing, with_video=grid is not None), [grid] if grid is not None else [])

    def qa_text(self, query: str, continuing: bool, with_video: bool) -> str:
        """video_qa turn (ref demo/infer.py:205-228): [video] + query, no Time= prefix."""
        content: List[dict] = []
        if with_video:
            content.append({"type": "video", "video": None})
        content.append({"type": "text", "text": query})
        text = self.apply_chat_template([{"role": "user", "content": content}], add_generation_prompt=True)
        if continuing:
            text = f"{IM_END}\n" + text[self.system_prompt_offset:]
        return text
