"""Model-shape description for the LiveCC / Qwen2-VL hot path.

The reference never defines shapes itself: `demo/infer.py:43-47` calls
`Qwen2VLForConditionalGeneration.from_pretrained(model_path, ...)` and the shapes come
from the checkpoint's `config.json` (HF `configuration_qwen2_vl.py`, classes
`Qwen2VLVisionConfig` / `Qwen2VLTextConfig` / `Qwen2VLConfig`).  This dataclass is the
single description both the native engine and the HF oracle are built from.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import List


@dataclasses.dataclass
class LiveCCConfig:
    # ---- language model (Qwen2VLTextConfig) ----
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: List[int] = dataclasses.field(default_factory=lambda: [16, 24, 24])
    max_position_embeddings: int = 32768
    tie_word_embeddings: bool = False
    # ---- vision tower (Qwen2VLVisionConfig) ----
    vit_depth: int = 32
    vit_embed_dim: int = 1280
    vit_num_heads: int = 16
    vit_mlp_ratio: int = 4
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    in_channels: int = 3
    # ---- special ids (Qwen2VLConfig) ----
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_id: int = 151645      # <|im_end|>
    bos_token_id: int = 151643      # <|endoftext|>
    name: str = "livecc-7b"

    # ---- derived ----
    @property
    def vit_head_dim(self) -> int:
        return self.vit_embed_dim // self.vit_num_heads

    @property
    def vit_mlp_dim(self) -> int:
        return self.vit_embed_dim * self.vit_mlp_ratio

    @property
    def patch_dim(self) -> int:  # 3*2*14*14 = 1176
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    @property
    def merge_dim(self) -> int:  # 1280*4 = 5120
        return self.vit_embed_dim * self.spatial_merge_size ** 2

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim

    @property
    def qkv_dim(self) -> int:
        return self.q_dim + 2 * self.kv_dim

    @property
    def kv_bytes_per_token(self) -> int:  # bf16 K+V over all layers
        return 2 * self.num_hidden_layers * self.kv_dim * 2

    def validate(self) -> None:
        assert self.head_dim == 128, "LLM attention kernels are built for head_dim 128 (all Qwen2-VL sizes)"
        assert self.vit_head_dim == 80, "ViT attention kernel is built for head_dim 80 (all Qwen2-VL sizes)"
        assert sum(self.mrope_section) * 2 == self.head_dim
        assert self.num_attention_heads % self.num_key_value_heads == 0
        assert self.num_attention_heads // self.num_key_value_heads <= 16
        for d in (self.hidden_size, self.intermediate_size, self.vit_embed_dim, self.vit_mlp_dim):
            assert d % 16 == 0, f"dimension {d} must be a multiple of 16"
        assert self.vocab_size % 16 == 0

    # ---- llm / vit parameter and byte counts ----
    def llm_layer_params(self) -> int:
        h, i = self.hidden_size, self.intermediate_size
        return h * self.qkv_dim + self.qkv_dim + self.q_dim * h + 3 * h * i + 2 * h

    def decode_weight_bytes(self) -> int:
        """bf16 weight bytes streamed by one decode step (layers + final norm + lm_head)."""
        return 2 * (self.num_hidden_layers * self.llm_layer_params() + self.hidden_size
                    + self.vocab_size * self.hidden_size)

    # ---- conversions ----
    def to_hf(self, dtype=None, attn_implementation: str = "sdpa"):
        """Build the HF `Qwen2VLConfig` with the same shapes (oracle side)."""
        from transformers import Qwen2VLConfig
        text = dict(
            vocab_size=self.vocab_size, hidden_size=self.hidden_size,
            intermediate_size=self.intermediate_size, num_hidden_layers=self.num_hidden_layers,
            num_attention_heads=self.num_attention_heads, num_key_value_heads=self.num_key_value_heads,
            max_position_embeddings=self.max_position_embeddings, rms_norm_eps=self.rms_norm_eps,
            rope_parameters={"rope_type": "default", "rope_theta": self.rope_theta,
                             "mrope_section": list(self.mrope_section)},
            bos_token_id=self.bos_token_id, eos_token_id=self.eos_token_id,
        )
        vision = dict(
            depth=self.vit_depth, embed_dim=self.vit_embed_dim, hidden_size=self.hidden_size,
            mlp_ratio=self.vit_mlp_ratio, num_heads=self.vit_num_heads, in_channels=self.in_channels,
            patch_size=self.patch_size, spatial_merge_size=self.spatial_merge_size,
            temporal_patch_size=self.temporal_patch_size,
        )
        cfg = Qwen2VLConfig(
            text_config=text, vision_config=vision,
            image_token_id=self.image_token_id, video_token_id=self.video_token_id,
            vision_start_token_id=self.vision_start_token_id, vision_end_token_id=self.vision_end_token_id,
            tie_word_embeddings=self.tie_word_embeddings,
        )
        cfg._attn_implementation = attn_implementation
        if dtype is not None:
            cfg.dtype = dtype
        return cfg

    @classmethod
    def from_hf(cls, hf_cfg, name: str = "from-hf") -> "LiveCCConfig":
        t, v = hf_cfg.text_config, hf_cfg.vision_config
        rp = t.rope_parameters
        eos = t.eos_token_id if not isinstance(t.eos_token_id, (list, tuple)) else t.eos_token_id[0]
        return cls(
            vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
            num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
            num_key_value_heads=t.num_key_value_heads,
            head_dim=getattr(t, "head_dim", None) or t.hidden_size // t.num_attention_heads,
            rms_norm_eps=t.rms_norm_eps, rope_theta=float(rp["rope_theta"]),
            mrope_section=list(rp["mrope_section"]), max_position_embeddings=t.max_position_embeddings,
            tie_word_embeddings=bool(getattr(hf_cfg, "tie_word_embeddings", False)),
            vit_depth=v.depth, vit_embed_dim=v.embed_dim, vit_num_heads=v.num_heads,
            vit_mlp_ratio=v.mlp_ratio, patch_size=v.patch_size, temporal_patch_size=v.temporal_patch_size,
            spatial_merge_size=v.spatial_merge_size, in_channels=v.in_channels,
            image_token_id=hf_cfg.image_token_id, video_token_id=hf_cfg.video_token_id,
            vision_start_token_id=hf_cfg.vision_start_token_id, vision_end_token_id=hf_cfg.vision_end_token_id,
            eos_token_id=eos, bos_token_id=t.bos_token_id if t.bos_token_id is not None else 151643,
            name=name,
        )

    @classmethod
    def from_pretrained(cls, path: str) -> "LiveCCConfig":
        """Read a checkpoint directory's config.json (what `from_pretrained` does in demo/infer.py:43)."""
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        from transformers import Qwen2VLConfig
        return cls.from_hf(Qwen2VLConfig(**raw), name=os.path.basename(os.path.normpath(path)))


# Public checkpoint shapes (re-verified at load when a checkpoint is given).
def livecc_7b() -> LiveCCConfig:
    return LiveCCConfig(name="livecc-7b")


def qwen2vl_2b() -> LiveCCConfig:
    return LiveCCConfig(vocab_size=151936, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28,
                        num_attention_heads=12, num_key_value_heads=2, tie_word_embeddings=True,
                        name="qwen2vl-2b")


def qwen2vl_2b_untied() -> LiveCCConfig:
    """Qwen2-VL-2B shapes with a separate lm_head (parity tests with the `decisive` synthetic weights, whose lm_head is a row
    permutation of the embedding table)."""
    c = qwen2vl_2b()
    c.tie_word_embeddings, c.name = False, "qwen2vl-2b-untied"
    return c


def qwen2vl_72b() -> LiveCCConfig:
    return LiveCCConfig(hidden_size=8192, intermediate_size=29568, num_hidden_layers=80,
                        num_attention_heads=64, num_key_value_heads=8, name="qwen2vl-72b")


def tiny(vocab_size: int = 2048) -> LiveCCConfig:
    """Small shapes with the true head dims (128 / 80) for fast CPU-oracle parity tests."""
    return LiveCCConfig(
        vocab_size=vocab_size, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, vit_depth=2, vit_embed_dim=160, vit_num_heads=2,
        image_token_id=vocab_size - 9, video_token_id=vocab_size - 8, vision_start_token_id=vocab_size - 12,
        vision_end_token_id=vocab_size - 11, eos_token_id=vocab_size - 3, bos_token_id=vocab_size - 5,
        name="tiny",
    )


def small() -> LiveCCConfig:
    """Mid-size shapes (GQA group 7 like the 7B, 4 layers) for GPU parity runs that finish in seconds."""
    return LiveCCConfig(
        vocab_size=8192, hidden_size=896, intermediate_size=2432, num_hidden_layers=4,
        num_attention_heads=7, num_key_value_heads=1, vit_depth=4, vit_embed_dim=320, vit_num_heads=4,
        image_token_id=8183, video_token_id=8184, vision_start_token_id=8180, vision_end_token_id=8181,
        eos_token_id=8189, bos_token_id=8187, name="small",
    )


PRESETS = {"livecc-7b": livecc_7b, "qwen2vl-2b": qwen2vl_2b, "qwen2vl-2b-untied": qwen2vl_2b_untied, "qwen2vl-72b": qwen2vl_72b,
           "tiny": tiny, "small": small}


def get_config(name: str) -> LiveCCConfig:
    if name in PRESETS:
        return PRESETS[name]()
    if os.path.isdir(name):
        return LiveCCConfig.from_pretrained(name)
    raise KeyError(f"unknown config {name!r}; presets: {sorted(PRESETS)}")
