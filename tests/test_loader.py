"""Checkpoint loading (ref demo/infer.py:43-47): config.json (legacy flat 4.x layout and the nested 5.x
one), sharded safetensors with either parameter naming scheme -> the flat weight arena; CPU only."""
import json
import os

import pytest
import torch

from livecc_amd import weights as W
from livecc_amd.config import LiveCCConfig, livecc_7b, tiny

LEGACY_7B_CONFIG = {   # the layout of the public Qwen2-VL-7B / LiveCC-7B config.json (transformers 4.4x): flat text fields
    "architectures": ["Qwen2VLForConditionalGeneration"], "attention_dropout": 0.0, "bos_token_id": 151643, "eos_token_id": 151645,
    "vision_start_token_id": 151652, "vision_end_token_id": 151653, "vision_token_id": 151654, "image_token_id": 151655,
    "video_token_id": 151656, "hidden_act": "silu", "hidden_size": 3584, "initializer_range": 0.02, "intermediate_size": 18944,
    "max_position_embeddings": 32768, "max_window_layers": 28, "model_type": "qwen2_vl", "num_attention_heads": 28,
    "num_hidden_layers": 28, "num_key_value_heads": 4, "rms_norm_eps": 1e-06, "rope_theta": 1000000.0, "sliding_window": 32768,
    "tie_word_embeddings": False, "torch_dtype": "bfloat16", "transformers_version": "4.41.2", "use_cache": True,
    "use_sliding_window": False,
    "vision_config": {"depth": 32, "embed_dim": 1280, "mlp_ratio": 4, "num_heads": 16, "in_chans": 3, "hidden_size": 3584,
                      "patch_size": 14, "spatial_merge_size": 2, "spatial_patch_size": 14, "temporal_patch_size": 2},
    "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "vocab_size": 152064,
}


def test_legacy_flat_config_json_gives_the_7b_shapes(tmp_path):
    d = tmp_path / "LiveCC-7B-Instruct"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(LEGACY_7B_CONFIG))
    cfg = LiveCCConfig.from_pretrained(str(d))
    ref = livecc_7b()
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
              "head_dim", "rms_norm_eps", "rope_theta", "vit_depth", "vit_embed_dim", "vit_num_heads", "patch_size",
              "temporal_patch_size", "spatial_merge_size", "image_token_id", "video_token_id", "vision_start_token_id",
              "vision_end_token_id", "eos_token_id", "tie_word_embeddings"):
        assert getattr(cfg, f) == getattr(ref, f), f
    assert list(cfg.mrope_section) == [16, 24, 24] and cfg.name == "LiveCC-7B-Instruct"


@pytest.mark.parametrize("naming", ["hf5", "hf4"])
@pytest.mark.parametrize("llm_fp8", [False, True])
def test_sharded_safetensors_load_into_the_arena(tmp_path, naming, llm_fp8):
    from safetensors.torch import save_file
    from oracle import hf_oracle as O
    cfg = tiny()
    hf = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=4)
    sd = {k: v.contiguous() for k, v in hf.state_dict().items()}
    if naming == "hf4":      # 4.5x checkpoints: visual.* / model.layers.* / model.embed_tokens / model.norm / lm_head
        sd = {k.replace("model.visual.", "visual.").replace("model.language_model.", "model."): v for k, v in sd.items()}
    keys = sorted(sd)
    d = tmp_path / "ckpt"
    d.mkdir()
    save_file({k: sd[k] for k in keys[: len(keys) // 2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[len(keys) // 2:]}, str(d / "model-00002-of-00002.safetensors"))
    (d / "config.json").write_text(json.dumps(cfg.to_hf().to_dict(), default=str))
    cfg2 = LiveCCConfig.from_pretrained(str(d))
    assert (cfg2.hidden_size, cfg2.num_hidden_layers, cfg2.vit_embed_dim, cfg2.video_token_id) == \
           (cfg.hidden_size, cfg.num_hidden_layers, cfg.vit_embed_dim, cfg.video_token_id)
    got = W.from_pretrained(str(d), cfg, "cpu", llm_fp8=llm_fp8)
    ref = W.from_hf_model(hf, cfg, "cpu", llm_fp8=llm_fp8)
    assert got.llm_fp8 == llm_fp8 and got.names() == ref.names()
    for name in ref.names():
        assert torch.equal(got.view(name), ref.view(name)), name
    with pytest.raises(FileNotFoundError):
        W.from_pretrained(str(tmp_path), cfg, "cpu")
    os.remove(d / "model-00002-of-00002.safetensors")
    with pytest.raises(KeyError):
        W.from_pretrained(str(d), cfg, "cpu")


def test_read_generation_config_gives_sampling_defaults_and_eos_list(tmp_path):
    """`from_pretrained` and the offline driver both take the checkpoint's generation_config.json: sampling defaults and the EOS id
    list ([<|im_end|>, <|endoftext|>] in the released checkpoints); a scalar EOS becomes a one-element list; no file -> ({}, None)."""
    import json
    from livecc_amd.modeling import read_generation_config
    assert read_generation_config(str(tmp_path)) == ({}, None)
    (tmp_path / "generation_config.json").write_text(json.dumps(
        {"do_sample": True, "top_k": 1, "top_p": 0.001, "temperature": 0.01, "eos_token_id": [151645, 151643]}))
    g, eos = read_generation_config(str(tmp_path))
    assert g["top_k"] == 1 and g["do_sample"] is True and eos == [151645, 151643]
    (tmp_path / "generation_config.json").write_text(json.dumps({"eos_token_id": 151645}))
    assert read_generation_config(str(tmp_path))[1] == [151645]
