#!/usr/bin/env python
"""A FULL-SIZE checkpoint through the loader.  No real LiveCC checkpoint exists offline (ref .MISSING_LARGE_BLOBS), so the next best thing:
the complete LiveCC-7B parameter set (8.29 B parameters, 16.6 GB of bf16) written as an HF-style sharded safetensors directory --
`model-0000i-of-0000n.safetensors` shards of <= 5 GB in the checkpoint's own key scheme (transformers 4.5x: `model.layers.*`, `visual.*`,
`lm_head.weight`) + `model.safetensors.index.json` -- and read back through `weights.from_pretrained`, the path ref demo/infer.py:43-47
takes.  The arena it builds must equal, byte for byte, the arena `WeightArena.fill_tiled` builds from the same seeded values in HBM.

    python tools/full_size_loader_check.py [--dir /tmp/lcc_ckpt] [--fp8]        (one JSON line; the directory is removed afterwards)
"""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.weights import WeightArena, from_pretrained, hf_param_shapes, synthetic_param  # noqa: E402

a = sys.argv[1:]
root = a[a.index("--dir") + 1] if "--dir" in a else "/tmp/lcc_ckpt"
fp8 = "--fp8" in a
cfg = get_config("livecc-7b")
dev = torch.device("cuda:0")
shapes = hf_param_shapes(cfg)


def hf4_key(name: str) -> str:      # the key scheme of the published Qwen2-VL / LiveCC checkpoints (transformers 4.5x)
    return "model." + name[len("language_model."):] if name.startswith("language_model.") else name


shutil.rmtree(root, ignore_errors=True)
os.makedirs(root)
t0 = time.time()
shards, cur, cur_bytes, total = [], {}, 0, 0
for name in shapes:
    t = synthetic_param(name, shapes, 0, "cpu").to(torch.bfloat16).contiguous()
    nb = t.numel() * 2
    if cur and cur_bytes + nb > 5 * 2 ** 30:
        shards.append(cur)
        cur, cur_bytes = {}, 0
    cur[hf4_key(name)] = t
    cur_bytes += nb
    total += nb
shards.append(cur)
weight_map = {}
for i, sh in enumerate(shards):
    fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
    save_file(sh, os.path.join(root, fn), metadata={"format": "pt"})
    weight_map.update({k: fn for k in sh})
json.dump(dict(metadata=dict(total_size=total), weight_map=weight_map), open(os.path.join(root, "model.safetensors.index.json"), "w"))
n_params = sum(t.numel() for sh in shards for t in sh.values())
del shards, cur
t_write = time.time() - t0

t0 = time.time()
arena = from_pretrained(root, cfg, dev, llm_fp8=fp8)
torch.cuda.synchronize()
t_load = time.time() - t0
t0 = time.time()
want = WeightArena(cfg, dev, llm_fp8=fp8).fill_tiled(seed=0)
torch.cuda.synchronize()
t_fill = time.time() - t0
def raw(t):
    return t.view(torch.uint8) if t.dtype == torch.uint8 else (t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16))


# entry by entry (the flat buffer has uninitialised alignment gaps between entries)
bad = [name for name in arena.offsets if not torch.equal(raw(arena.view(name)), raw(want.view(name)))]
equal = not bad and sorted(arena.offsets) == sorted(want.offsets)
print(json.dumps(dict(config=cfg.name, llm_fp8=fp8, parameters=n_params, checkpoint_bytes=total, shards=len(set(weight_map.values())),
                      tensors=len(weight_map), key_scheme="transformers 4.5x (model.layers.* / visual.* / lm_head.weight)",
                      write_s=round(t_write, 1), from_pretrained_s=round(t_load, 1), checkpoint_gb_per_s=round(total / t_load / 1e9, 2),
                      fill_tiled_s=round(t_fill, 1), arena_bytes=int(arena.flat.numel() * 2), arena_entries=len(arena.offsets), arena_equals_fill_tiled=equal, differing_entries=bad[:8])), flush=True)
shutil.rmtree(root, ignore_errors=True)
