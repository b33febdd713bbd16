"""ORACLE / TEST INFRASTRUCTURE: generates tests/golden/*.npz from the HF CPU oracle (run in the build container):

    python oracle/make_golden.py

The reference holds no golden vectors for this path, so these fixtures pin the oracle's behaviour at the time
of generation (transformers 5.15.0, torch 2.10.0): a 3-turn streaming replay at `tiny` shapes with seeded weights,
frames and prompt ids.  Stored: turn ids, grids, generated tokens, raw logits (fp32 oracle and bf16 oracle), ViT features,
first-turn position ids / rope_delta (stream_tiny.npz); and the PER-STAGE tensors of the first turn -- PatchEmbed, every vision
block, merger, embeddings, every decoder layer after attention / after the MLP, bf16 run + fp32 truth per stage (stages_tiny.npz).  tests/test_oracle.py re-runs the oracle against them (CPU), tests/test_gpu_golden.py
checks the HIP path against them (GPU box, no oracle import needed for the comparison itself).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import tiny  # noqa: E402
from oracle import hf_oracle as O  # noqa: E402

SEED_W, SEED_IN, INIT_SCALE = 0, 1234, 2.0
FRAMES, H, W, MAX_NEW, PENALTY = 10, 56, 84, 8, 1.05


def generate():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = tiny()
    hf16 = O.build_hf_model(cfg, torch.bfloat16, SEED_W, INIT_SCALE)
    hf32 = O.build_hf_model(cfg, torch.float32, SEED_W, INIT_SCALE)
    frames = torch.from_numpy(protocol.synth_frames(FRAMES, H, W, seed=SEED_IN, layout="TCHW"))
    res16 = O.replay_stream(hf16, cfg, frames, protocol.TurnBuilder(cfg, seed=SEED_IN), MAX_NEW, PENALTY, max_turns=3)
    # fp32 oracle teacher-forced along the bf16 oracle's tokens (identical histories)
    s32 = O.OracleStream(hf32, cfg)
    out = dict(n_turns=np.int64(len(res16)))
    for ti, (r, (a, b)) in enumerate(zip(res16, protocol.split_clip(FRAMES))):
        pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
        r32 = s32.turn(r["turn_ids"], pv, grid, max_new_tokens=MAX_NEW, repetition_penalty=PENALTY, teacher_tokens=r["new_tokens"])
        out[f"t{ti}_ids"] = np.asarray(r["turn_ids"], dtype=np.int64)
        out[f"t{ti}_grid"] = np.asarray(grid, dtype=np.int64)
        out[f"t{ti}_tokens"] = np.asarray(r["new_tokens"], dtype=np.int64)
        out[f"t{ti}_logits_bf16"] = torch.stack(r["logits"]).numpy().astype(np.float32)
        out[f"t{ti}_logits_fp32"] = torch.stack(r32["logits"]).numpy().astype(np.float32)
        out[f"t{ti}_vit_bf16"] = O.vit_forward_ref(hf16, pv, grid).float().numpy()
        out[f"t{ti}_vit_fp32"] = O.vit_forward_ref(hf32, pv, grid).float().numpy()
    pos, delta = protocol.rope_index_first_turn(res16[0]["turn_ids"], [res16[0]["grid"]], cfg)
    out["t0_pos3"], out["rope_delta"] = pos, np.int64(delta)
    out["meta"] = np.asarray([SEED_W, SEED_IN, FRAMES, H, W, MAX_NEW], dtype=np.int64)
    return out


STAGE_KEYS16 = ("patch_embed", "vit_in", "vit_out", "merger_in", "vit_merged", "embeds", "llm_in", "llm_mid", "llm_out", "final_in")
STAGE_KEYS32 = ("patch_embed", "vit_out", "vit_merged", "llm_mid", "llm_out")


def generate_stages():
    """Per-stage tensors of the first turn: the bf16 oracle's PatchEmbed / block-k / merger / embeddings /
    hidden-per-layer tensors (stored as exact bf16 bit patterns) and the fp32 oracle's output of every stage on the bf16 run's INPUT of
    that stage (oracle/layer_probe.py: teacher forcing per layer) + both runs' last-position logits."""
    from oracle import layer_probe as P
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = tiny()
    hf16 = O.build_hf_model(cfg, torch.bfloat16, SEED_W, INIT_SCALE)
    hf32 = O.build_hf_model(cfg, torch.float32, SEED_W, INIT_SCALE)
    frames = torch.from_numpy(protocol.synth_frames(FRAMES, H, W, seed=SEED_IN, layout="TCHW"))[:6]
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=SEED_IN).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    a16 = P.probe(hf16, cfg, ids, pv, grid)
    t32 = P.probe(hf32, cfg, ids, pv, grid, P.inputs_of(a16))
    out = dict(ids=np.asarray(ids, dtype=np.int64), grid=np.asarray(grid, dtype=np.int64),
               meta=np.asarray([SEED_W, SEED_IN, 6, H, W], dtype=np.int64),
               logits_bf16=a16["logits"].numpy(), logits_fp32=t32["logits"].numpy())
    for k in STAGE_KEYS16:
        out["bf16_" + k] = a16[k].contiguous().view(torch.int16).numpy()
    for k in STAGE_KEYS32:
        out["fp32_" + k] = t32[k].float().numpy()
    return out


def load_stages(path):
    """(bf16-run dict, fp32-run dict, raw npz) with torch tensors under the keys of oracle.layer_probe.probe."""
    g = np.load(path)
    a16 = {k: torch.from_numpy(g["bf16_" + k].copy()).view(torch.bfloat16) for k in STAGE_KEYS16}
    t32 = {k: torch.from_numpy(g["fp32_" + k].copy()) for k in STAGE_KEYS32}
    a16["logits"], t32["logits"] = torch.from_numpy(g["logits_bf16"].copy()), torch.from_numpy(g["logits_fp32"].copy())
    return a16, t32, g


if __name__ == "__main__":
    st = generate_stages()
    path = os.path.join(ROOT, "tests", "golden", "stages_tiny.npz")
    np.savez_compressed(path, **st)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    if "--stages-only" in sys.argv:
        sys.exit(0)
    o = generate()
    path = os.path.join(ROOT, "tests", "golden", "stream_tiny.npz")
    np.savez_compressed(path, **o)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for ti in range(int(o["n_turns"])):
        print(ti, o[f"t{ti}_grid"].tolist(), o[f"t{ti}_tokens"].tolist())

