"""Data-parallel offline commentary generation: the driver of
ref evaluation/livesports3kcc/distributed_generate_livecc.py with the reference's prompt construction, static strided
sharding, per-item JSON files (resumable), merge to JSONL -- but ONE process per GPU launched by torchrun, the checkpoint read
by rank 0 only and broadcast over RCCL/xGMI (`distributed.broadcast_weights`) instead of every worker re-reading it (ref :46).

The dataset (`datasets.load_dataset('stdKonjac/LiveSports-3K', ...)`) and video decoding are external: `records` is any
sequence of dicts with the reference's fields (`video`, `video_id`, `event_id`, `begin`, `end`, `event_title`, `preasr_text`):
`video` names a decoded video (`livecc_amd.video`: registered path or DecodedVideo) from which the [begin, end] range is
sampled and resized exactly as the reference does; alternatively `frames` -- an already sampled uint8 clip tensor or a
zero-argument callable returning one (decoded lazily on the owning rank).
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Callable, Optional, Sequence

COMMENTARY_PROMPT = ("You are an expert video commentator providing real-time, insightful, "
                     "and engaging commentary on visual content.\n")


def overall_prompt(title: Optional[str], preasr: Optional[str], simple_ctx: bool = False) -> str:
    """ref distributed_generate_livecc.py:69-84 (simple_ctx = `--not_instruct_model`)."""
    if simple_ctx:
        title = "" if preasr else title                    # title or preasr
        return f"{title}\n{preasr}".strip()
    out = COMMENTARY_PROMPT
    if title:
        out += f"This is a video titled \"{title}\".\n"
    if preasr:
        out += f"Here is previous commentary of the video:\n\n{preasr}\n\n"
        out += "Please continue to comment the video."
    return out


def merge_commentary(responses) -> str:
    """ref :93-96: join the per-chunk texts, dropping the ' ...' fillers, and close with '...'."""
    return " ".join(cc.replace(" ...", "") for _, _, cc in responses if cc).strip() + "..."


def generate_shard(records: Sequence[dict], infer, save_dir: str, rank: int, world: int, simple_ctx: bool = False,
                   repetition_penalty: float = 1.15, max_new_tokens: int = 32, progress: Optional[Callable[[int], None]] = None) -> int:
    """This rank's items idxs[rank::world] (ref :49-50), one `{idx}.json` each, existing files skipped (resume, ref :57-59)."""
    os.makedirs(save_dir, exist_ok=True)
    done = 0
    for idx in range(rank, len(records), world):
        path = os.path.join(save_dir, f"{idx}.json")
        if os.path.exists(path):
            continue
        rec = records[idx]
        # ref :61-68, 84-89: `video` names the source (a registered path / DecodedVideo: frame selection for [begin, end] and the
        # resize run inside live_cc_once_for_evaluation); `frames` = an already sampled + resized clip (or a callable making one)
        video = rec.get("video")
        if video is None:
            video = rec["frames"]() if callable(rec["frames"]) else rec["frames"]
        responses = infer.live_cc_once_for_evaluation(query=overall_prompt(rec.get("event_title"), rec.get("preasr_text"), simple_ctx),
                                                      video=video, video_start=rec.get("begin"), video_end=rec.get("end"),
                                                      max_new_tokens=max_new_tokens, repetition_penalty=repetition_penalty,
                                                      **({"frames_layout": rec["frames_layout"]} if "frames_layout" in rec else {}))
        tmp = path + ".tmp"
        with open(tmp, "w") as wf:
            json.dump({"video_id": rec.get("video_id"), "event_id": rec.get("event_id"), "begin": rec.get("begin"),
                       "end": rec.get("end"), "pred": merge_commentary(responses)}, wf)
        os.replace(tmp, path)                             # atomic: a killed worker never leaves a half-written item
        done += 1
        if progress:
            progress(idx)
    return done


def merge_jsonl(save_dir: str) -> str:
    """ref :124-130: jsons -> `<save_dir>.jsonl`, then remove the directory."""
    out = save_dir.rstrip("/") + ".jsonl"
    with open(out, "w") as wf:
        for name in sorted(os.listdir(save_dir), key=lambda n: int(n.split(".")[0]) if n.split(".")[0].isdigit() else 1 << 60):
            if name.endswith(".json"):
                wf.write(json.dumps(json.load(open(os.path.join(save_dir, name)))) + "\n")
    shutil.rmtree(save_dir)
    return out


def distributed_generate(records: Sequence[dict], model_path: str, output_dir: str, simple_ctx: bool = False,
                         repetition_penalty: float = 1.15, max_new_tokens: int = 32, **model_kw) -> Optional[str]:
    """torchrun entry (`--nproc-per-node N`): rank 0 reads the checkpoint, one RCCL broadcast, every rank serves
    records[rank::world]; rank 0 merges after a barrier.  Returns the JSONL path on rank 0."""
    import torch

    from . import distributed as D
    from .config import get_config
    from .infer import LiveCCDemoInfer
    from .modeling import LiveCCForConditionalGeneration, read_generation_config
    from .weights import WeightArena, from_pretrained

    rank, local, world = D.init_from_env()
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    cfg = get_config(model_path)
    arena = from_pretrained(model_path, cfg, dev) if rank == 0 else WeightArena(cfg, dev)
    D.broadcast_weights(arena.flat, src=0)
    gen_cfg, eos_ids = read_generation_config(model_path)      # as from_pretrained: the checkpoint's sampling defaults and EOS ids
    if eos_ids is not None:
        model_kw.setdefault("eos_token_ids", eos_ids)
    model = LiveCCForConditionalGeneration(cfg, arena, dev, **model_kw)
    model.generation_config = gen_cfg
    infer = LiveCCDemoInfer(model_path=model_path, model=model)
    save_dir = os.path.join(output_dir, os.path.basename(os.path.normpath(model_path)))
    generate_shard(records, infer, save_dir, rank, world, simple_ctx, repetition_penalty, max_new_tokens)
    D.barrier(dev)
    return merge_jsonl(save_dir) if rank == 0 else None
