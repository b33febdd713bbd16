"""ORACLE / TEST INFRASTRUCTURE: CPU baseline leg of bench.py (never imported by the product).

Times the reference's arithmetic -- HF `Qwen2VLForConditionalGeneration.generate` (ref demo/infer.py:165-172), bf16,
SDPA, on the host cores -- on ONE streaming turn: ViT over 2 frames + prompt prefill + N greedy tokens, at the given
shapes.  Run as a subprocess; prints one JSON object per line as it progresses (build, every generated token) so that
the parent can report a measured or a partially-extrapolated rate inside a fixed wall-clock budget.

Weights are filled by tiling a 4M-element random bf16 block (CPU throughput is data independent; initialising 8.3 B
parameters with a CPU RNG takes minutes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit(**kw):
    print(json.dumps(kw), flush=True)


T_IMPORT = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="livecc-7b")
    ap.add_argument("--height", type=int, default=392)
    ap.add_argument("--width", type=int, default=728)
    ap.add_argument("--max-new-tokens", type=int, default=16)
    a = ap.parse_args()
    import torch
    from transformers import LogitsProcessorList, Qwen2VLForConditionalGeneration
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from oracle import hf_oracle as O
    cfg = get_config(a.config)
    cores = os.cpu_count() or 1
    emit(event="imported", seconds=round(time.perf_counter() - T_IMPORT, 2))
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    emit(event="start", cores=cores, threads=torch.get_num_threads(), cpu=cpu_model, config=cfg.name)
    t0 = time.perf_counter()
    with torch.device("meta"):
        m = Qwen2VLForConditionalGeneration._from_config(cfg.to_hf(), dtype=torch.bfloat16)
    emit(event="meta_model", seconds=round(time.perf_counter() - t0, 2))
    m = m.to_empty(device="cpu")
    emit(event="allocated", seconds=round(time.perf_counter() - t0, 2))
    blk = (torch.randn(1 << 22) * 0.02).to(torch.bfloat16)
    with torch.no_grad():
        for name, p in m.named_parameters():
            flat = p.data.view(-1)
            if p.dim() == 1:
                flat.fill_(1.0 if name.endswith("weight") else 0.0)
                continue
            for o in range(0, flat.numel(), blk.numel()):
                k = min(blk.numel(), flat.numel() - o)
                flat[o:o + k].copy_(blk[:k])
        for name, buf in m.named_buffers():
            if "inv_freq" in name:
                dim = buf.numel() * 2
                theta = 10000.0 if "visual" in name else cfg.rope_theta
                buf.copy_(1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim)))
    m.eval()
    m.generation_config.do_sample = False
    m.generation_config.top_k = m.generation_config.top_p = m.generation_config.temperature = None
    emit(event="built", seconds=round(time.perf_counter() - t0, 2))

    frames = torch.from_numpy(protocol.synth_frames(2, a.height, a.width, seed=1234, layout="TCHW"))
    pv, grid = O.patchify_normalize_ref(frames, cfg)
    ids = protocol.TurnBuilder(cfg, seed=1234).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    emit(event="inputs", patches=int(pv.shape[0]), prompt_tokens=int(len(ids)))

    class Tick:
        def __init__(self):
            self.t0 = None

        def __call__(self, input_ids, scores):
            emit(event="token", i=int(input_ids.shape[1] - len(ids)), t=round(time.perf_counter() - self.t0, 4))
            return scores

    tick = Tick()
    input_ids = torch.as_tensor(ids).view(1, -1)
    kw = dict(pixel_values_videos=pv, video_grid_thw=torch.as_tensor([list(grid)]),
              mm_token_type_ids=torch.as_tensor(protocol.mm_token_type_ids(input_ids.numpy(), cfg)))
    tick.t0 = time.perf_counter()
    with torch.inference_mode():
        m.generate(input_ids=input_ids, do_sample=False, repetition_penalty=1.05, logits_processor=LogitsProcessorList([tick]),
                   max_new_tokens=a.max_new_tokens, min_new_tokens=a.max_new_tokens, pad_token_id=cfg.eos_token_id,
                   eos_token_id=cfg.eos_token_id, **kw)
    emit(event="done", t=round(time.perf_counter() - tick.t0, 4))


if __name__ == "__main__":
    main()
