#!/usr/bin/env python
"""Soak run of the live-paced server leg (bench.live2fps) at one stream count: N live 2-fps streams for `video_s` seconds of video, the KV
history of every stream growing the whole time (60 s = 58 chunks = ~26k cached keys per stream), latency p99 per 10-s window.

    python tools/r5_live_soak.py [streams=48] [video_s=60]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.weights import WeightArena  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
video_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
args = bench.parse([])
dev = torch.device("cuda:0")
cfg = get_config("livecc-7b")
arena = WeightArena(cfg, dev).fill_tiled(seed=0)
r = bench.live2fps(cfg, arena, dev, args, protocol, ladder=(N,), video_s=video_s)
print(json.dumps(r["ladder"][0]), flush=True)
print(json.dumps(dict(gpu_mem_peak_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 1))), flush=True)
