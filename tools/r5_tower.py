#!/usr/bin/env python
"""Round-5 A/B target: wall time of the LiveCC-7B vision tower (hipEvents, median of 7 calls) on 8 and on 1 co-scheduled 2-frame 392x728
chunks, plus the LLM prefill of 8 x 386 rows against ~6.2k cached keys is left to bench.py.  argv[1] = tag."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.modeling import LiveCCForConditionalGeneration  # noqa: E402
from livecc_amd.weights import WeightArena  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
cfg = get_config("livecc-7b")
native = LiveCCForConditionalGeneration(cfg, WeightArena(cfg, dev).fill_random(seed=5), dev, max_streams=1, max_kv_len=1024, max_new_rows=512,
                                        max_patches=16384, max_history=16)
for streams in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (8, 1)):      # argv[2] = "1" or "8,1": which chunk counts to time
    clips = [dict(frames=torch.from_numpy(protocol.synth_frames(2, 392, 728, seed=21 + i, layout="TCHW")).to(dev), layout="TCHW") for i in range(streams)]
    for _ in range(2):
        native.engine.vit_encode(clips)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.engine.vit_encode(clips)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(json.dumps(dict(tag=tag, tower_streams=streams, ms_median=round(ts[3], 3), ms_min=round(ts[0], 3))), flush=True)
