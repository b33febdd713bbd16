#!/usr/bin/env python
"""Target of a rocprofv3 kernel trace of the vision tower alone (tools/gpu_call.sh vitprof): LiveCC-7B tower on N co-scheduled 2-frame
392x728 chunks, the q|k|v projection with RoPE + V transpose in its epilogue (default) and as separate launches, 5 tower calls each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops, protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.modeling import LiveCCForConditionalGeneration  # noqa: E402
from livecc_amd.weights import WeightArena  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = get_config("livecc-7b")
native = LiveCCForConditionalGeneration(cfg, WeightArena(cfg, dev).fill_random(seed=5), dev, max_streams=1, max_kv_len=1024, max_new_rows=512,
                                        max_patches=16384, max_history=16)
clips = [dict(frames=torch.from_numpy(protocol.synth_frames(2, 392, 728, seed=21 + i, layout="TCHW")).to(dev), layout="TCHW") for i in range(streams)]
for on in (True, False):
    ops.set_vit_fused_qkv(on)
    for _ in range(5):
        native.engine.vit_encode(clips)
    torch.cuda.synchronize()
print("ok")
