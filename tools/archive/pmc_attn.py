#!/usr/bin/env python
"""rocprofv3 PMC target: ONLY the LLM prefill attention of a multi-stream streaming turn (8 streams x 386 new rows against 6000
cached keys each, 28 q / 4 kv heads) a few times -- to read SQ counters of attn_shared_kernel per dispatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, S, L, Hq, Hkv = 8, 386, 6000, 28, 4
kv = ops.KvArena(B, 1, Hkv, 6528, dev)
kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
q = (torch.randn(B * S, Hq * 128, device=dev) * 0.7).to(torch.bfloat16)
segs = [(b, S, L) for b in range(B)]
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tile_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if len(sys.argv) > 3:
    ops.set_attn_variant(int(sys.argv[3]))      # 2 = attn_shared_kernel (16x16x32 MFMAs), 3 = attn_gqa32_kernel (32x32x16, attn32.hip)
for i in range(4):
    out = ops.attn_prefill(q, kv, 0, segs, Hq, tile_rows=tile_rows, nsplit=nsplit)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10):
    out = ops.attn_prefill(q, kv, 0, segs, Hq, tile_rows=tile_rows, nsplit=nsplit)
e1.record()
e1.synchronize()
print("ok", float(out.float().abs().mean()), "us per call (incl. host tile-table setup)", round(e0.elapsed_time(e1) * 100, 1), "nsplit", nsplit, "tile_rows", tile_rows)
