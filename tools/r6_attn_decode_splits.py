#!/usr/bin/env python
"""Round-6 probe: duration of the decode attention launch and of its combine launch as a function of the key split count, one stream at
L = 6,552 keys (204 key tiles), K / V rotated over 20 caches (268 MB > the Infinity Cache).  Run under `rocprofv3 --kernel-trace`; the
launches go out in a fixed order (REPS per split count), tools/r6_attn_decode_splits.py --reduce <trace.csv> prints the medians."""
import csv
import json
import os
import sys

SPLITS = (13, 26, 51, 68, 102, 128)
REPS = 24
if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    rows = []
    for r in csv.DictReader(open(sys.argv[2])):
        kn = r["Kernel_Name"]
        if "attn_decode_kernel" in kn or "attn_decode_combine_kernel" in kn:
            rows.append((int(r["Start_Timestamp"]), "combine" if "combine" in kn else "attn", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    a = [d for _, k, d in rows if k == "attn"]
    c = [d for _, k, d in rows if k == "combine"]
    assert len(a) == len(c) == len(SPLITS) * REPS, (len(a), len(c))
    for i, ns in enumerate(SPLITS):
        xa, xc = sorted(a[i * REPS + 4:(i + 1) * REPS]), sorted(c[i * REPS + 4:(i + 1) * REPS])
        print(json.dumps(dict(nsplit=ns, tiles_per_wave=-(-204 // ns), attn_us_median=round(xa[len(xa) // 2], 2), attn_us_min=round(xa[0], 2),
                              combine_us_median=round(xc[len(xc) // 2], 2), sum_us=round(xa[len(xa) // 2] + xc[len(xc) // 2], 2))))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
NS, L, HQ, HKV = 20, 6552, 28, 4
kv = ops.KvArena(NS, 1, HKV, 8192, dev)
kv.buf.copy_((torch.randn(NS, kv.per_slot, device=dev) * 0.5).to(torch.bfloat16))
kv_len = torch.full((NS,), L, dtype=torch.int32, device=dev)
q = torch.randn(1, HQ * 128, device=dev).to(torch.bfloat16)
filler = torch.empty(64 << 20, dtype=torch.float32, device=dev)          # a 256-MB write between launches: the cache the next launch reads is cold
for ns in SPLITS:
    for i in range(REPS):
        filler.add_(1.0)
        ops.attn_decode(q, kv, 0, torch.tensor([i % NS], dtype=torch.int32, device=dev), kv_len, HQ, ns)
torch.cuda.synchronize()
print("ok")
