#!/usr/bin/env python
"""Round-6 probe: tile height x key-split count of the LLM prefill attention (32x32x16 kernel) for 2 / 4 / 8 co-scheduled 386-row chunks
against 6.2k / 12k cached keys -- the grids on which "rounds of blocks on 256 CUs" decides (profiles/r06/attn_tall_tiles_splits_probe.jsonl).
Two repetitions per case (the second one runs at steady clocks); one JSON line per (case, tile_rows, nsplit)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
Hq, Hkv = 28, 4


def run(kv, segs, tile_rows, nsplit, iters=20):
    ts, tq, tn, tp, row = [], [], [], [], 0
    for slot, n_new, past in segs:
        for o in range(0, n_new, tile_rows):
            ts.append(slot); tq.append(row + o); tn.append(min(tile_rows, n_new - o)); tp.append(past + o)
        row += n_new
    a, b, c, d = (torch.tensor(x, dtype=torch.int32, device=dev) for x in (ts, tq, tn, tp))
    torch.manual_seed(1234 + row)
    q = (torch.randn(row, Hq * 128, device=dev) * 0.7).to(torch.bfloat16)
    out = torch.empty_like(q)
    ws_o = torch.empty(row * Hq * max(nsplit, 1) * 128, dtype=torch.float32, device=dev)
    ws_ml = torch.empty(row * Hq * max(nsplit, 1) * 2, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.lcc_attn_prefill_bf16(q.data_ptr(), out.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), kv.ptrs.data_ptr(), kv.lay, 0,
                                             a.numel(), Hq, tile_rows, nsplit, row, ws_o.data_ptr(), ws_ml.data_ptr(), st), "attn")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = sum(4.0 * n * (past + (n + 1) / 2.0) * 128 * Hq for _, n, past in segs)
    return us, flops / (us * 1e-6) / 1e15


kv = ops.KvArena(8, 1, Hkv, 24576 + 8192, dev)
kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
tall = ops.attn_tile_rows(Hq, Hkv)
for name, segs in (("chunk_8streams", [(s, 386, 6200) for s in range(8)]), ("chunk_4streams", [(s, 386, 6200) for s in range(4)]),
                   ("chunk_8streams_12k", [(s, 386, 12000) for s in range(8)]), ("chunk_2streams", [(s, 386, 6200) for s in range(2)]),
                   ("chunk_3streams", [(s, 386, 6200) for s in range(3)])):
    for rep in range(2):
        for tr, nsplit in ((32, 1), (tall, 1), (tall, 2), (32, 2), (tall, 3), (tall, 4)) + (((32, 7), (tall, 8)) if len(segs) == 2 else ()):
            us, pf = run(kv, segs, tr, nsplit)
            print(json.dumps(dict(case=name, repetition=rep, tile_rows=tr, nsplit=nsplit, us=round(us, 1), pflops=round(pf, 3))), flush=True)
