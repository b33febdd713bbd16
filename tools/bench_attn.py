#!/usr/bin/env python
"""LLM prefill attention micro-benchmark on MI355X (hipEvents, tile tables prebuilt, launches back to back through the C-ABI):
variant 2 (attn_shared_kernel, 16x16x32 MFMAs, 16- or 32-row tiles) vs variant 3 (attn_gqa32_kernel, 32x32x16 MFMAs, 32-row tiles) at
the shapes of the streaming chunk (386 rows x ~6.5k keys, 1 and 8 streams), of the first turn (1131 rows) and of the one-shot
prefill pieces (4096 rows against 8k / 24k keys).  One JSON line per case: microseconds per launch (+ combine) and PFLOP/s
(4 * rows * mean visible keys * 128 * heads flops)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
ZEROS = "--zeros" in sys.argv
dev = torch.device("cuda:0")
Hq, Hkv = 28, 4


def tables(segs, tile_rows):
    ts, tq, tn, tp = [], [], [], []
    row = 0
    for slot, n_new, past in segs:
        for o in range(0, n_new, tile_rows):
            ts.append(slot); tq.append(row + o); tn.append(min(tile_rows, n_new - o)); tp.append(past + o)
        row += n_new
    return [torch.tensor(x, dtype=torch.int32, device=dev) for x in (ts, tq, tn, tp)], row


def run(kv, segs, variant, tile_rows, nsplit, iters=20):
    (a, b, c, d), rows = tables(segs, tile_rows)
    torch.manual_seed(1234 + rows)         # the same q for every variant of a case: max_abs_diff_vs_first compares like with like
    q = (torch.randn(rows, Hq * 128, device=dev) * 0.7).to(torch.bfloat16)
    if ZEROS:
        q.zero_()
    out = torch.empty_like(q)
    ws_o = torch.empty(rows * Hq * max(nsplit, 1) * 128, dtype=torch.float32, device=dev)
    ws_ml = torch.empty(rows * Hq * max(nsplit, 1) * 2, dtype=torch.float32, device=dev)
    ops.set_attn_variant(variant)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.lcc_attn_prefill_bf16(q.data_ptr(), out.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(),
                                             kv.ptrs.data_ptr(), kv.lay, 0, a.numel(), Hq, tile_rows, nsplit, rows, ws_o.data_ptr(),
                                             ws_ml.data_ptr(), st), "attn")
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
    ops.set_attn_variant(ops.ATTN_DEFAULT_VARIANT)
    return us, flops / (us * 1e-6) / 1e15, out


cases = [("chunk_1stream", [(0, 386, 6200)]), ("chunk_8streams", [(s, 386, 6200) for s in range(8)]),
         ("first_turn_1stream", [(0, 1131, 0)]), ("first_turn_8streams", [(s, 1131, 0) for s in range(8)]),
         ("oneshot_piece_4096_at_8k", [(0, 4096, 8192)]), ("oneshot_piece_4096_at_20k", [(0, 4096, 20480)]),
         ("chunk_1stream_long_cache", [(0, 100, 30000)])]
quick = "--quick" in sys.argv
kv = ops.KvArena(8, 1, Hkv, 24576 + 8192, dev)
kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
if ZEROS:      # --zeros: the same launches on zero-filled q / K / V (is the kernel clock-throttled on real data like the GEMMs are?)
    kv.buf.zero_()
if "--only32" in sys.argv:
    cases = [c for c in cases if c[0] in ("chunk_1stream", "chunk_8streams", "first_turn_8streams", "oneshot_piece_4096_at_8k")]
for name, segs in cases:
    rows = sum(n for _, n, _ in segs)
    ref = None
    TALL = ops.attn_tile_rows(Hq, Hkv)     # the engine's tile height under variant 3 (36 rows at 28 / 4 heads)
    for variant, tr, nss in (((3, 32, (1, 4)), (3, TALL, (1, 4, 5))) if "--only32" in sys.argv else
                             ((2, 16, (1, 4, 8)), (2, 32, (1, 4)), (3, 32, (1, 2, 3, 4, 6, 8)), (3, TALL, (1, 2, 3, 4, 5, 6, 8)))):
        for ns in nss:
            if ns > 1 and rows > 1024 * 8:
                continue
            if quick and ns not in (1, 4):
                continue
            us, pf, out = run(kv, segs, variant, tr, ns)
            if ref is None:
                ref = out.float()
            err = float((out.float() - ref).abs().max())
            print(json.dumps(dict(operands="zeros" if ZEROS else "random", case=name, rows=rows, variant=variant, tile_rows=tr, nsplit=ns, us=round(us, 1), pflops=round(pf, 3),
                                  max_abs_diff_vs_first=round(err, 5))), flush=True)
