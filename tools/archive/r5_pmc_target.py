#!/usr/bin/env python
"""Target of the round-5 SQ counter passes (tools/gpu_call.sh r5b): the MFMA-bound kernels of one GPU's share of configs[2] -- the LLM
prefill gate/up GEMM at M = 3088, the prefill attention (attn_gqa32_kernel) at 8 x 386 rows x 6.2k keys, and the vision tower on 8 chunks
(attn_vit32_kernel + the tower GEMMs + layernorm_kernel) -- a few launches each."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import _lib, ops, protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402
from livecc_amd.modeling import LiveCCForConditionalGeneration  # noqa: E402
from livecc_amd.weights import WeightArena  # noqa: E402

dev = torch.device("cuda:0")
H, I = 3584, 18944
ws = [ops.pack_weight((torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
x = torch.randn(3088, H, device=dev).to(torch.bfloat16)
for i in range(5):
    ops.linear(x, ws[i % 2], None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))
del ws
# prefill attention, 8 streams x 386 new rows over 6200 cached keys (tools/bench_attn.py chunk_8streams, variant 3, no key split)
lib = _lib.load()
Hq, Hkv = 28, 4
kv = ops.KvArena(8, 1, Hkv, 8192, dev)
kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
ts, tq, tn, tp, row = [], [], [], [], 0
for slot in range(8):
    for o in range(0, 386, 32):
        ts.append(slot); tq.append(row + o); tn.append(min(32, 386 - o)); tp.append(6200 + o)
    row += 386
a, b, c, d = [torch.tensor(v, dtype=torch.int32, device=dev) for v in (ts, tq, tn, tp)]
q = (torch.randn(row, Hq * 128, device=dev) * 0.7).to(torch.bfloat16)
out = torch.empty_like(q)
wso = torch.empty(row * Hq * 128, dtype=torch.float32, device=dev)
wsm = torch.empty(row * Hq * 2, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    _lib.check(lib.lcc_attn_prefill_bf16(q.data_ptr(), out.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), kv.ptrs.data_ptr(), kv.lay, 0,
                                         a.numel(), Hq, 32, 1, row, wso.data_ptr(), wsm.data_ptr(), st), "attn")
torch.cuda.synchronize()
cfg = get_config("livecc-7b")
native = LiveCCForConditionalGeneration(cfg, WeightArena(cfg, dev).fill_random(seed=5), dev, max_streams=1, max_kv_len=1024, max_new_rows=512,
                                        max_patches=16384, max_history=16)
clips = [dict(frames=torch.from_numpy(protocol.synth_frames(2, 392, 728, seed=21 + i, layout="TCHW")).to(dev), layout="TCHW") for i in range(8)]
for _ in range(2):
    native.engine.vit_encode(clips)
torch.cuda.synchronize()
print("ok")
