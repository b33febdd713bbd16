#!/usr/bin/env python
"""sha256 of the 8-wave GEMM's outputs at the engine's shapes (fixed seeds): two schedules of the same kernel (LCC_GEMM_SCHED=1 / 6) must
print identical digests -- they differ only in WHEN the LDS-DMA pieces are issued, never in the accumulation order."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
if len(sys.argv) > 1:            # python tools/gemm_checksum.py <gemm variant>: e.g. 3 (gemm_big_kernel<256> everywhere) vs 11 (gemm_pp_kernel)
    ops.set_gemm_variant(int(sys.argv[1]))
g = torch.Generator(device="cpu").manual_seed(7)


def rnd(*shape, s=1.0):
    return (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).to(dev)


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


H, I = 3584, 18944
cases = []
x = rnd(3088, H); w = ops.pack_weight(rnd(2 * I, H, s=0.02))
cases.append(("gate_up_M3088_swiglu", ops.linear(x, w, None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))))
x = rnd(1131, H)
cases.append(("gate_up_M1131_swiglu", ops.linear(x, w, None, ops.EPI_SWIGLU, packed_shape=(2 * I, H))))
del w
xi = rnd(386, I); wd = ops.pack_weight(rnd(H, I, s=0.02))
cases.append(("down_M386_splitk4", ops.linear_partial(xi, wd, 4, packed_shape=(H, I))))
xi = rnd(3088, I); res = rnd(3088, H)
cases.append(("down_M3088_residual", ops.linear(xi, wd, None, ops.EPI_RESIDUAL, res, packed_shape=(H, I))))
del wd
x = rnd(386, H); wq = ops.pack_weight(rnd(4608, H, s=0.02))
cases.append(("qkv_M386_splitk3", ops.linear_partial(x, wq, 3, packed_shape=(4608, H))))
xv = rnd(1456, 1280); wv = ops.pack_weight(rnd(3840, 1280, s=0.05)); bv = rnd(3840, s=0.1)
cases.append(("vit_qkv_P1456_bias", ops.linear(xv, wv, bv, ops.EPI_NONE, packed_shape=(3840, 1280))))
xv = rnd(11648, 1280)
cases.append(("vit_qkv_P11648_bias", ops.linear(xv, wv, bv, ops.EPI_NONE, packed_shape=(3840, 1280))))
torch.cuda.synchronize()
for name, out in cases:
    print(name, digest(out))
