#!/usr/bin/env python
"""Device-wide hand-off cost on MI355X: grid barrier inside one launch vs a kernel boundary (lcc_debug_bench_grid_barrier): the
single-counter barrier of round 2 and the XCD-hierarchical barrier of csrc/grid_sync.h (MI355X_MICROARCH.md "barrier-xcd").
Every wait in the barrier kernels is bounded, so a mis-sized grid reports `fails` instead of hanging."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
scratch = torch.randint(0, 2 ** 31 - 1, ((4096 + 1024 * 128 + 4096 * 1024 * 16) // 4,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((1, "kernel_boundary"), (0, "grid_barrier_single_counter"), (2, "grid_barrier_single_counter_under_16KB_reads"),
                   (3, "grid_barrier_xcd_hierarchical"), (4, "grid_barrier_xcd_hierarchical_under_16KB_reads")):
    for blocks in (64, 256, 512):
        us, fails = C.c_float(), C.c_int()
        rc = lib.lcc_debug_bench_grid_barrier(mode, blocks, 2000, scratch.data_ptr(), scratch.numel() * 4, C.byref(us), C.byref(fails), st)
        print(json.dumps(dict(mode=name, blocks=blocks, us_per_handoff=round(us.value, 3), fails=fails.value, rc=rc)))
