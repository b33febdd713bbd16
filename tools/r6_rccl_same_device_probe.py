#!/usr/bin/env python
"""Probe (round 6): can RCCL run a WORLD-2 communicator with both ranks on the ONE GPU of a gpurun box, so that
`distributed.broadcast_weights` runs over RCCL with world > 1 at least once before the driver's 8-GPU run?  Two ranks, same device, the
chunked broadcast of a 256-MB buffer + the bench's reductions; prints one JSON line: {"ok": bool, "detail": ...}.  Bounded by its own timeouts."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from livecc_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=int(os.environ["RANK"]), world_size=2, device_id=torch.device("cuda:0"))
x = torch.arange(1 << 26, dtype=torch.float32, device="cuda:0") if dist.get_rank() == 0 else torch.zeros(1 << 26, dtype=torch.float32, device="cuda:0")
s = D.broadcast_weights(x, src=0, chunk_bytes=64 << 20)
ok = bool(torch.equal(x, torch.arange(1 << 26, dtype=torch.float32, device="cuda:0")))
g = D.gather_floats(float(dist.get_rank() + 1), "cuda:0")
m = D.max_over_ranks(float(dist.get_rank()), "cuda:0")
print("RANK", dist.get_rank(), "OK" if (ok and g == [1.0, 2.0] and m == 1.0) else "BAD", round(s, 4), flush=True)
D.shutdown("cuda:0")
""" % ROOT

env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
procs = [subprocess.Popen([sys.executable, "-c", CODE], env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
         for r in range(2)]
t0, outs = time.time(), []
for p in procs:
    try:
        o, _ = p.communicate(timeout=max(5, 100 - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        p.kill()
        o, _ = p.communicate()
        o = (o or "") + "\nTIMEOUT"
    outs.append(o)
ok = all(p.returncode == 0 for p in procs) and all("OK" in o for o in outs)
print(json.dumps(dict(ok=ok, seconds=round(time.time() - t0, 1), rank0=outs[0][-600:], rank1=outs[1][-600:])))
