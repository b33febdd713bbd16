"""-m gpu: RCCL itself on the MI355X box.  The product's only collective is ONE broadcast of the flat weight arena at start-up
(livecc_amd/distributed.py); the multi-rank logic is covered on CPU over gloo (tests/test_weights_dist.py,
tests/test_bench_launch.py).  A gpurun box has ONE GPU, so a multi-rank RCCL run cannot happen here (RCCL rejects two ranks on one
device); what can be checked on hardware before the driver's 8-GPU run: the `nccl` (= RCCL) backend of this torch build initialises on
this box, and `broadcast_weights` / the bench's reductions run through it on device memory with a single-rank communicator."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from livecc_amd import distributed as D
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
x = torch.arange(1 << 24, dtype=torch.float32, device="cuda:0")       # 64 MB
ref = x.clone()
dist.broadcast(x, src=0)
dist.all_reduce(x, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(x, ref)
t = torch.zeros(3, device="cuda:0"); t[0] = 7
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t[0]) == 7.0
info = D.pin_to_gpu_numa_node(0)
print("RCCL_OK", torch.cuda.get_device_name(0), info)
dist.destroy_process_group()
""" % ROOT


def test_rccl_backend_initialises_and_moves_device_memory_on_this_box():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
