"""Multi-GPU: one process per GPU, streams sharded data-parallel, weights broadcast ONCE over RCCL/xGMI.

The reference's data-parallel inference (ref evaluation/livesports3kcc/distributed_generate_livecc.py:38-122) spawns
N processes, every one re-reads the full checkpoint from disk (line 46) and takes the strided shard idxs[i::N]
(49-50); there is no collective anywhere.  Here rank 0 owns the checkpoint (or generates the random weights) and the
flat weight arena is broadcast with `torch.distributed.broadcast` (backend "nccl" = RCCL) in a few large chunks;
the decode / prefill path has no collective at all (streams are independent: own KV, ids, rope_delta).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): the broadcast is per-link bound, ~0.11 s ideal for the 7B arena.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> tuple:
    """(rank, local_rank, world) from torchrun's env; initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def numa_cpus_of_gpu(local_rank: int):
    """CPU ids of the NUMA node the GPU `local_rank` hangs off (PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/node<N>/cpulist), or None when the platform does not say."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{int(getattr(p, 'pci_domain_id', 0)):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return sorted(cpus) or None
    except Exception:
        return None


def pin_to_gpu_numa_node(local_rank: int) -> dict:
    """One process per GPU: keep the rank's host threads (Python orchestration, pinned meta ring, launches) on the CPUs of its GPU's
    NUMA node -- on an 8-GPU node the ranks otherwise migrate across sockets and every launch crosses the inter-socket link.
    Best effort (returns what it did); LCC_NO_NUMA_PIN=1 disables it."""
    if os.environ.get("LCC_NO_NUMA_PIN") == "1":
        return dict(pinned=False, reason="LCC_NO_NUMA_PIN=1")
    cpus = numa_cpus_of_gpu(local_rank) if torch.cuda.is_available() else None
    if not cpus:
        return dict(pinned=False, reason="NUMA node of the GPU unknown")
    try:
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return dict(pinned=False, reason="no overlap with the allowed CPU set")
        # every thread that already exists (HIP runtime, torch intra-op pool), not only the caller: sched_setaffinity(0) pins one thread
        tids = [0]
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")] or [0]
        except OSError:
            pass
        n = 0
        for tid in tids:
            try:
                os.sched_setaffinity(tid, allowed)
                n += 1
            except OSError:
                pass                      # a thread that just exited
        return dict(pinned=n > 0, cpus=len(allowed), first_cpu=allowed[0], threads=n)
    except Exception as e:                   # never fatal
        return dict(pinned=False, reason=repr(e))


def shard_streams(stream_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Static strided sharding, stream s -> GPU s % world (ref distributed_generate_livecc.py:49-50: idxs[i::N])."""
    return [s for s in stream_ids if s % world == rank]


def broadcast_weights(flat: torch.Tensor, src: int = 0, chunk_bytes: int = 1 << 30) -> float:
    """Broadcast the flat weight arena in chunks of ~1 GiB (large enough to run at link rate, small enough to pipeline
    along the ring).  Returns seconds (max over ranks is taken by the caller if needed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    import time
    n = flat.numel()
    step = max(1, chunk_bytes // flat.element_size())
    if flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    t0 = time.perf_counter()
    for o in range(0, n, step):
        dist.broadcast(flat[o:min(n, o + step)], src=src)
    if flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    return time.perf_counter() - t0


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value: float, device=None) -> List[float]:
    """[value of rank 0, ..., value of rank W-1] on every rank (end-of-run counters only: never in the data path)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    w = dist.get_world_size()
    t = torch.zeros(w, dtype=torch.float64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def barrier(device=None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and torch.device(device).type == "cuda":
            dist.barrier(device_ids=[torch.device(device).index])
        else:
            dist.barrier()


def shutdown(device=None) -> None:
    """Orderly end of a multi-rank run: every rank waits for every other one, then the process group is destroyed -- a rank that exits while a
    peer still holds its sockets / RCCL communicator open makes that peer abort at interpreter exit (seen with 8 gloo ranks: SIGABRT in one of
    them, once in a few runs)."""
    if dist.is_initialized():
        try:
            barrier(device)
        finally:
            dist.destroy_process_group()
