"""Weight arena layout (CPU) and the multi-process paths on gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch

from livecc_amd.config import livecc_7b, tiny
from livecc_amd.weights import WeightArena, interleave_gate_up, weight_shapes


def test_interleave_gate_up():
    g = torch.arange(64 * 4).view(64, 4).float()
    u = -g
    w = interleave_gate_up(g, u)
    assert w.shape == (128, 4)
    assert torch.equal(w[0:16], g[0:16]) and torch.equal(w[16:32], u[0:16]) and torch.equal(w[32:48], g[16:32])


def test_pack_weight_index_formula():
    """packed[((n//16*K32 + k//32)*4 + (k%32)//8)*128 + (n%16)*8 + k%8] == W[n,k]  (include/livecc_amd.h)"""
    from livecc_amd.weights import pack_weight
    N, K = 48, 96
    w = torch.arange(N * K, dtype=torch.float32).view(N, K)
    p = pack_weight(w).reshape(-1)
    K32 = K // 32
    for n, k in [(0, 0), (5, 7), (17, 33), (47, 95), (31, 64), (16, 8)]:
        idx = ((n // 16 * K32 + k // 32) * 4 + (k % 32) // 8) * 128 + (n % 16) * 8 + k % 8
        assert p[idx] == w[n, k]


def test_weight_shapes_cover_7b_param_count():
    n = 0
    for name, shp in weight_shapes(livecc_7b()):
        k = 1
        for s in shp:
            k *= s
        n += k
    assert abs(n / 1e9 - 8.291) < 0.005          # 8.291 B parameters (7B: lm_head untied)


def test_arena_from_hf_state_dict_cpu():
    from oracle import hf_oracle as O
    from livecc_amd.weights import from_hf_model
    cfg = tiny()
    hf = O.build_hf_model(cfg, torch.bfloat16, seed=0)
    a = from_hf_model(hf, cfg, "cpu")
    sd = hf.state_dict()
    q = sd["model.language_model.layers.1.self_attn.q_proj.weight"]
    assert torch.equal(a.logical("llm.1.qkv_w")[:q.shape[0]], q)
    assert a.is_packed("llm.1.qkv_w") and not a.is_packed("embed") and not a.is_packed("vit.patch_embed")
    gate, up = sd["model.language_model.layers.0.mlp.gate_proj.weight"], sd["model.language_model.layers.0.mlp.up_proj.weight"]
    gu = a.logical("llm.0.gate_up_w")
    assert torch.equal(gu[32:48], gate[16:32]) and torch.equal(gu[48:64], up[16:32])
    pe = sd["model.visual.patch_embed.proj.weight"]
    assert torch.equal(a.view("vit.patch_embed"), pe.reshape(pe.shape[0], -1))
    for name in a.names():
        assert a.view(name).data_ptr() % 16 == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from livecc_amd import distributed as D
    r, l, w = D.init_from_env("gloo")
    cfg = tiny()
    arena = WeightArena(cfg, "cpu")
    if r == 0:
        torch.manual_seed(7)
        arena.flat.copy_(torch.randn(arena.flat.numel()).to(torch.bfloat16))
    else:
        arena.flat.zero_()
    D.broadcast_weights(arena.flat, src=0, chunk_bytes=1 << 16)     # several chunks
    torch.manual_seed(7)
    expect = torch.randn(arena.flat.numel()).to(torch.bfloat16)
    ok = torch.equal(arena.flat, expect)
    mine = D.shard_streams(range(7), r, w)
    tot = D.sum_over_ranks(float(len(mine)))
    mx = D.max_over_ranks(float(r + 1))
    D.barrier()
    q.put((r, ok, mine, tot, mx))
    torch.distributed.destroy_process_group()


def test_weight_broadcast_and_stream_sharding_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "every rank must hold rank 0's weights after the broadcast"
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]     # stream s -> rank s % world
    assert res[0][3] == res[1][3] == 7.0 and res[0][4] == res[1][4] == 2.0
