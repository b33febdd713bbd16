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
    for name, shp in weight_shapes(livecc_7b(), decode_copies=False):
        k = 1
        for s in shp:
            k *= s
        n += k
    assert abs(n / 1e9 - 8.291) < 0.005          # 8.291 B parameters (7B: lm_head untied)
    # the arena additionally holds the row-permuted decode copy of every q|k|v weight (decode pipeline v2): +0.462 B elements
    extra = sum(shp[0] * shp[1] for name, shp in weight_shapes(livecc_7b()) if name.endswith("qkv_w_dec"))
    assert extra == 28 * 4608 * 3584


def test_decode_copy_is_the_row_permuted_qkv_weight():
    from livecc_amd.weights import WeightArena, qkv_decode_row_permutation
    cfg = tiny()
    perm = qkv_decode_row_permutation(cfg)
    assert sorted(perm.tolist()) == list(range(cfg.qkv_dim))
    assert perm[:16].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 64, 65, 66, 67, 68, 69, 70, 71]      # one MFMA tile = 8 channels + partners
    assert perm[128 + 16:128 + 24].tolist() == [128 + 8 + i for i in range(8)]
    for arena in (WeightArena(cfg, "cpu").fill_tiled(2), WeightArena(cfg, "cpu").fill_random(2)):
        for l in range(cfg.num_hidden_layers):
            assert torch.equal(arena.logical(f"llm.{l}.qkv_w_dec"), arena.logical(f"llm.{l}.qkv_w")[perm])
    # fp8 arenas (round 3: the W8 GEMVs of decode pipeline v2) carry the decode copy too: the SAME e4m3 bytes and row scales, permuted
    a8 = WeightArena(cfg, "cpu", llm_fp8=True).fill_random(2)
    assert any(n.endswith("qkv_w_dec") for n in a8.names()) and any(n.endswith("qkv_w_dec.scale") for n in a8.names())
    for l in range(cfg.num_hidden_layers):
        assert torch.equal(a8.logical(f"llm.{l}.qkv_w_dec"), a8.logical(f"llm.{l}.qkv_w")[perm])
        assert torch.equal(a8.view(f"llm.{l}.qkv_w_dec.scale"), a8.view(f"llm.{l}.qkv_w.scale")[perm])


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


def test_fp8_quantise_pack_roundtrip_and_arena_layout():
    """OCP e4m3 row quantisation (the 72B weight path): error bound, PACKED8 pack/unpack, fp8 arena store/logical, and that a
    second quantisation of the dequantised weights (even after bf16 rounding) reproduces the same integers -- which is what
    lets the oracle run on `fake_quantize_llm_fp8` weights while the native arena is built from the same model."""
    from livecc_amd.config import tiny
    from livecc_amd.weights import (WeightArena, dequantize_fp8_rows, pack_weight_fp8, quantize_fp8_rows, unpack_weight_fp8)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g) * 0.05
    w[3] = 0.0                                            # an all-zero row must not divide by zero
    q, sc = quantize_fp8_rows(w)
    assert q.dtype == torch.uint8 and sc.dtype == torch.float32 and sc.shape == (64,)
    d = dequantize_fp8_rows(q, sc)
    assert torch.isfinite(d).all() and float(d[3].abs().max()) == 0.0
    amax = w.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert float(((d - w).abs() / amax).max()) <= 2.0 ** -4 + 1e-6     # e4m3: 3 mantissa bits -> half-ulp 2^-4 of the row max
    assert torch.equal(unpack_weight_fp8(pack_weight_fp8(q)), q)
    q2, sc2 = quantize_fp8_rows(d.to(torch.bfloat16))
    assert torch.equal(q2, q) and torch.allclose(sc2, sc, rtol=2.0 ** -7)
    # PACKED8: lane (g,row) of fragment (row tile rt, 64-k block kb) holds k = kb*64 + g*16 .. +15
    p = pack_weight_fp8(q).view(4, 4, 4, 16, 16)
    assert torch.equal(p[2, 1, 3, 5], q[2 * 16 + 5, 64 + 48:64 + 64])
    cfg = tiny()
    arena = WeightArena(cfg, "cpu", llm_fp8=True)
    assert "llm.0.qkv_w.scale" in arena.names() and "lm_head.scale" in arena.names() and "vit.0.qkv_w.scale" not in arena.names()
    wq = torch.randn(cfg.qkv_dim, cfg.hidden_size, generator=g) * 0.03
    arena.store("llm.0.qkv_w", wq)
    assert arena.view("llm.0.qkv_w").dtype == torch.uint8 and arena.view("llm.0.qkv_w.scale").dtype == torch.float32
    qq, ss = quantize_fp8_rows(wq)
    assert torch.equal(arena.logical("llm.0.qkv_w"), dequantize_fp8_rows(qq, ss))
    bf16 = WeightArena(cfg, "cpu")
    assert arena.nbytes() < bf16.nbytes() and all(o[0] % 128 == 0 for o in arena.offsets.values())


def test_fp8_quantiser_is_the_ieee_rule_spelled_out():
    """`quantize_fp8_rows` is written in device-independent operations (fp64 division, frexp / ldexp / round-half-even) so that the CPU
    oracle and the GPU arena hold the same bytes (round 5: the devices' own float -> float8 casts disagreed on 0.15 % of a 72B matrix, which
    put the 80-layer 72B parity at 1.12-1.18 x the reference's error instead of 1.0).  On the CPU the spelled-out rule must equal the plain
    one (fp32 division + torch's e4m3 cast: round-to-nearest-even, saturating): every e4m3 value, every midpoint between neighbours, values
    just beside both, subnormals, +-448; and a seeded matrix byte for byte."""
    from livecc_amd.weights import FP8_MAX, dequantize_fp8_rows, quantize_fp8_rows
    vals = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    vals = vals[~vals.isnan()].sort().values
    mids = (vals[:-1] + vals[1:]) / 2
    probe = torch.cat([vals, mids, mids * (1 + 2.0 ** -20), mids * (1 - 2.0 ** -20), vals * 1.03, torch.tensor([0.0, 2.0 ** -10, 2.0 ** -11, 3 * 2.0 ** -11])]).clamp(-FP8_MAX, FP8_MAX)
    # one row per probe value next to a +-448 entry, so the row scale is exactly 1 and x = the probe value itself
    w = torch.zeros(probe.numel(), 64)
    w[:, 0] = FP8_MAX
    w[:, 1] = probe
    q, sc = quantize_fp8_rows(w)
    assert torch.equal(sc, torch.ones_like(sc))
    want = probe.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q[:, 1], want), (probe[q[:, 1] != want][:8], q[:, 1][q[:, 1] != want][:8], want[q[:, 1] != want][:8])
    g = torch.Generator().manual_seed(5)
    wm = (torch.randn(512, 2048, generator=g) * 0.02).to(torch.bfloat16)
    q1, s1 = quantize_fp8_rows(wm)
    wf = wm.float()
    s0 = (wf.abs().amax(dim=1) / FP8_MAX).clamp_min(1e-12)
    q0 = (wf / s0[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(s1, s0) and torch.equal(q1, q0)
    q2, s2 = quantize_fp8_rows(dequantize_fp8_rows(q1, s1))                  # a dequantised checkpoint re-quantises to itself
    assert torch.equal(q2, q1) and torch.equal(s2, s1)
    qz, sz = quantize_fp8_rows(torch.zeros(16, 64))                          # all-zero rows: scale floor, zero bytes
    assert int(qz.sum()) == 0 and bool((sz > 0).all())


def test_vit_rope_copy_is_the_row_permuted_qkv_weight_in_rotation_pair_order():
    """`vit.{i}.qkv_w_rope` / `qkv_b_rope` (the q|k|v projection whose GEMM epilogue applies the 2-D RoPE, csrc/gemm.hip vit_qkv_epilogue):
    a permutation; inside q and k every 32 stored rows are 16 first-half channels followed by their partners 40 channels later IN THE SAME
    head, 4-row groups never straddle a head (the epilogue stores 4 channels per lane); V rows keep their order; excluded from the
    parameter count (`decode_copies=False`); every filler derives the copies from the natural weight."""
    from livecc_amd.weights import WeightArena, vit_qkv_rope_row_permutation
    for cfg in (tiny(), livecc_7b()):
        E = cfg.vit_embed_dim
        perm = vit_qkv_rope_row_permutation(cfg)
        assert sorted(perm.tolist()) == list(range(3 * E))
        assert torch.equal(perm[2 * E:], torch.arange(2 * E, 3 * E))
        for which in range(2):
            blk = perm[which * E:(which + 1) * E].view(-1, 2, 16)
            first, partner = blk[:, 0], blk[:, 1]
            assert torch.equal(partner, first + 40)
            assert ((first - which * E) % 80 < 40).all() and (first // 80 == partner // 80).all()
            quads = first.reshape(-1, 4)
            assert (quads[:, 3] - quads[:, 0] == 3).all() and (quads[:, 0] // 80 == quads[:, 3] // 80).all()
        assert not any(n.endswith("_rope") for n, _ in weight_shapes(cfg, decode_copies=False))
        assert sum(n.endswith("_rope") for n, _ in weight_shapes(cfg)) == 2 * cfg.vit_depth
    cfg = tiny()
    perm = vit_qkv_rope_row_permutation(cfg)
    for arena in (WeightArena(cfg, "cpu").fill_random(seed=3), WeightArena(cfg, "cpu").fill_tiled(seed=3)):
        for l in range(cfg.vit_depth):
            assert torch.equal(arena.logical(f"vit.{l}.qkv_w_rope"), arena.logical(f"vit.{l}.qkv_w")[perm])
            assert torch.equal(arena.view(f"vit.{l}.qkv_b_rope"), arena.view(f"vit.{l}.qkv_b")[perm])


def test_rotation_in_the_stored_row_order_equals_hf_rotate_half():
    """What the q|k|v GEMM epilogue does with the `qkv_w_rope` order, restated on the CPU: take the projection output in STORED column order,
    rotate every (column t, column t + 16) pair of a 32-column group with the cos / sin of channel c = f % 40 (f = group * 16 + t), store the
    results at natural columns head * 80 + c and + 40 -- and compare with HF's apply_rotary_pos_emb_vision formula
    q * cos + rotate_half(q) * sin on the natural layout (modeling_qwen2_vl.py:225-248)."""
    from livecc_amd.weights import vit_qkv_rope_row_permutation
    cfg = tiny()
    E, heads = cfg.vit_embed_dim, cfg.vit_num_heads
    perm = vit_qkv_rope_row_permutation(cfg)
    g = torch.Generator().manual_seed(0)
    P = 12
    x = torch.randn(P, 3 * E, generator=g)                      # natural-order Linear output
    ang = torch.randn(P, 40, generator=g)
    cos, sin = ang.cos(), ang.sin()
    # HF: cos / sin = cat(freqs, freqs) over the 80 channels of a head; rotate_half(q) = cat(-q[40:], q[:40])
    qk = x[:, :2 * E].view(P, 2 * heads, 80)
    c80, s80 = torch.cat([cos, cos], -1)[:, None, :], torch.cat([sin, sin], -1)[:, None, :]
    want = (qk * c80 + torch.cat([-qk[..., 40:], qk[..., :40]], -1) * s80).reshape(P, 2 * E)
    # the epilogue's view
    xs = x[:, perm]
    got = torch.full((P, 2 * E), float("nan"))
    for n in range(0, 2 * E, 32):
        which = n // E
        for t in range(16):
            f = ((n - which * E) // 32) * 16 + t
            h, c = f // 40, f % 40
            x1, x2 = xs[:, n + t], xs[:, n + 16 + t]
            got[:, which * E + h * 80 + c] = x1 * cos[:, c] - x2 * sin[:, c]
            got[:, which * E + h * 80 + c + 40] = x2 * cos[:, c] + x1 * sin[:, c]
    assert torch.equal(got, want)
    assert torch.equal(xs[:, 2 * E:], x[:, 2 * E:])             # V columns are not permuted
