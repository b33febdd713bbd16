"""-m gpu: kernel-level parity of the decode pipeline v2 operators (csrc/decode_v2.hip) through the C-ABI, each against an fp32
torch restatement of the HF ops it fuses (HF's bf16 rounding points), at LiveCC-7B / 2B / 72B-like and small shapes, M = 1..4."""
import math

import numpy as np
import pytest
import torch

from tests.test_gpu_ops import _hf_mrope_ref, _rand, _ref_linear
from tests.util import assert_bf16_close, rb, record

pytestmark = pytest.mark.gpu


def _rmsnorm_ref(h, w, eps):
    """Q2VL:96-110: fp32 variance, bf16(h * rsqrt(var + eps)), bf16(weight * that)."""
    x = h.float()
    r = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return rb(w.float() * rb(x * r))


def _staged(Ms, NKs):
    """(M, N, K) triples the decode pipeline v2 serves: its GEMVs stage M normalised rows of K bf16 in LDS (M * K <= 16384).  A filtered
    parameter list instead of skip-by-construction."""
    return [(m, n, k) for (n, k) in NKs for m in Ms if m * k <= 16384]


@pytest.mark.parametrize("M,N,K", _staged([1, 2, 4], [(37888, 3584), (17920, 1536), (1024, 256), (4864, 896)]))
def test_norm_swiglu_gemv(dev, M, N, K):
    from livecc_amd import ops
    h, w, nw = _rand((M, K), dev, 2.0, 1), _rand((N, K), dev, 0.03, 2), (1.0 + 0.1 * _rand((K,), dev, 1.0, 3).float()).to(torch.bfloat16)
    got = ops.dgemv_norm_linear(ops.pack_weight(w), h, ops.tile_stats(h), nw, 1e-6, (N, K), swiglu=True)
    ref, atol = _ref_linear(_rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16), w, None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"dgemv_norm_swiglu[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)


@pytest.mark.parametrize("M", [1, 3])
@pytest.mark.parametrize("N,K", [(152064, 3584), (2048, 256), (8192, 896)])
def test_norm_linear_gemv_lm_head(dev, M, N, K):
    from livecc_amd import ops
    h, w, nw = _rand((M, K), dev, 2.0, 4), _rand((N, K), dev, 0.03, 5), (1.0 + 0.1 * _rand((K,), dev, 1.0, 6).float()).to(torch.bfloat16)
    got = ops.dgemv_norm_linear(ops.pack_weight(w), h, ops.tile_stats(h), nw, 1e-6, (N, K))
    ref, atol = _ref_linear(_rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16), w, None, with_atol=True)
    assert_bf16_close(got, ref, f"dgemv_norm_linear[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)


@pytest.fixture(params=[0, 2], ids=["blocks_of_8_waves", "blocks_of_16_waves"])
def resid_waves(request):
    """Both block shapes of lcc_dgemv_resid (the engine's default picks 16 waves for bf16 weights with K >= 8192)."""
    from livecc_amd import _lib
    old = _lib.load().lcc_debug_set_resid_waves(request.param)
    yield request.param
    _lib.load().lcc_debug_set_resid_waves(old)


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("N,K", [(3584, 18944), (3584, 3584), (1536, 8960), (256, 512), (896, 2432), (8192, 8192)])
def test_residual_gemv_and_tile_statistics(dev, M, N, K, resid_waves):
    """h += Linear(x): HF rounds the Linear output to bf16, then the residual sum to bf16; stats = per-16-channel sums of squares."""
    from livecc_amd import ops
    x, w, h0 = _rand((M, K), dev, 1.0, 7), _rand((N, K), dev, 0.03, 8), _rand((M, N), dev, 2.0, 9)
    h = h0.clone()
    stats = ops.dgemv_resid_(ops.pack_weight(w), x, h, (N, K))
    ref, atol = _ref_linear(x, w, None, 3, h0, with_atol=True)
    assert_bf16_close(h, ref, f"dgemv_resid[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)
    want = ops.tile_stats(h)                                   # statistics of the rows the kernel actually wrote
    assert torch.allclose(stats, want, rtol=1e-6, atol=1e-6), float((stats - want).abs().max())


@pytest.mark.parametrize("Hq,Hkv,K", [(28, 4, 3584), (12, 2, 1536), (2, 1, 256), (7, 1, 896)])
@pytest.mark.parametrize("M", [1, 2])
def test_qkv_gemv_with_rope_and_kv_append(dev, Hq, Hkv, K, M):
    """[RMSNorm] q|k|v Linear [bias + M-RoPE + KV append] in one launch against the separate kernels of the round-1 path on the
    logical (un-permuted) weight: RMSNorm -> GEMV -> rope_kv_append.  The two differ only by the fp32 summation order of the GEMV
    (8 waves inside a block vs inter-block split-K slabs), i.e. rare one-ulp flips of the Linear output that the rotation carries."""
    from livecc_amd import ops
    from livecc_amd.config import LiveCCConfig
    from livecc_amd.weights import qkv_decode_row_permutation
    D = 128
    N = (Hq + 2 * Hkv) * D
    cfg = LiveCCConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=K)
    h, w, b = _rand((M, K), dev, 2.0, 11), _rand((N, K), dev, 0.03, 12), _rand((N,), dev, 0.2, 13)
    nw = (1.0 + 0.1 * _rand((K,), dev, 1.0, 14).float()).to(torch.bfloat16)
    lens = [37, 100][:M]
    kv_a, kv_b = ops.KvArena(M, 2, Hkv, 256, dev), ops.KvArena(M, 2, Hkv, 256, dev)
    g = torch.Generator().manual_seed(3)
    kv_a.buf.copy_((torch.randn(kv_a.buf.shape, generator=g) * 0.5).to(torch.bfloat16))
    kv_b.buf.copy_(kv_a.buf)
    slots = torch.arange(M, dtype=torch.int32, device=dev).flip(0).contiguous()          # row m -> slot M-1-m
    kv_len = torch.zeros(M, dtype=torch.int32, device=dev)
    kv_len[slots.long()] = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos3 = torch.tensor([[l + 3 for l in lens]] * 3, dtype=torch.int32)
    _, _, inv = _hf_mrope_ref(pos3)
    c, s_ = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    layer = 1
    perm = qkv_decode_row_permutation(cfg).to(dev)
    q_got = ops.dgemv_qkv_rope(ops.pack_weight(w[perm].contiguous()), h, ops.tile_stats(h), nw, 1e-6, b, c, s_, slots, kv_len, kv_a, layer, Hq)
    # reference chain: RMSNorm (fp32 restatement) -> split-K GEMV slabs -> rope_kv_append (bit-exact vs HF's op sequence, test_gpu_ops)
    xn = _rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16)
    part = ops.linear_partial(xn, ops.pack_weight(w), ops.gemv_num_splits(N, K), packed_shape=(N, K))
    q_ref = ops.rope_kv_append(None, c, s_, slots, None, kv_b, layer, Hq, partial=part, bias=b, kv_len=kv_len)
    assert_bf16_close(q_got, q_ref, f"dgemv_qkv_rope.q[{Hq},{Hkv},{K},M{M}]", max_ulp=2.0, max_frac=2e-2, atol=2e-2)
    for m in range(M):
        sl, n = int(slots[m]), lens[m]
        assert_bf16_close(kv_a.k_view(sl, layer)[:, n], kv_b.k_view(sl, layer)[:, n], "dgemv_qkv_rope.k_row", max_ulp=2.0, max_frac=2e-2, atol=2e-2)
        assert_bf16_close(kv_a.v_view(sl, layer)[:, n], kv_b.v_view(sl, layer)[:, n], "dgemv_qkv_rope.v_row", max_ulp=1.0, max_frac=2e-2, atol=1e-2)
    # nothing but the new token's rows was touched
    mask = torch.ones_like(kv_a.buf, dtype=torch.bool)
    ka, kb = kv_a.buf.clone(), kv_b.buf.clone()
    for m in range(M):
        sl, n = int(slots[m]), lens[m]
        for arena in (kv_a, kv_b):
            arena.k_view(sl, layer)[:, n] = 0
            o = (layer * 2 + 1) * Hkv * 256 * 128
            arena.buf[sl, o:o + Hkv * 256 * 128].view(Hkv, 8, 128, 32)[:, n // 32, :, n % 32] = 0
    assert torch.equal(kv_a.buf, kv_b.buf), "the epilogue wrote outside the new token's K/V rows"


def test_decode_step_begin(dev):
    """seen-bitmap mark + embedding gather + per-tile statistics + M-RoPE row in one launch, against the separate round-1 ops."""
    from livecc_amd import _lib, ops
    V, K, B, n_slots = 4096, 896, 3, 5
    table = _rand((V, K), dev, 1.0, 21)
    slots = torch.tensor([4, 0, 2], dtype=torch.int32, device=dev)
    cur = torch.tensor([7, 0, 99, 0, 4000], dtype=torch.int32, device=dev)
    done = torch.tensor([0, 0, 1, 0, 0], dtype=torch.int32, device=dev)
    pos = torch.tensor([5, 0, 11, 0, 30000], dtype=torch.int32, device=dev)
    seen = torch.zeros(n_slots, V // 32, dtype=torch.int32, device=dev)
    inv = (1.0 / (1e6 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).to(dev)
    h = torch.empty(B, K, dtype=torch.bfloat16, device=dev)
    stats = torch.empty(B, K // 16, dtype=torch.float32, device=dev)
    cs, sn = torch.empty(B, 64, dtype=torch.bfloat16, device=dev), torch.empty(B, 64, dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.load().lcc_decode_step_begin(slots.data_ptr(), cur.data_ptr(), done.data_ptr(), seen.data_ptr(), V // 32, table.data_ptr(),
                                                 h.data_ptr(), stats.data_ptr(), K, pos.data_ptr(), inv.data_ptr(), cs.data_ptr(), sn.data_ptr(), B,
                                                 torch.cuda.current_stream(dev).cuda_stream), "lcc_decode_step_begin")
    toks = [4000, 7, 99]
    assert torch.equal(h, table[toks])
    assert torch.allclose(stats, ops.tile_stats(h), rtol=1e-6, atol=1e-6)
    p = torch.tensor([[30000, 5, 11]] * 3, dtype=torch.int32)
    c_ref, s_ref = ops.mrope_table(p.to(dev), inv, [16, 24, 24])
    assert torch.equal(cs, c_ref) and torch.equal(sn, s_ref)
    want = np.zeros((n_slots, V // 32), dtype=np.uint32)
    for sl, t in ((4, 4000), (0, 7)):                          # slot 2 is done: its token is not marked
        want[sl, t >> 5] |= np.uint32(1 << (t & 31))
    assert np.array_equal(seen.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize("Hq,Hkv,H,I", [(28, 4, 3584, 18944), (12, 2, 1536, 8960), (7, 1, 896, 2432)])
@pytest.mark.parametrize("M", [1, 2])
def test_chained_down_qkv_launch_is_bit_identical_to_the_two_launches(dev, request, Hq, Hkv, H, I, M):
    """Round 3: down_proj of layer l and q/k/v of layer l+1 as ONE launch (the q/k/v blocks prefetch their weights, then wait for the
    down_proj blocks' write-through residual rows: csrc/decode_v2.hip) must give bit-identical h, tile statistics, rotated q and KV
    rows to lcc_dgemv_resid followed by lcc_dgemv_qkv_rope -- repeated 20 times on the same monotonic counter with a fresh residual
    each time (a consumer that read a stale row would reproduce the previous iteration's values), error word never set."""
    from livecc_amd import ops
    from livecc_amd.config import LiveCCConfig
    from livecc_amd.weights import qkv_decode_row_permutation
    N = (Hq + 2 * Hkv) * 128
    cfg = LiveCCConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=H)
    w_down, w_qkv, b = _rand((H, I), dev, 0.03, 31), _rand((N, H), dev, 0.03, 32), _rand((N,), dev, 0.2, 33)
    nw = (1.0 + 0.1 * _rand((H,), dev, 1.0, 34).float()).to(torch.bfloat16)
    wd_p = ops.pack_weight(w_down)
    wq_p = ops.pack_weight(w_qkv[qkv_decode_row_permutation(cfg).to(dev)].contiguous())
    lens = [37, 100][:M]
    slots = torch.arange(M, dtype=torch.int32, device=dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos3 = torch.tensor([[l + 3 for l in lens]] * 3, dtype=torch.int32)
    _, _, inv = _hf_mrope_ref(pos3)
    c, s_ = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    before = 0
    # the chained kernel's producer part is the 8-wave down_proj; the stand-alone default for K >= 8192 is the 16-wave block shape, whose
    # K split across waves (fp32 summation order) differs -- compare like with like
    from livecc_amd import _lib
    old_mode = _lib.load().lcc_debug_set_resid_waves(0)
    request.addfinalizer(lambda: _lib.load().lcc_debug_set_resid_waves(old_mode))
    for it in range(20):
        x, h0 = _rand((M, I), dev, 1.0, 40 + it), _rand((M, H), dev, 2.0, 80 + it)
        kv_a, kv_b = ops.KvArena(M, 1, Hkv, 128, dev), ops.KvArena(M, 1, Hkv, 128, dev)
        # the two launches
        h_ref = h0.clone()
        st_ref = ops.dgemv_resid_(wd_p, x, h_ref, (H, I))
        q_ref = ops.dgemv_qkv_rope(wq_p, h_ref, st_ref, nw, 1e-6, b, c, s_, slots, kv_len, kv_b, 0, Hq)
        # the chained launch
        h = h0.clone()
        try:
            st, q = ops.dgemv_down_qkv_(wd_p, x, h, I, wq_p, nw, 1e-6, b, c, s_, slots, kv_len, kv_a, 0, Hq, counter, before, err)
        except RuntimeError as e:
            if "(-5)" in str(e) and it == 0:
                pytest.skip(f"the two grids ({H // 16} + {N // 16} blocks) do not fit this chip at once: {e}")
            raise
        before += H // 16
        assert int(err[0]) == 0, "a consumer block gave up waiting"
        assert int(counter[0]) == before
        assert torch.equal(h, h_ref) and torch.equal(st, st_ref), f"iteration {it}: residual stream / statistics differ"
        assert torch.equal(q, q_ref), f"iteration {it}: rotated q differs (stale residual row?)"
        assert torch.equal(kv_a.buf, kv_b.buf), f"iteration {it}: appended K/V rows differ"


# ---------------------------------------------------------------------------------------------------------------------
# fp8 (e4m3 + row scale) weights on the pipeline-v2 GEMVs (round 3): against the SAME kernels on the exactly dequantised bf16 weights
# ---------------------------------------------------------------------------------------------------------------------
def _w8(dev, N, K, seed):
    from livecc_amd import ops
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    w = _rand((N, K), dev, 0.03, seed).float()
    q, sc = quantize_fp8_rows(w)
    wd = dequantize_fp8_rows(q, sc)                       # fp32 = e4m3 value x scale
    w8, sc2 = ops.quantize_fp8(w)
    assert torch.equal(sc, sc2)
    return w8, sc.contiguous(), wd


@pytest.mark.parametrize("M,N,K", _staged([1, 2, 4], [(37888, 3584), (59136, 8192), (1024, 256)]))
def test_fp8_norm_swiglu_gemv(dev, M, N, K):
    from livecc_amd import ops
    h, nw = _rand((M, K), dev, 2.0, 1), (1.0 + 0.1 * _rand((K,), dev, 1.0, 3).float()).to(torch.bfloat16)
    w8, sc, wd = _w8(dev, N, K, 2)
    got = ops.dgemv_norm_linear(w8, h, ops.tile_stats(h), nw, 1e-6, (N, K), swiglu=True, wscale=sc)
    ref, atol = _ref_linear(_rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16), wd, None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"dgemv_w8_norm_swiglu[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K", [(152064, 3584), (2048, 256)])
def test_fp8_norm_linear_gemv_lm_head(dev, M, N, K):
    from livecc_amd import ops
    h, nw = _rand((M, K), dev, 2.0, 4), (1.0 + 0.1 * _rand((K,), dev, 1.0, 6).float()).to(torch.bfloat16)
    w8, sc, wd = _w8(dev, N, K, 5)
    got = ops.dgemv_norm_linear(w8, h, ops.tile_stats(h), nw, 1e-6, (N, K), wscale=sc)
    ref, atol = _ref_linear(_rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16), wd, None, with_atol=True)
    assert_bf16_close(got, ref, f"dgemv_w8_norm_linear[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K", [(3584, 18944), (3584, 3584), (8192, 29568), (256, 512)])      # fp8 fragments are 64 k wide: K % 64 == 0
def test_fp8_residual_gemv_and_tile_statistics(dev, M, N, K, resid_waves):
    from livecc_amd import ops
    x, h0 = _rand((M, K), dev, 1.0, 7), _rand((M, N), dev, 2.0, 9)
    w8, sc, wd = _w8(dev, N, K, 8)
    h = h0.clone()
    stats = ops.dgemv_resid_(w8, x, h, (N, K), wscale=sc)
    ref, atol = _ref_linear(x, wd, None, 3, h0, with_atol=True)
    assert_bf16_close(h, ref, f"dgemv_w8_resid[{M}x{N}x{K}]", max_ulp=1.0, max_frac=1e-2, atol=atol)
    assert torch.allclose(stats, ops.tile_stats(h), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("Hq,Hkv,K,M", [(hq, hkv, k, m) for (hq, hkv, k) in [(28, 4, 3584), (64, 8, 8192), (2, 1, 256)] for m in (1, 2)
                                         if m * k <= 16384])
def test_fp8_qkv_gemv_with_rope_and_kv_append(dev, Hq, Hkv, K, M):
    """fp8 decode copy (rows AND scales permuted) in one launch against the round-1 fp8 chain on the un-permuted weight: RMSNorm ->
    fp8 split-K GEMV slabs (lcc_gemm_w8_bf16) -> rope_kv_append.  Same e4m3 values and scales on both sides; they differ only by the
    fp32 summation order of the GEMV."""
    from livecc_amd import ops
    from livecc_amd.config import LiveCCConfig
    from livecc_amd.weights import pack_weight_fp8, quantize_fp8_rows, qkv_decode_row_permutation
    D = 128
    N = (Hq + 2 * Hkv) * D
    cfg = LiveCCConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=K)
    h, b = _rand((M, K), dev, 2.0, 11), _rand((N,), dev, 0.2, 13)
    w = _rand((N, K), dev, 0.03, 12).float()
    nw = (1.0 + 0.1 * _rand((K,), dev, 1.0, 14).float()).to(torch.bfloat16)
    perm = qkv_decode_row_permutation(cfg).to(dev)
    q, sc = quantize_fp8_rows(w)
    w8, w8_dec, sc_dec = pack_weight_fp8(q), pack_weight_fp8(q[perm].contiguous()), sc[perm].contiguous()
    lens = [37, 100][:M]
    kv_a, kv_b = ops.KvArena(M, 1, Hkv, 256, dev), ops.KvArena(M, 1, Hkv, 256, dev)
    slots = torch.arange(M, dtype=torch.int32, device=dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos3 = torch.tensor([[l + 3 for l in lens]] * 3, dtype=torch.int32)
    _, _, inv = _hf_mrope_ref(pos3)
    c, s_ = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    q_got = ops.dgemv_qkv_rope(w8_dec, h, ops.tile_stats(h), nw, 1e-6, b, c, s_, slots, kv_len, kv_a, 0, Hq, wscale=sc_dec)
    xn = _rmsnorm_ref(h, nw, 1e-6).to(torch.bfloat16)
    part = ops.linear_w8(xn, w8, sc.contiguous(), nsplit=min(8, ops.gemv_num_splits(N, K)))
    q_ref = ops.rope_kv_append(None, c, s_, slots, None, kv_b, 0, Hq, partial=part, bias=b, kv_len=kv_len)
    assert_bf16_close(q_got, q_ref, f"dgemv_w8_qkv_rope.q[{Hq},{Hkv},{K},M{M}]", max_ulp=2.0, max_frac=2e-2, atol=2e-2)
    assert_bf16_close(kv_a.buf, kv_b.buf, "dgemv_w8_qkv_rope.kv", max_ulp=2.0, max_frac=2e-2, atol=2e-2)
