"""-m gpu: `torch.ops.livecc_amd.*` (csrc/torch_ops.cpp) against the ctypes binding of the same C-ABI symbols -- the registered ops
forward to the SAME kernels, so every result must be bit-identical -- plus what the registration is for: dispatcher visibility
(profiler ranges), TORCH_CHECK errors instead of UB, the caller's current stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dev, scale, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


@pytest.fixture()
def both(monkeypatch):
    """(run through torch.ops, run through ctypes): the same livecc_amd.ops wrapper under the two bindings."""
    from livecc_amd import torch_ops
    assert torch_ops.try_load() is not None, "liblivecc_torch_ops.so missing on the GPU box (python -m livecc_amd.build)"

    def via_ctypes(fn, *a, **kw):
        with monkeypatch.context() as m:
            m.setattr(torch_ops, "op", lambda name: None)
            return fn(*a, **kw)

    def via_torch(fn, name, *a, **kw):
        before = torch_ops.CALLS[name]
        out = fn(*a, **kw)
        assert torch_ops.CALLS[name] == before + 1, f"{name} did not go through torch.ops"
        return out
    return via_torch, via_ctypes


def test_norms_swiglu_linear_are_bit_identical_under_both_bindings(dev, both):
    from livecc_amd import ops
    vt, vc = both
    x, w, b = _rand((37, 3584), dev, 2.0, 1), _rand((3584,), dev, 1.0, 2), _rand((3584,), dev, 1.0, 3)
    assert torch.equal(vt(ops.rmsnorm, "rmsnorm", x, w, 1e-6), vc(ops.rmsnorm, x, w, 1e-6))
    assert torch.equal(vt(ops.layernorm, "layernorm", x, w, b, 1e-6), vc(ops.layernorm, x, w, b, 1e-6))
    g, u = _rand((37, 1024), dev, 2.0, 4), _rand((37, 1024), dev, 2.0, 5)
    assert torch.equal(vt(ops.swiglu, "swiglu", g, u), vc(ops.swiglu, g, u))
    W, bias = _rand((512, 3584), dev, 0.05, 6), _rand((512,), dev, 1.0, 7)
    for M in (1, 37, 386):
        xm = _rand((M, 3584), dev, 1.0, 8 + M)
        assert torch.equal(vt(ops.linear, "linear", xm, W, bias), vc(ops.linear, xm, W, bias))
        res = _rand((M, 512), dev, 1.0, 9)
        assert torch.equal(vt(ops.linear, "linear", xm, W, bias, ops.EPI_RESIDUAL, res), vc(ops.linear, xm, W, bias, ops.EPI_RESIDUAL, res))
        wp = ops.pack_weight(W)
        assert torch.equal(vt(ops.linear, "linear", xm, wp, bias, ops.EPI_SWIGLU, None, (512, 3584)), vc(ops.linear, xm, wp, bias, ops.EPI_SWIGLU, None, (512, 3584)))


def test_rope_append_and_attention_are_bit_identical_under_both_bindings(dev, both):
    from livecc_amd import ops
    vt, vc = both
    Hq, Hkv, S = 14, 2, 75
    qkv = _rand((S, (Hq + 2 * Hkv) * 128), dev, 1.0, 20)
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    inv = (1.0 / (1e6 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))).to(dev)
    c, s = ops.mrope_table(torch.stack([pos, pos, pos]).contiguous(), inv, [16, 24, 24])
    zeros = torch.zeros(S, dtype=torch.int32, device=dev)
    outs = []
    for run in ("torch", "ctypes"):
        kv = ops.KvArena(1, 1, Hkv, 128, dev)
        if run == "torch":
            q = vt(ops.rope_kv_append, "rope_kv_append", qkv, c, s, zeros, pos, kv, 0, Hq)
            o = vt(ops.attn_prefill, "attn_prefill", q, kv, 0, [(0, S, 0)], Hq)
            kl = torch.tensor([S - 1], dtype=torch.int32, device=dev)
            d = vt(ops.attn_decode, "attn_decode", q[-1:].contiguous(), kv, 0, torch.zeros(1, dtype=torch.int32, device=dev), kl, Hq, 2)
        else:
            q = vc(ops.rope_kv_append, qkv, c, s, zeros, pos, kv, 0, Hq)
            o = vc(ops.attn_prefill, q, kv, 0, [(0, S, 0)], Hq)
            kl = torch.tensor([S - 1], dtype=torch.int32, device=dev)
            d = vc(ops.attn_decode, q[-1:].contiguous(), kv, 0, torch.zeros(1, dtype=torch.int32, device=dev), kl, Hq, 2)
        outs.append((q, o, d, kv.buf.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][3].abs().sum() > 0, "the registered op really appended to the arena it was handed (Tensor(a!) kv_buf)"


def test_front_end_and_sampler_are_bit_identical_under_both_bindings(dev, both):
    from livecc_amd import ops, resize
    from livecc_amd.engine import fused_mean_std
    vt, vc = both
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (4, 90, 120, 3), dtype=torch.uint8, generator=g).to(dev)
    a = vt(resize.resize_bicubic_aa, "resize_bicubic_aa", frames, 56, 84, "THWC")
    assert torch.equal(a, vc(resize.resize_bicubic_aa, frames, 56, 84, "THWC"))
    m, s = fused_mean_std()
    assert torch.equal(vt(ops.patchify_norm, "patchify_norm", a, "TCHW", m, s), vc(ops.patchify_norm, a, "TCHW", m, s))
    V, B = 2048, 3
    logits = (torch.randn(B, V, generator=g) * 2).to(torch.bfloat16).to(dev)
    seen = torch.from_numpy(np.random.RandomState(0).randint(0, 2 ** 31, size=(B, V // 32)).astype(np.int32)).to(dev)
    slots = torch.arange(B, dtype=torch.int32, device=dev)
    t1, s1 = vt(ops.sample_greedy, "sample_greedy", logits, seen, slots, 1.05, 5, 0.01, 7, True, want_scores=True)
    t2, s2 = vc(ops.sample_greedy, logits, seen, slots, 1.05, 5, 0.01, 7, True, want_scores=True)
    assert torch.equal(t1, t2) and torch.equal(s1, s2)


def test_the_ops_are_visible_to_the_dispatcher(dev):
    """What ctypes-on-data_ptr() cannot give: a profiler range per op, TORCH_CHECK errors, the caller's current stream."""
    from livecc_amd import torch_ops
    ns = torch_ops.load()
    x, w = _rand((16, 256), dev, 1.0, 1), _rand((256,), dev, 1.0, 2)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        ns.rmsnorm(x, w, 1e-6)
    assert any("livecc_amd::rmsnorm" in e.key for e in prof.key_averages()), [e.key for e in prof.key_averages()][:8]
    with pytest.raises(RuntimeError, match="expected"):
        ns.rmsnorm(x.float(), w, 1e-6)                      # dtype: a TORCH_CHECK, not UB
    with pytest.raises(RuntimeError, match="contiguous"):
        ns.rmsnorm(x.t(), w, 1e-6)
    with pytest.raises(RuntimeError, match="weight has"):
        ns.rmsnorm(x, w[:128].contiguous(), 1e-6)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        y = ns.rmsnorm(x, w, 1e-6)                          # launched on the CURRENT stream (c10::hip::getCurrentHIPStream)
    side.synchronize()
    assert torch.equal(y, ns.rmsnorm(x, w, 1e-6))
