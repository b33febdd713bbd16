"""Per-kernel parity tests (-m gpu): every HIP kernel, called through the C-ABI (livecc_amd.ops -> ctypes ->
liblivecc_amd.so), against an fp32 torch restatement of the HF op it replaces with HF's bf16 rounding points.

Tolerances (written per test): results that HF rounds once to bf16 must agree with the reference to <= 1 bf16 ulp
with only rare (<0.2%) 1-ulp flips (different fp32 summation order); chains of rounded ops get 2 ulp; attention
(P is rounded to bf16 before P.V, as in HF eager/flash kernels) is checked at 2% of the output scale.
"""
import math

import numpy as np
import pytest
import torch

from tests.util import assert_bf16_close, rb, record

pytestmark = pytest.mark.gpu


def _rand(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _ref_linear(x, w, bias=None, epi=0, residual=None, with_atol=False):
    """fp32 reference with HF's bf16 rounding points, plus the per-element absolute tolerance that separates
    'different fp32 summation order' from 'wrong':
      * the dot product itself: 32 * eps32 * sum_k |x_k w_k| (matters where the sum cancels to ~0);
      * chained epilogues: the rounded Linear output y may legitimately flip by one bf16 ulp, which moves the final
        value by <= ~1.2 ulp(y) even when the final value is much smaller than y (gelu of a negative y, residual
        cancellation, silu(g)*u)."""
    acc = x.float() @ w.float().t()
    noise = 32 * 2.0 ** -24 * (x.float().abs() @ w.float().abs().t())
    if bias is not None:
        acc = acc + bias.float()
    if epi == 4:  # swiglu over interleaved [16 gate | 16 up] column blocks
        M, N2 = acc.shape
        a = acc.view(M, N2 // 32, 2, 16)
        nz = noise.view(M, N2 // 32, 2, 16)
        g, u = rb(a[:, :, 0]), rb(a[:, :, 1])
        s = rb(g / (1.0 + torch.exp(-g)))
        out = rb(s * u).reshape(M, N2 // 2)
        atol = ((2.0 ** -7 * g.abs() * 1.2 + nz[:, :, 0]) * u.abs() + (2.0 ** -7 * u.abs() + nz[:, :, 1]) * s.abs()
                + 2.0 ** -8 * (s * u).abs()).reshape(M, N2 // 2)
        return (out, atol) if with_atol else out
    y = rb(acc)
    atol = noise
    if epi == 1:
        t = rb(1.702 * y)
        pre = y.abs()
        y = rb(y * rb(torch.sigmoid(t)))
        atol = noise + 2.0 ** -7 * pre * 2.0 + 2.0 ** -7 * y.abs()
    elif epi == 2:
        pre = y.abs()
        y = rb(torch.nn.functional.gelu(y))
        atol = noise + 2.0 ** -7 * pre * 2.0 + 2.0 ** -7 * y.abs()
    elif epi == 3:
        pre = y.abs()
        y = rb(y + residual.float())
        atol = noise + 2.0 ** -7 * pre * 1.5 + 2.0 ** -7 * y.abs()
    return (y, atol) if with_atol else y


def _layout_variants(variants):
    """(packed, variant) pairs that exist: the 8-wave kernels (variant >= 3) take packed weights only -- a filtered parameter list, not a
    skip, so that the tier's skip count only holds tests that did not run for a reason worth reading."""
    return [(packed, v) for v in variants for packed in (False, True) if packed or v < 3]


@pytest.mark.parametrize("M,N,K", [(200, 512, 256), (386, 1280, 1176), (130, 480, 160), (64, 1024, 640), (1456, 3840, 1280),
                                   (17, 256, 512), (300, 4608, 3584), (260, 272, 192), (100, 256, 64), (600, 768, 512)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("packed,variant", _layout_variants([0, 1, 2, 3, 4, 5, 6, 13, 14]))
def test_gemm_tiled(dev, M, N, K, epi, packed, variant):
    """variant 0: register-staged 2-stage kernel; 1: LDS-DMA (global_load_lds) 3-stage ring; 2: auto (default);
    3 / 4: the 8-wave 256x256 / 128x256 LDS-DMA kernel wherever it is eligible (packed W, K % 64 == 0), pinned fragment-read
    schedule; 5 / 6: the same with the compiler's schedule; 13: the 192x256 tile (round 4); 14: row tiles of variable height (256 / 272 / 288
    rows, gemm_vh_kernel, round 5) wherever they are legal (M = 260, 300: one tile of 17 / 19 -> two tiles; 386: 25 fragments -> not legal,
    256-row tiles; 600: two tiles of 19 fragments -> not legal; 1456: 91 fragments = 5 tiles of 18 / 19 -> not legal)."""
    from livecc_amd import ops
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2), _rand((N,), dev, 0.1, 3)
    res = _rand((M, N), dev, 1.0, 4) if epi == 3 else None
    ops.set_gemm_variant(variant)
    try:
        got = ops.linear(x, ops.pack_weight(w), b, epi, res, packed_shape=(N, K)) if packed else ops.linear(x, w, b, epi, res)
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref, atol = _ref_linear(x, w, b, epi, res, with_atol=True)
    assert_bf16_close(got, ref, f"gemm_tiled[{M}x{N}x{K},epi{epi}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M,I,K", [(100, 512, 256), (386, 2432, 896), (530, 400, 128)])
@pytest.mark.parametrize("packed,variant", _layout_variants([0, 1, 3, 4, 6, 13, 14]))
def test_gemm_tiled_swiglu(dev, M, I, K, packed, variant):
    from livecc_amd import ops
    x, w = _rand((M, K), dev, 1.0, 1), _rand((2 * I, K), dev, 0.05, 2)
    ops.set_gemm_variant(variant)
    try:
        got = ops.linear(x, ops.pack_weight(w), None, ops.EPI_SWIGLU, packed_shape=(2 * I, K)) if packed else ops.linear(x, w, None, ops.EPI_SWIGLU)
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref, atol = _ref_linear(x, w, None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"gemm_tiled_swiglu[{M}x{I}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M", [17, 113, 225, 337, 386, 448])
@pytest.mark.parametrize("N,K", [(160, 64), (352, 256), (4608, 3584)])
@pytest.mark.parametrize("epi", [0, 1, 3, 4])
@pytest.mark.parametrize("variant", [8], ids=["ring_of_2_tiles"])
def test_gemm_tall_kernel(dev, M, N, K, epi, variant):
    """Variant 8 forces the tall tile (one block row covers all of M <= 448, 448 x 160 tiles, 4 x 2 waves with 6 + 4 column
    tiles, 2-stage ring of whole k-tiles; the 4-stage ring of 32-k half tiles of round 3 was retired in round 5 as a measured
    loser): every live-row-tile path of the last M-wave (M = 337: one row tile, 386: four, 448: seven), M-waves without
    rows (M = 17, 113, 225), a ragged last column block (N = 352), bias / quick-GELU / residual / SwiGLU epilogues."""
    from livecc_amd import ops
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2), _rand((N,), dev, 0.1, 3)
    res = _rand((M, N), dev, 1.0, 4) if epi == 3 else None
    ops.set_gemm_variant(variant)
    try:
        before = ops.launch_counts()["gemm_tall"]
        got = ops.linear(x, ops.pack_weight(w), None if epi == 4 else b, epi, res, packed_shape=(N, K))
        assert ops.launch_counts()["gemm_tall"] == before + 1
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref, atol = _ref_linear(x, w, None if epi == 4 else b, epi, res, with_atol=True)
    assert_bf16_close(got, ref, f"gemm_tall[{M}x{N}x{K},epi{epi},v{variant}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M", [256 + 16, 272 + 7, 288, 2 * 256 + 1, 2 * 288 - 16, 3088, 1131])
@pytest.mark.parametrize("N,K,epi", [(512, 256, 0), (768, 192, 3), (3584, 1024, 4)])
def test_gemm_variable_height_tiles_are_bit_identical_to_the_256_row_tiles(dev, M, N, K, epi):
    """gemm_vh_kernel (round 5): F = ceil(M / 16) row fragments dealt out over floor(F / 16) row tiles of 16-18 fragments -- one tile of
    17 / 18 fragments, a partial last fragment (M = 279, 513), two tiles of 17 + 16 and 18 + 17, the benchmark's 8 x 386 = 3088 rows (12
    tiles, one of them 272 rows) and the first turn's 1131 rows (4 tiles of 288 / 272) -- with bias, residual and SwiGLU epilogues: the
    accumulation order per output element is gemm_big_kernel<256>'s, so the outputs are bit-identical to variant 3, and the kernel really
    ran (launch counter)."""
    from livecc_amd import ops
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2), _rand((N,), dev, 0.1, 3)
    res = _rand((M, N), dev, 1.0, 4) if epi == 3 else None
    wp = ops.pack_weight(w)
    try:
        ops.set_gemm_variant(3)
        ref = ops.linear(x, wp, None if epi == 4 else b, epi, res, packed_shape=(N, K))
        ops.set_gemm_variant(14)
        before = ops.launch_counts()["gemm_vh"]
        got = ops.linear(x, wp, None if epi == 4 else b, epi, res, packed_shape=(N, K))
        assert ops.launch_counts()["gemm_vh"] == before + 1
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {got.numel()} outputs differ from the 256-row tiles"
    exp, atol = _ref_linear(x, w, None if epi == 4 else b, epi, res, with_atol=True)
    assert_bf16_close(got, exp, f"gemm_vh[{M}x{N}x{K},epi{epi}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M,N,K,S", [(386, 3584, 18944, 6), (386, 4608, 3584, 4), (386, 3584, 3584, 6), (400, 3600, 1024, 2), (130, 512, 512, 2),
                                     (272, 1024, 1024, 4), (144, 256, 4096, 8), (391, 1536, 8960, 5)])
def test_gemm_small_variable_height_tiles_give_the_slabs_of_the_128_row_tiles(dev, M, N, K, S):
    """gemm_vh_kernel, small class (round 5): split-K slabs on row tiles of 8 / 9 fragments (M = 386: 9 + 8 + 8 instead of four 128-row
    tiles whose last holds 2 rows).  Same k ranges per split and the same accumulation order per element as gemm_big_kernel<128>: every fp32
    slab is bit-identical to variant 4's; one tile of 9 (M = 130, 144), two tiles of 9 + 8 (M = 272), a ragged column tile (N = 3600),
    uneven k-tile counts per split (56 / 6, 140 / 5), the 2B shapes (1536 x 8960); and the kernel really ran (launch counter)."""
    from livecc_amd import ops
    x, w = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2)
    wp = ops.pack_weight(w)
    try:
        ops.set_gemm_variant(4)
        ref = ops.linear_partial(x, wp, S, packed_shape=(N, K))
        ops.set_gemm_variant(15)
        before = ops.launch_counts()["gemm_vh"]
        got = ops.linear_partial(x, wp, S, packed_shape=(N, K))
        assert ops.launch_counts()["gemm_vh"] == before + 1
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {got.numel()} slab entries differ from the 128-row tiles"
    full = got.sum(dim=0)
    exp = x.float() @ w.float().t()
    assert float((full - exp).abs().max()) <= 2e-5 * float(exp.abs().max()) * math.sqrt(K / 256) + 1e-4


def test_gemm_auto_choice_takes_the_small_variable_height_tiles_for_one_chunk_with_six_splits(dev):
    """The 7B down projection of one streaming chunk as the engine asks for it (M = 386, N = 3584, 6 slabs): the default tile choice is the
    small variable-height class (3 x 14 x 6 = 252 blocks); with 4 slabs the 128-row tiles keep the call (224 blocks beat 168)."""
    from livecc_amd import ops
    x, w = _rand((386, 1024), dev, 1.0, 1), _rand((3584, 1024), dev, 0.05, 2)
    wp = ops.pack_weight(w)
    before = ops.launch_counts()["gemm_vh"]
    ops.linear_partial(x, wp, 6, packed_shape=(3584, 1024))
    assert ops.launch_counts()["gemm_vh"] == before + 1
    ops.linear_partial(x, wp, 4, packed_shape=(3584, 1024))
    assert ops.launch_counts()["gemm_vh"] == before + 1


def test_gemm_auto_choice_picks_the_variable_height_tiles_for_eight_chunks(dev):
    """M = 3088 (8 x 386 rows), 7B gate/up: the default tile choice takes gemm_vh_kernel (12 x 148 blocks = 6.9 rounds instead of 13 x 148 =
    7.5 -> 8); the o / down shapes (14 column tiles) keep the 192-row tile."""
    from livecc_amd import ops
    x, w = _rand((3088, 512), dev, 1.0, 1), _rand((2 * 18944, 512), dev, 0.05, 2)
    before = ops.launch_counts()["gemm_vh"]
    ops.linear(x, ops.pack_weight(w), None, ops.EPI_SWIGLU, packed_shape=(2 * 18944, 512))
    assert ops.launch_counts()["gemm_vh"] == before + 1
    xd, wd, res = _rand((3088, 512), dev, 1.0, 3), _rand((3584, 512), dev, 0.05, 4), _rand((3088, 3584), dev, 1.0, 5)
    ops.linear(xd, ops.pack_weight(wd), None, ops.EPI_RESIDUAL, res, packed_shape=(3584, 512))
    assert ops.launch_counts()["gemm_vh"] == before + 1


def test_gemm_auto_choice_at_the_7b_chunk_shape_is_the_tall_kernel_and_matches(dev):
    """LiveCC-7B gate/up of one streaming chunk (M = 386, N = 2 x 18944, K = 3584, SwiGLU): the default variant takes the tall kernel
    (237 blocks = one round); same result as the forced 128 x 256 tiles up to the summation order, and within the bound of the fp32
    reference."""
    from livecc_amd import ops
    M, I, K = 386, 18944, 3584
    x, w = _rand((M, K), dev, 1.0, 1), _rand((2 * I, K), dev, 0.02, 2)
    wp = ops.pack_weight(w)
    got = ops.linear(x, wp, None, ops.EPI_SWIGLU, packed_shape=(2 * I, K))
    ops.set_gemm_variant(8)
    try:
        tall = ops.linear(x, wp, None, ops.EPI_SWIGLU, packed_shape=(2 * I, K))
        ops.set_gemm_variant(4)
        big = ops.linear(x, wp, None, ops.EPI_SWIGLU, packed_shape=(2 * I, K))
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    assert torch.equal(got, tall), "the default variant must be the tall kernel at this shape"
    ref, atol = _ref_linear(x, w, None, 4, with_atol=True)
    assert_bf16_close(got, ref, "gemm_tall_7b_gate_up", max_ulp=1.0, max_frac=5e-3, atol=atol)
    assert_bf16_close(big, ref, "gemm_big128_7b_gate_up", max_ulp=1.0, max_frac=5e-3, atol=atol)


def test_fp8_quantiser_gives_the_same_bytes_on_the_gpu_as_on_the_cpu(dev):
    """The arena quantises on the GPU, the oracle on the CPU: same e4m3 bytes and same row scales for the same bf16 matrix (the devices'
    own float -> float8 casts and `tensor / 448.0` do NOT agree: 0.15 % of the elements of a 72B matrix, tools/probes/fp8_quantiser_cpu_vs_gpu.py)."""
    from livecc_amd.weights import quantize_fp8_rows
    g = torch.Generator().manual_seed(11)
    for shape, std in (((1024, 4096), 0.02), ((256, 8192), 1.0), ((64, 512), 1e-4)):
        w = (torch.randn(*shape, generator=g) * std).to(torch.bfloat16)
        qc, sc = quantize_fp8_rows(w)
        qg, sg = quantize_fp8_rows(w.to(dev))
        assert torch.equal(sc, sg.cpu()), shape
        assert torch.equal(qc, qg.cpu()), (shape, int((qc != qg.cpu()).sum()))


@pytest.mark.parametrize("M", [1, 2, 8, 16])
@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944), (1024, 192)])
def test_gemv_w8_fp8_weights(dev, M, N, K):
    """fp8 (OCP e4m3 + fp32 row scale) weight-streaming GEMV: plain (+bias), split-K slabs, and the reference is the bf16 linear
    of the exactly dequantised weights (e4m3 -> bf16 is exact; the scale multiplies the fp32 sum)."""
    from livecc_amd import ops
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.03, 2).float(), _rand((N,), dev, 0.1, 3)
    q, sc = quantize_fp8_rows(w)
    wd = dequantize_fp8_rows(q, sc)                       # fp32, exact products q * scale
    w8, sc2 = ops.quantize_fp8(w)
    assert torch.equal(sc2, sc)
    got = ops.linear_w8(x, w8, sc, b)
    ref, atol = _ref_linear(x, wd, b, with_atol=True)
    assert_bf16_close(got, ref, f"gemv_w8[{M}x{N}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    S = ops.gemv_num_splits(N, K)
    if S <= K // 64:
        part = ops.linear_w8(x, w8, sc, nsplit=S)
        r32 = x.float() @ wd.t()
        err = (part.sum(0) - r32).abs().max().item()
        assert err <= 2e-5 * float(r32.abs().max()) * math.sqrt(K / 256) + 1e-5, f"fp8 split-K slabs: {err}"


@pytest.mark.parametrize("M,I,K", [(1, 512, 256), (8, 2432, 896), (300, 2432, 896), (40, 512, 256), (530, 400, 128)])
@pytest.mark.parametrize("variant", [2, 3, 4])
def test_gemm_w8_swiglu(dev, M, I, K, variant):
    """gate/up with fp8 weights: skinny GEMV (M <= 16) and the dequantise-then-GEMM path (M > 16), SwiGLU epilogue with the
    row scales applied before the activation."""
    from livecc_amd import ops
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    x, w = _rand((M, K), dev, 1.0, 1), _rand((2 * I, K), dev, 0.05, 2).float()
    q, sc = quantize_fp8_rows(w)
    w8, _ = ops.quantize_fp8(w)
    ops.set_gemm_variant(variant)
    try:
        got = ops.linear_w8(x, w8, sc, None, ops.EPI_SWIGLU)
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref, atol = _ref_linear(x, dequantize_fp8_rows(q, sc), None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"gemm_w8_swiglu[{M}x{I}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M,N,K", [(100, 512, 256), (386, 1536, 1024), (700, 768, 512), (33, 272, 192), (300, 4608, 3584)])
@pytest.mark.parametrize("epi", [0, 1, 3])
@pytest.mark.parametrize("variant", [2, 3, 4, 7])
def test_gemm_w8_tiled(dev, M, N, K, epi, variant):
    """M > 16 with fp8 weights.  variant 3 / 4: the 8-wave 256x256 / 128x256 kernel reads the fp8 fragments directly (e4m3 ->
    bf16 after the LDS read, three 48-KB stages); 7: exact dequantisation into the bf16 fragment order + the 4-wave bf16 GEMM;
    2: the shipped choice.  The row scale is applied in the epilogue in every case."""
    from livecc_amd import ops
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2).float(), _rand((N,), dev, 0.1, 3)
    res = _rand((M, N), dev, 1.0, 4) if epi == 3 else None
    q, sc = quantize_fp8_rows(w)
    w8, _ = ops.quantize_fp8(w)
    ops.set_gemm_variant(variant)
    try:
        got = ops.linear_w8(x, w8, sc, b, epi, res)
        part = ops.linear_w8(x, w8, sc, nsplit=2)
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref, atol = _ref_linear(x, dequantize_fp8_rows(q, sc), b, epi, res, with_atol=True)
    assert_bf16_close(got, ref, f"gemm_w8[{M}x{N}x{K},epi{epi},v{variant}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    r32 = x.float() @ dequantize_fp8_rows(q, sc).t()
    assert (part.sum(0) - r32).abs().max().item() <= 2e-5 * float(r32.abs().max()) * math.sqrt(K / 256) + 1e-5


def test_gemm_no_bias_identity_layout(dev):
    """A = I with an asymmetric W catches a transposed C write (symmetric inputs would not)."""
    from livecc_amd import ops
    K = 256
    x = torch.eye(K, dtype=torch.bfloat16, device=dev)[:200].contiguous()
    w = (torch.arange(512 * K, device=dev).view(512, K) % 251 - 125).to(torch.bfloat16).contiguous()
    got = ops.linear(x, w)
    assert torch.equal(got.float(), w.float().t()[:200].contiguous())


@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944), (1024, 160)])
@pytest.mark.parametrize("packed,variant", [(False, 0), (True, 0), (True, 1), (True, 2), (False, 2)])
def test_gemv_skinny(dev, M, N, K, packed, variant):
    from livecc_amd import ops
    ops.set_gemv_variant(variant)
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.03, 2), _rand((N,), dev, 0.1, 3)
    ps = (N, K) if packed else None
    if packed:
        w_row, w = w, ops.pack_weight(w)
    else:
        w_row = w
    try:
        got = ops.linear(x, w, b, packed_shape=ps)
        S = ops.gemv_num_splits(N, K)
        part = ops.linear_partial(x, w, S, packed_shape=ps)
    finally:
        ops.set_gemv_variant(ops.GEMV_DEFAULT_VARIANT)
    w = w_row
    ref, atol = _ref_linear(x, w, b, with_atol=True)
    assert_bf16_close(got, ref, f"gemv[{M}x{N}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    ref = x.float() @ w.float().t()
    err = (part.sum(0) - ref).abs().max().item()
    record(f"gemv_partial[{M}x{N}x{K},S{S}]", dict(max_abs=err, scale=float(ref.abs().max())))
    assert err <= 2e-5 * float(ref.abs().max()) * math.sqrt(K / 256) + 1e-5, f"split-K slabs: {err}"


@pytest.mark.parametrize("M", [17, 32, 33, 48, 64])
@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944), (1024, 160)])
@pytest.mark.parametrize("packed", [False, True])
def test_gemv_17_to_64_rows_streams_the_weights_once(dev, M, N, K, packed):
    """Round 4: decode batches of 17-64 streams.  The weight-streaming kernel multiplies every weight fragment with ceil(M / 16) activation
    fragments (gemv_skinny_kernel<..., MG>): bf16 + bias output and split-K slabs against the fp32 reference, and against the round-3 routing
    of the same call through the 64-row GEMM tiles (lcc_debug_set_skinny_rows(16))."""
    from livecc_amd import ops
    x, w, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.03, 2), _rand((N,), dev, 0.1, 3)
    ps = (N, K) if packed else None
    wk = ops.pack_weight(w) if packed else w
    assert ops.set_skinny_rows(64) == 64, "64 rows is the library default"
    got = ops.linear(x, wk, b, packed_shape=ps)
    S = ops.gemv_num_splits(N, K)
    part = ops.linear_partial(x, wk, S, packed_shape=ps)
    ops.set_skinny_rows(16)
    try:
        tiles = ops.linear(x, wk, b, packed_shape=ps)
    finally:
        ops.set_skinny_rows(64)
    ref, atol = _ref_linear(x, w, b, with_atol=True)
    assert_bf16_close(got, ref, f"gemv_mid[{M}x{N}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    assert_bf16_close(got, tiles, f"gemv_mid_vs_tiles[{M}x{N}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    r32 = x.float() @ w.float().t()
    err = (part.sum(0) - r32).abs().max().item()
    assert part.shape == (S, M, N) and err <= 2e-5 * float(r32.abs().max()) * math.sqrt(K / 256) + 1e-5, f"split-K slabs: {err}"


@pytest.mark.parametrize("M", [17, 40, 64])
@pytest.mark.parametrize("I,K", [(512, 256), (18944, 3584)])
def test_gemv_17_to_64_rows_swiglu(dev, M, I, K):
    from livecc_amd import ops
    x, w = _rand((M, K), dev, 1.0, 1), _rand((2 * I, K), dev, 0.03, 2)
    got = ops.linear(x, ops.pack_weight(w), None, ops.EPI_SWIGLU, packed_shape=(2 * I, K))
    ref, atol = _ref_linear(x, w, None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"gemv_mid_swiglu[{M}x{I}x{K}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


@pytest.mark.parametrize("M,N,K,S", [(100, 512, 1024, 2), (386, 3584, 3584, 4), (70, 256, 640, 3)])
@pytest.mark.parametrize("packed,variant", _layout_variants([0, 1, 3, 4, 5, 13, 14]))
def test_gemm_tiled_splitk_slabs(dev, M, N, K, S, packed, variant):
    """prefill split-K: fp32 slabs [S][M][N] whose sum is the product (reduced by add_rmsnorm in the engine)."""
    from livecc_amd import ops
    x, w = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2)
    ops.set_gemm_variant(variant)
    try:
        part = ops.linear_partial(x, ops.pack_weight(w) if packed else w, S, packed_shape=(N, K) if packed else None)
    finally:
        ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
    ref = x.float() @ w.float().t()
    err = (part.sum(0) - ref).abs().max().item()
    assert err <= 2e-5 * float(ref.abs().max()) * math.sqrt(K / 256) + 1e-5, f"split-K slabs: {err}"
    h = _rand((M, N), dev, 1.0, 3)
    wn = _rand((N,), dev, 1.0, 4)
    h2 = h.clone()
    y = ops.add_rmsnorm_(h2, wn, 1e-6, partial=part)
    h_ref = rb(h.float() + rb(part.sum(0)))
    assert_bf16_close(h2, h_ref, f"splitk_add[{M}x{N}x{K}]", 1.0, 5e-3, atol=32 * 2.0 ** -24 * (x.float().abs() @ w.float().abs().t()))
    assert_bf16_close(y, _ref_rmsnorm(h2, wn, 1e-6), f"splitk_add_rmsnorm[{M}x{N}x{K}]", 1.0, 5e-3)


@pytest.mark.parametrize("M", [1, 7, 16])
@pytest.mark.parametrize("packed", [False, True])
def test_gemv_swiglu(dev, M, packed):
    from livecc_amd import ops
    I, K = 2432, 896
    x, w = _rand((M, K), dev, 1.0, 1), _rand((2 * I, K), dev, 0.05, 2)
    got = ops.linear(x, ops.pack_weight(w), None, ops.EPI_SWIGLU, packed_shape=(2 * I, K)) if packed else ops.linear(x, w, None, ops.EPI_SWIGLU)
    ref, atol = _ref_linear(x, w, None, 4, with_atol=True)
    assert_bf16_close(got, ref, f"gemv_swiglu[{M}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


def test_gemm_rejects_bad_shapes(dev):
    from livecc_amd import _lib, ops
    x, w = _rand((4, 100), dev), _rand((32, 100), dev)   # K % 8 != 0
    with pytest.raises(_lib.LccError):
        ops.linear(x, w)
    with pytest.raises((TypeError, _lib.LccError)):
        ops.linear(x.float(), w)


@pytest.mark.parametrize("rows,dim", [(37, 1280), (5, 160), (300, 320)])
def test_layernorm(dev, rows, dim):
    from livecc_amd import ops
    x, w, b = _rand((rows, dim), dev, 2.0, 1), _rand((dim,), dev, 1.0, 2), _rand((dim,), dev, 0.5, 3)
    got = ops.layernorm(x, w, b, 1e-6)
    ref = rb(torch.nn.functional.layer_norm(x.float(), (dim,), w.float(), b.float(), 1e-6))
    assert_bf16_close(got, ref, f"layernorm[{rows}x{dim}]", max_ulp=1.0, max_frac=5e-3)


def _ref_rmsnorm(x, w, eps):
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return rb(w.float() * rb(xf * torch.rsqrt(var + eps)))


@pytest.mark.parametrize("rows,dim", [(1, 3584), (9, 256), (386, 3584), (3, 8192)])
def test_rmsnorm_and_add(dev, rows, dim):
    from livecc_amd import ops
    x, w = _rand((rows, dim), dev, 3.0, 1), _rand((dim,), dev, 1.0, 2)
    assert_bf16_close(ops.rmsnorm(x, w, 1e-6), _ref_rmsnorm(x, w, 1e-6), f"rmsnorm[{rows}x{dim}]", 1.0, 5e-3)
    # bf16 delta
    d = _rand((rows, dim), dev, 1.0, 3)
    h = x.clone()
    y = ops.add_rmsnorm_(h, w, 1e-6, delta=d)
    h_ref = rb(x.float() + d.float())
    assert torch.equal(h.float(), h_ref), "residual add must be exact (one fp32 add, one rounding)"
    assert_bf16_close(y, _ref_rmsnorm(h_ref, w, 1e-6), f"add_rmsnorm[{rows}x{dim}]", 1.0, 5e-3)
    # fp32 split-K slabs
    if rows <= 16:
        part = torch.randn(4, rows, dim, device=dev) * 0.5
        h = x.clone()
        y = ops.add_rmsnorm_(h, w, 1e-6, partial=part)
        h_ref = rb(x.float() + rb(part[0] + part[1] + part[2] + part[3]))
        assert_bf16_close(h, h_ref, f"add_partial[{rows}x{dim}]", 1.0, 1e-3)
        assert_bf16_close(y, _ref_rmsnorm(h, w, 1e-6), f"add_partial_rmsnorm[{rows}x{dim}]", 1.0, 5e-3)


def test_swiglu(dev):
    from livecc_amd import ops
    g, u = _rand((33, 512), dev, 2.0, 1), _rand((33, 512), dev, 1.0, 2)
    gf = g.float()
    ref = rb(rb(gf / (1.0 + torch.exp(-gf))) * u.float())
    assert_bf16_close(ops.swiglu(g, u), ref, "swiglu", 1.0, 5e-3)


@pytest.mark.parametrize("layout", ["THWC", "TCHW"])
@pytest.mark.parametrize("T,H,W", [(2, 56, 84), (6, 112, 56), (3, 28, 28)])
def test_patchify_norm_matches_oracle(dev, layout, T, H, W):
    """bit-exact vs the oracle restatement of the HF video processor (same fp32 ops, one rounding to bf16)."""
    from livecc_amd import ops, protocol
    from livecc_amd.config import tiny
    from livecc_amd.engine import fused_mean_std
    from oracle import hf_oracle as O
    f = protocol.synth_frames(T, H, W, seed=7, layout="TCHW")
    ref, grid = O.patchify_normalize_ref(torch.from_numpy(f), tiny())
    fin = torch.from_numpy(f if layout == "TCHW" else np.ascontiguousarray(f.transpose(0, 2, 3, 1))).to(dev)
    m, s = fused_mean_std()
    got = ops.patchify_norm(fin, layout, m, s)
    assert got.shape == ref.shape
    assert torch.equal(got.float().cpu(), rb(ref)), "patchify+normalise must be bit-exact"


def test_cast(dev):
    from livecc_amd import ops
    x = torch.randn(1000, 8, device=dev) * 3
    assert torch.equal(ops.cast_f32_bf16(x), x.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def _check_attn(got, ref, name, rel=0.02):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all(), f"{name}: non-finite"
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    record(name, dict(max_abs=err, scale=scale, mean_abs=float((got - ref).abs().mean())))
    assert err <= rel * scale, f"{name}: max err {err:.4g} > {rel} * scale {scale:.4g}"
    assert (got - ref).abs().mean().item() <= 0.25 * rel * scale


@pytest.fixture(params=[0, 1, 2, 3], ids=["attn_per_wave", "attn_prefill_shared", "attn_lds_shared_default", "attn_prefill_mfma32"])
def attn_variant(request):
    from livecc_amd import ops
    ops.set_attn_variant(request.param)
    yield request.param
    ops.set_attn_variant(ops.ATTN_DEFAULT_VARIANT)


@pytest.mark.parametrize("grids", [[(1, 4, 6)], [(3, 4, 6)], [(1, 10, 14), (2, 6, 8)], [(1, 28, 52)]])
def test_vit_rope_attention(dev, grids, attn_variant):
    """VisionAttention core: fp32 2-D RoPE (one rounding) + per-temporal-slice non-causal attention, d = 80."""
    from livecc_amd import ops
    from livecc_amd.config import tiny
    from livecc_amd.engine import vision_rope_tables
    heads, D = 2, 80
    E = heads * D
    P = sum(t * h * w for t, h, w in grids)
    qkv = _rand((P, 3 * E), dev, 1.5, 5)
    cos, sin = vision_rope_tables(grids, tiny())
    assert cos.shape == (P, 40)
    q, k, v = [x.reshape(P, heads, D).float() for x in qkv.float().split(E, dim=1)]
    c = torch.cat([cos, cos], -1).to(dev)[:, None, :]
    s = torch.cat([sin, sin], -1).to(dev)[:, None, :]
    rot = lambda x: torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
    qr, kr = rb(q * c + rot(q) * s), rb(k * c + rot(k) * s)
    ref = torch.zeros(P, heads, D, device=dev)
    off = 0
    for t, h, w in grids:
        n = h * w
        for _ in range(t):
            sl = slice(off, off + n)
            att = torch.einsum("qhd,khd->hqk", qr[sl], kr[sl]) / math.sqrt(D)
            ref[sl] = torch.einsum("hqk,khd->qhd", att.softmax(-1), v[sl])
            off += n
    got = ops.vit_attention(qkv.clone(), cos.to(dev), sin.to(dev), grids, heads)
    _check_attn(got.view(P, heads, D), rb(ref), f"vit_attn{grids}")
    if attn_variant == 3:     # the 4-wave workgroups (128-row groups) of the same kernel: what the engine launches for ONE streaming chunk
        before = ops.launch_counts()["attn_vit32"]
        got4 = ops.vit_attention(qkv.clone(), cos.to(dev), sin.to(dev), grids, heads, group_rows=128)
        assert ops.launch_counts()["attn_vit32"] == before + 1
        _check_attn(got4.view(P, heads, D), rb(ref), f"vit_attn_group128{grids}")
        assert torch.equal(got4, got), "a query row's arithmetic does not depend on the group size"


def _hf_mrope_ref(pos3, theta=1e6):
    inv = 1.0 / (theta ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))
    freqs = pos3.float()[:, :, None] * inv[None, None, :]          # [3,S,64]
    emb = torch.cat([freqs, freqs], -1)
    cos, sin = emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)
    sec = [16, 24, 24] * 2
    cs = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], -1)   # [S,128]
    sn = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], -1)
    return cs, sn, inv


def test_mrope_table(dev):
    from livecc_amd import ops
    S = 300
    pos = torch.stack([torch.randint(0, 33000, (S,)), torch.randint(0, 33000, (S,)), torch.randint(0, 33000, (S,))]).int()
    cs, sn, inv = _hf_mrope_ref(pos)
    c, s = ops.mrope_table(pos.to(dev), inv.to(dev), [16, 24, 24])
    # cos/sin of large fp32 angles: device and host libm may differ by an fp32 ulp, i.e. a rare bf16 flip
    assert_bf16_close(c, cs[:, :64], "mrope_cos", 1.0, 2e-3, atol=1e-3)
    assert_bf16_close(s, sn[:, :64], "mrope_sin", 1.0, 2e-3, atol=1e-3)


def _ref_attn_causal(q, k, v, past):
    """q [S,Hq,128]; k,v [L,Hkv,128] (L = past+S); bottom-right aligned causal GQA, fp32."""
    S, Hq, D = q.shape
    L, Hkv, _ = k.shape
    G = Hq // Hkv
    kk, vv = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    att = torch.einsum("qhd,khd->hqk", q, kk) / math.sqrt(D)
    qpos = past + torch.arange(S, device=q.device)
    mask = torch.arange(L, device=q.device)[None, :] > qpos[:, None]
    att = att.masked_fill(mask[None], float("-inf"))
    return torch.einsum("hqk,khd->qhd", att.softmax(-1), vv)


def test_attn_prefill_tile_rows_contract(dev):
    """tile_rows: 16 or 32 for every kernel; under variant 3 also up to lcc_debug_attn_tile_rows (256 / G packed columns); anything else is
    LCC_ERR_SHAPE -- never a launch over tiles the kernel cannot hold."""
    from livecc_amd import _lib, ops
    Hq, Hkv, D, S = 28, 4, 128, 40
    kv = ops.KvArena(1, 1, Hkv, 256, dev)
    q = _rand((S, Hq * D), dev, 1.0, 3)
    assert ops.attn_tile_rows(28, 4) == 36 and ops.attn_tile_rows(64, 8) == 32 and ops.attn_tile_rows(12, 2) == 42
    with pytest.raises(_lib.LccError):
        ops.attn_tile_rows(28, 5)
    for bad in (37, 48, 24, 8):
        with pytest.raises(_lib.LccError):
            ops.attn_prefill(q, kv, 0, [(0, S, 0)], Hq, tile_rows=bad)
    try:
        ops.set_attn_variant(2)
        assert ops.attn_tile_rows(28, 4) == 32
        with pytest.raises(_lib.LccError):
            ops.attn_prefill(q, kv, 0, [(0, S, 0)], Hq, tile_rows=36)
    finally:
        ops.set_attn_variant(ops.ATTN_DEFAULT_VARIANT)


@pytest.mark.parametrize("Hq,Hkv", [(2, 1), (7, 1), (28, 4), (12, 2), (8, 1)])
def test_rope_append_prefill_decode_attention(dev, Hq, Hkv, attn_variant):
    """M-RoPE apply + KV append (bit-level vs HF's bf16 op sequence), then prefill and decode attention over the cache."""
    from livecc_amd import ops
    D, L0, S1, S2 = 128, 0, 70, 45
    qkv_dim = (Hq + 2 * Hkv) * D
    kv = ops.KvArena(2, 2, Hkv, 256, dev)
    layer, slot = 1, 1
    k_all, v_all = [], []
    past = 0
    for turn, S in enumerate([S1, S2]):
        qkv = _rand((S, qkv_dim), dev, 1.0, 10 + turn)
        pos = torch.arange(past, past + S).int()
        pos3 = torch.stack([pos, pos + (3 if turn == 0 else 0), pos + (5 if turn == 0 else 0)])
        cs, sn, inv = _hf_mrope_ref(pos3)
        c, s = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
        q_got = ops.rope_kv_append(qkv, c, s, torch.full((S,), slot, dtype=torch.int32, device=dev),
                                   pos.to(dev), kv, layer, Hq)
        # HF apply_multimodal_rotary_pos_emb on bf16 tensors: every op rounds
        x = qkv.float().view(S, Hq + 2 * Hkv, D)
        cf, sf = c.float().repeat(1, 2)[:, None, :], s.float().repeat(1, 2)[:, None, :]   # device tables (tested above)
        rot = torch.cat([-x[..., 64:], x[..., :64]], -1)
        emb = rb(rb(x * cf) + rb(rot * sf))
        q_ref, k_new, v_new = emb[:, :Hq], emb[:, Hq:Hq + Hkv], x[:, Hq + Hkv:]
        assert torch.equal(q_got.float().view(S, Hq, D), q_ref), "rotated q must be bit-exact"
        k_all.append(k_new); v_all.append(v_new)
        K, V = torch.cat(k_all), torch.cat(v_all)
        assert torch.equal(kv.k_view(slot, layer)[:, :past + S].float(), K.transpose(0, 1)), "K cache append"
        assert torch.equal(kv.v_view(slot, layer)[:, :past + S].float(), V.transpose(0, 1)), "V cache append (blocked-transposed)"
        ref = _ref_attn_causal(q_ref, K, V, past)
        # the engine's own tile height (36 rows at 7 heads per KV head under variant 3: (row, head) pairs fill the 8 x 32 columns of a block)
        tall = ops.attn_tile_rows(Hq, Hkv)
        for tr, ns in ((32, 1), (16, 1), (16, 3), (32, 2)) + (((tall, 1), (tall, 2)) if tall > 32 else ()):
            got = ops.attn_prefill(q_got, kv, layer, [(slot, S, past)], Hq, tile_rows=tr, nsplit=ns)
            _check_attn(got.view(S, Hq, D), rb(ref), f"attn_prefill[Hq{Hq},turn{turn},rows{tr},split{ns}]")
        past += S
    # decode: one new token, appended at kv_len through the device counter path, several split counts
    qkv = _rand((1, qkv_dim), dev, 1.0, 99)
    bias = _rand((qkv_dim,), dev, 0.2, 98)
    part = torch.stack([qkv[0].float() * 0.25, qkv[0].float() * 0.75]).view(2, 1, qkv_dim).contiguous()
    x_lin = rb(part[0] + part[1] + bias.float())
    pos3 = torch.full((3, 1), past, dtype=torch.int32)
    cs, sn, inv = _hf_mrope_ref(pos3)
    c, s = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    kv_len = torch.tensor([0, past], dtype=torch.int32, device=dev)
    slots = torch.tensor([slot], dtype=torch.int32, device=dev)
    q_got = ops.rope_kv_append(None, c, s, slots, None, kv, layer, Hq, partial=part, bias=bias, kv_len=kv_len)
    x = x_lin.view(1, Hq + 2 * Hkv, D)
    cf, sf = c.float().repeat(1, 2)[:, None, :], s.float().repeat(1, 2)[:, None, :]
    rot = torch.cat([-x[..., 64:], x[..., :64]], -1)
    emb = rb(rb(x * cf) + rb(rot * sf))
    assert torch.equal(q_got.float().view(1, Hq, D), emb[:, :Hq]), "decode-path rope (fp32 slabs + bias)"
    K = torch.cat(k_all + [emb[:, Hq:Hq + Hkv]]); V = torch.cat(v_all + [x[:, Hq + Hkv:]])
    ref = rb(_ref_attn_causal(emb[:, :Hq], K, V, past))
    for nsplit in (1, 2, 5):
        got = ops.attn_decode(q_got, kv, layer, slots, kv_len, Hq, nsplit)
        _check_attn(got.view(1, Hq, D), ref, f"attn_decode[Hq{Hq},nsplit{nsplit}]")


@pytest.mark.parametrize("Hq,Hkv", [(7, 1), (28, 4), (12, 2), (8, 1), (2, 1)])
@pytest.mark.parametrize("NS", [1, 3, 4])
def test_attn_decode_fused_matches_separate_kernels(dev, Hq, Hkv, NS):
    """The fused decode kernel (bias + M-RoPE + KV append + attention + split merge in one launch) against the three separate
    kernels on the same slabs and the same caches: identical K/V appends (bit-exact), attention output within the fp32
    merge-order noise (<= 1 bf16 ulp), for ragged stream lengths (1 key ... several tiles, tile-boundary cases), every split
    count, and with the arrival counters reused across consecutive launches."""
    from livecc_amd import ops
    D, Lmax = 128, 1024
    lens = [0, 31, 32, 33, 500, 991]          # cached keys per stream before the new token
    B = len(lens)
    qkv_dim = (Hq + 2 * Hkv) * D
    kv_a = ops.KvArena(B + 1, 2, Hkv, Lmax, dev)
    g = torch.Generator().manual_seed(5)
    kv_a.buf.copy_((torch.randn(kv_a.buf.shape, generator=g) * 0.7).to(torch.bfloat16))
    kv_b = ops.KvArena(B + 1, 2, Hkv, Lmax, dev)
    kv_b.buf.copy_(kv_a.buf)
    layer = 1
    slots = torch.tensor([3, 0, 5, 1, 6, 2], dtype=torch.int32, device=dev)          # batch row -> slot (a permutation)
    kv_len = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    kv_len[slots.long()] = torch.tensor(lens, dtype=torch.int32, device=dev)
    part = (torch.randn(NS, B, qkv_dim, generator=g) * 0.6).to(dev)
    bias = _rand((qkv_dim,), dev, 0.2, 98)
    pos3 = torch.tensor([[l + 2 for l in lens]] * 3, dtype=torch.int32)
    _, _, inv = _hf_mrope_ref(pos3)
    c, s_ = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    q_ref = ops.rope_kv_append(None, c, s_, slots, None, kv_a, layer, Hq, partial=part, bias=bias, kv_len=kv_len)
    ref = ops.attn_decode(q_ref, kv_a, layer, slots, kv_len, Hq, 4).float()
    counters = None
    for nsplit in (1, 2, 3, 8, 16):
        kv_b.buf.copy_(kv_a.buf)     # (already holds the appended rows of the reference; the fused kernel rewrites them)
        got, counters = ops.attn_decode_fused(part, bias, c, s_, kv_b, layer, slots, kv_len, Hq, nsplit, counters)
        assert int(counters.abs().sum()) == 0, "arrival counters must be left at zero"
        assert torch.equal(kv_b.buf, kv_a.buf), f"fused KV append differs (nsplit {nsplit})"
        _check_attn(got.view(B, Hq, D), ref.view(B, Hq, D), f"attn_decode_fused[Hq{Hq},NS{NS},nsplit{nsplit}]")
    # the append itself: start from caches WITHOUT the new rows
    kv_c = ops.KvArena(B + 1, 2, Hkv, Lmax, dev)
    kv_c.buf.copy_(kv_a.buf)
    for b in range(B):
        sl, n = int(slots[b]), lens[b]
        kv_c.k_view(sl, layer)[:, n] = 7.0
        o = (layer * 2 + 1) * Hkv * Lmax * 128
        raw = kv_c.buf[sl, o:o + Hkv * Lmax * 128].view(Hkv, Lmax // 32, 128, 32)
        raw[:, n // 32, :, n % 32] = 7.0
    assert not torch.equal(kv_c.buf, kv_a.buf)
    got, _ = ops.attn_decode_fused(part, bias, c, s_, kv_c, layer, slots, kv_len, Hq, 4)
    assert torch.equal(kv_c.buf, kv_a.buf), "fused kernel must append exactly the rows rope_kv_append appends"
    _check_attn(got.view(B, Hq, D), ref.view(B, Hq, D), f"attn_decode_fused_append[Hq{Hq},NS{NS}]")


def test_attention_masks_garbage_beyond_length(dev, attn_variant):
    """Keys past the valid length (stale cache contents) must not leak into the result."""
    from livecc_amd import ops
    Hq, Hkv, D, S = 2, 1, 128, 40
    kv = ops.KvArena(1, 1, Hkv, 128, dev)
    kv.buf.fill_(1e4)   # finite garbage everywhere
    qkv = _rand((S, (Hq + 2 * Hkv) * D), dev, 1.0, 3)
    pos = torch.arange(S).int()
    pos3 = torch.stack([pos, pos, pos])
    _, _, inv = _hf_mrope_ref(pos3)
    c, s = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    q = ops.rope_kv_append(qkv, c, s, torch.zeros(S, dtype=torch.int32, device=dev), pos.to(dev), kv, 0, Hq)
    got = ops.attn_prefill(q, kv, 0, [(0, S, 0)], Hq)
    K = kv.k_view(0, 0)[:, :S].float().transpose(0, 1); V = kv.v_view(0, 0)[:, :S].float().transpose(0, 1)
    ref = _ref_attn_causal(q.float().view(S, Hq, D), K, V, 0)
    _check_attn(got.view(S, Hq, D), rb(ref), "attn_garbage_tail")


# ---------------------------------------------------------------------------------------------
# sampler
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("V,two_stage", [(2048, False), (16384, True), (152064, True)])
def test_sampler_matches_hf_processors(dev, V, two_stage):
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    from livecc_amd import ops
    from oracle.hf_oracle import ThresholdLogitsProcessor
    B = 3
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(B, V, generator=g) * 2).to(torch.bfloat16)
    hist = [torch.randint(0, V, (50,), generator=g) for _ in range(B)]
    seen_np = np.zeros((B, V // 32), dtype=np.uint32)
    for b in range(B):
        for t in hist[b].tolist():
            seen_np[b, t >> 5] |= np.uint32(1 << (t & 31))
    seen = torch.from_numpy(seen_np.view(np.int32))
    slots = torch.arange(B, dtype=torch.int32)
    for pen, thr, eos_sup in [(1.0, None, False), (1.15, None, False), (1.05, 0.002, False), (1.05, None, True)]:
        thr_tok = int(logits[0].float().argmax())          # make the threshold token matter for stream 0
        eos = int(logits[1].float().argmax())              # and EOS suppression for stream 1
        tok, scores = ops.sample_greedy(logits.to(dev), seen.to(dev), slots.to(dev), pen, thr_tok if thr is not None else -1,
                                        thr, eos if eos_sup else -1, eos_sup, want_scores=True, two_stage=two_stage)
        for b in range(B):
            sc = logits[b:b + 1].float().clone()
            sc = RepetitionPenaltyLogitsProcessor(pen)(hist[b].view(1, -1), sc) if pen != 1.0 else sc
            if eos_sup:
                sc[:, eos] = -float("inf")
            if thr is not None:
                sc = ThresholdLogitsProcessor(thr_tok, thr, 0.0)(hist[b].view(1, -1), sc)
            assert int(tok[b]) == int(sc.argmax()), f"token mismatch pen={pen} thr={thr} stream {b}"
            assert torch.allclose(scores[b].cpu(), sc[0], rtol=1e-6, atol=1e-6), "processed scores"


def test_liger_style_plugin_patches_hf_modules(dev):
    """apply_livecc_amd_kernel_to_qwen2_vl() rebinds the HF module names like liger does (ref demo/infer.py:2-3)."""
    import transformers.models.qwen2_vl.modeling_qwen2_vl as m
    from livecc_amd.config import tiny
    from livecc_amd.plugin import apply_livecc_amd_kernel_to_qwen2_vl
    saved = (m.Qwen2VLRMSNorm, m.LayerNorm, m.Qwen2MLP)
    try:
        ref_norm = m.Qwen2VLRMSNorm(256, eps=1e-6).to(dev, torch.bfloat16)
        ref_norm.weight.data.copy_(_rand((256,), dev, 1.0, 1))
        ref_mlp = m.Qwen2MLP(tiny().to_hf().text_config).to(dev, torch.bfloat16)
        apply_livecc_amd_kernel_to_qwen2_vl()
        assert m.Qwen2VLRMSNorm is not saved[0] and m.LayerNorm is not saved[1] and m.Qwen2MLP is not saved[2]
        new_norm = m.Qwen2VLRMSNorm(256, eps=1e-6).to(dev, torch.bfloat16)
        new_norm.load_state_dict(ref_norm.state_dict())
        x = _rand((3, 7, 256), dev, 2.0, 2)
        assert_bf16_close(new_norm(x), ref_norm(x), "plugin_rmsnorm", 1.0, 5e-3)
        new_mlp = m.Qwen2MLP(tiny().to_hf().text_config).to(dev, torch.bfloat16)
        new_mlp.load_state_dict(ref_mlp.state_dict())
        y_ref, y_new = ref_mlp(x), new_mlp(x)
        assert (y_ref.float() - y_new.float()).abs().max() <= 0.02 * y_ref.float().abs().max() + 1e-3
        ln = m.LayerNorm(160, eps=1e-6).to(dev, torch.bfloat16)
        xv = _rand((10, 160), dev, 2.0, 3)
        assert_bf16_close(ln(xv), torch.nn.functional.layer_norm(xv.float(), (160,), ln.weight.float(), ln.bias.float(), 1e-6).to(torch.bfloat16),
                          "plugin_layernorm", 1.0, 5e-3)
    finally:
        m.Qwen2VLRMSNorm, m.LayerNorm, m.Qwen2MLP = saved


def test_unmodified_hf_model_on_the_gpu_with_every_plugin_matches_hf_cpu(dev):
    """SURVEY 8b items 1-3 under the HF module graph: `apply_livecc_amd_kernel_to_qwen2_vl()` (RMSNorm, LayerNorm, SwiGLU,
    apply_multimodal_rotary_pos_emb) + `attn_implementation="livecc_amd"` (AttentionInterface: ViT slices and causal GQA) +
    `past_key_values=NativeKVCache(...)` (Cache.update appends in place) -- an UNMODIFIED HF Qwen2VLForConditionalGeneration placed
    on the ROCm device runs the reference's two-turn streaming protocol through them; logits vs the same model on the CPU with
    HF's own modules (bf16), teacher-forced along the GPU tokens."""
    from livecc_amd import plugin, protocol, torch_ops
    from livecc_amd.config import tiny
    from oracle import hf_oracle as O
    cfg = tiny()
    hf_cpu = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=0, init_scale=2.0)
    # round 6: the plugins reach the kernels through the dispatcher -- torch.ops.livecc_amd.* (csrc/torch_ops.cpp) -- and
    # this test proves it by counting the calls routed that way
    assert torch_ops.try_load() is not None, "liblivecc_torch_ops.so must be built (python -m livecc_amd.build) and loadable on the GPU box"
    torch_ops.CALLS.clear()
    plugin.apply_livecc_amd_kernel_to_qwen2_vl()
    try:
        import transformers.models.qwen2_vl.modeling_qwen2_vl as m
        hf_gpu = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=0, init_scale=2.0, attn_implementation=plugin.ATTN_NAME)
        assert type(hf_gpu.model.language_model.norm).__name__ == "LccRMSNorm" and m.apply_multimodal_rotary_pos_emb is plugin.lcc_apply_multimodal_rotary_pos_emb
        hf_gpu.load_state_dict(hf_cpu.state_dict())
        hf_gpu = hf_gpu.to(dev)
        cache = plugin.NativeKVCache(hf_gpu.config, 1024, dev)
        frames = torch.from_numpy(protocol.synth_frames(8, 56, 84, seed=1234, layout="TCHW"))
        builder = protocol.TurnBuilder(cfg, seed=1234)
        s_gpu, s_cpu = O.OracleStream(hf_gpu, cfg, device=dev, past_key_values=cache), O.OracleStream(hf_cpu, cfg)
        worst = 0.0
        for ti, (a, b) in enumerate(protocol.split_clip(8)):
            pv, grid = O.patchify_normalize_ref(frames[a:b], cfg)
            ids = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
            rg = s_gpu.turn(ids, pv, grid, max_new_tokens=4, repetition_penalty=1.05)
            rc = s_cpu.turn(ids, pv, grid, max_new_tokens=4, repetition_penalty=1.05, teacher_tokens=rg["new_tokens"])
            for k in range(4):
                lg, lc = rg["logits"][k], rc["logits"][k]
                worst = max(worst, ((lg - lc).abs().max() / lc.abs().max()).item())
        assert s_gpu.past_key_values is cache and cache.get_seq_length() == s_gpu.past_ids.shape[1]
        record("hf_model_with_plugins_vs_hf_cpu", dict(worst_rel_dlogit=worst))
        assert worst <= 6e-2, f"HF-on-GPU with the native plugins differs from HF CPU by {worst:.3g} of the logit scale"
        routed = {n: torch_ops.CALLS[n] for n in ("rmsnorm", "layernorm", "swiglu", "rope_kv_append", "attn_prefill", "attn_decode")}
        record("hf_model_with_plugins_torch_ops_calls", routed)
        assert all(v > 0 for v in routed.values()), f"every plugin op must have run through torch.ops.livecc_amd: {routed}"
        # HF crop semantics (keep n tokens / negative: remove n) and the weak registry of live caches
        n = cache.get_seq_length()
        cache.crop(n + 5); assert cache.get_seq_length() == n
        cache.crop(n - 3); assert cache.get_seq_length() == n - 3
        cache.crop(-2); assert cache.get_seq_length() == n - 5
        import gc
        live_before = sum(r() is not None for r in plugin._LIVE_CACHES)
        s_gpu.past_key_values = None
        del cache
        gc.collect()
        assert sum(r() is not None for r in plugin._LIVE_CACHES) == live_before - 1, "a dropped NativeKVCache must free its arena"
    finally:
        plugin.revert_livecc_amd_kernel_to_qwen2_vl()


def test_embed_gather_with_video_row_scatter_is_bit_exact(dev):
    """SURVEY a12 (Q2VL:1159-1176): embedding rows + the ViT rows scattered over the <|video_pad|> positions in order, and the decode
    form (current token of a stream slot through an indirection) -- pure data movement, bit-exact."""
    from livecc_amd import ops
    V, H, S = 1000, 896, 301
    g = torch.Generator().manual_seed(7)
    table = _rand((V, H), dev, 1.0, 11)
    ids = torch.randint(0, V, (S,), generator=g, dtype=torch.int32)
    is_vid = torch.zeros(S, dtype=torch.bool)
    is_vid[20:120] = True
    is_vid[200:230] = True
    n_vid = int(is_vid.sum())
    vit = _rand((n_vid, H), dev, 1.0, 12)
    vit_index = torch.full((S,), -1, dtype=torch.int32)
    vit_index[is_vid] = torch.arange(n_vid, dtype=torch.int32)
    got = ops.embed_gather(ids.to(dev), table, vit_index.to(dev), vit)
    ref = table[ids.long().to(dev)].clone()
    ref[is_vid.to(dev)] = vit                                   # masked_scatter: video rows in order
    assert torch.equal(got, ref)
    assert torch.equal(ops.embed_gather(ids.to(dev), table), table[ids.long().to(dev)])
    cur_tok = torch.randint(0, V, (16,), generator=g, dtype=torch.int32)      # per-slot current token; streams use slots 9, 2, 14
    slots = torch.tensor([9, 2, 14], dtype=torch.int32)
    got = ops.embed_gather(cur_tok.to(dev), table, indirect=slots.to(dev))
    assert torch.equal(got, table[cur_tok[slots.long()].long().to(dev)])
