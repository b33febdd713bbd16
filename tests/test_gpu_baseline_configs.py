"""Parity at BASELINE.json's OWN configurations (-m gpu), each against a reference that is not this library:

  configs[1]  LiveCC-7B, 6-frame turn + 2-frame turn on the carried KV : the HF CPU path with the SAME seeded weights,
              teacher-forced along the native tokens -- logits of all 32 steps against the bf16 reference AND the fp32 truth
              (error-ratio bounds, worst logit and rms), plus greedy token identity (32/32) on the `decisive` synthetic weights at
              7B and 2B shapes.  Slow (minutes of host time): skipped with LCC_SKIP_SLOW=1.
  configs[3]  KV up to 32k: decode attention (16 / 32 / 64 key splits + combine), prefill attention (2-8 key splits, 16- and
              32-row tiles + combine) and the fused decode attention at L = 4,096 / 12,288 / 32,736 keys against the fp32
              softmax(QK^T)V of `_ref_attn_causal`; one end-to-end turn at `small` shapes over a 12k-token history vs HF.
  configs[4]  Qwen2-VL-72B shapes, fp8 weights: the weight-streaming GEMV and the 8-wave fp8 GEMM at K = 8192,
              N = 2 x 29568 / 8192 / 10240 and K = 29568 against the fp32 linear of the exactly dequantised weights.

Tolerances are the ones of tests/test_gpu_ops.py (<= 1 bf16 ulp + fp32 summation-order noise for Linear; 2 % of the output
scale for attention, whose P matrix is rounded to bf16 as in HF's kernels) and of tests/test_gpu_e2e.py for logits.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.test_gpu_ops import _check_attn, _rand, _ref_attn_causal, _ref_linear
from tests.util import assert_bf16_close, rb, record

pytestmark = pytest.mark.gpu

LONG_L = [4096, 12288, 32736]


def _filled_arena(dev, Hkv, Lmax, n_slots=1, seed=3, scale=1.0):
    from livecc_amd import ops
    kv = ops.KvArena(n_slots, 1, Hkv, Lmax, dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for s in range(n_slots):   # filled slot by slot to bound the host staging
        kv.buf[s].copy_((torch.randn(kv.buf.shape[1], generator=g) * scale).to(torch.bfloat16))
    return kv


def _ref_attn_chunked(q, K, V, past, chunk=64):
    """`_ref_attn_causal` over query chunks (fp32 scores of 386 x 28 x 32k would be 1.4 GB at once); keys beyond a chunk's last
    row are cut (the causal mask would remove them anyway)."""
    outs = []
    for a in range(0, q.shape[0], chunk):
        n = min(chunk, q.shape[0] - a)
        outs.append(_ref_attn_causal(q[a:a + n], K[:past + a + n], V[:past + a + n], past + a))
    return torch.cat(outs)


@pytest.mark.parametrize("L", LONG_L)
@pytest.mark.parametrize("qscale", [1.0, 3.0], ids=["flat", "peaky"])
def test_decode_attention_long_cache_vs_fp32(dev, L, qscale):
    """configs[1]/[3]: one query row against L cached keys, GQA 28:4, every key-split count the engine uses (16/32/64) plus 1 and 5."""
    from livecc_amd import ops
    Hq, Hkv, D = 28, 4, 128
    kv = _filled_arena(dev, Hkv, 32768)
    q = _rand((1, Hq * D), dev, qscale, 11)
    slots = torch.zeros(1, dtype=torch.int32, device=dev)
    kv_len = torch.tensor([L - 1], dtype=torch.int32, device=dev)          # attends to kv_len + 1 keys
    K = kv.k_view(0, 0)[:, :L].float().transpose(0, 1)
    V = kv.v_view(0, 0)[:, :L].float().transpose(0, 1)
    ref = rb(_ref_attn_causal(q.float().view(1, Hq, D), K, V, L - 1))
    for nsplit in (1, 5, 16, 32, 64):
        got = ops.attn_decode(q, kv, 0, slots, kv_len, Hq, nsplit)
        _check_attn(got.view(1, Hq, D), ref, f"attn_decode_long[L{L},q{qscale},nsplit{nsplit}]")


def test_decode_attention_ragged_batch_long_caches(dev):
    """Three streams with 4k / 12k / 32k keys in ONE launch (the multi-stream decode of configs[2]/[3])."""
    from livecc_amd import ops
    Hq, Hkv, D = 28, 4, 128
    kv = _filled_arena(dev, Hkv, 32768, n_slots=3)
    q = _rand((3, Hq * D), dev, 1.5, 12)
    slots = torch.tensor([2, 0, 1], dtype=torch.int32, device=dev)
    lens = {2: 4096, 0: 12288, 1: 32736}
    kv_len = torch.tensor([lens[0] - 1, lens[1] - 1, lens[2] - 1], dtype=torch.int32, device=dev)
    for nsplit in (16, 64):
        got = ops.attn_decode(q, kv, 0, slots, kv_len, Hq, nsplit)
        for b, s in enumerate(slots.tolist()):
            L = lens[s]
            K = kv.k_view(s, 0)[:, :L].float().transpose(0, 1)
            V = kv.v_view(s, 0)[:, :L].float().transpose(0, 1)
            ref = rb(_ref_attn_causal(q[b:b + 1].float().view(1, Hq, D), K, V, L - 1))
            _check_attn(got[b:b + 1].view(1, Hq, D), ref, f"attn_decode_ragged[slot{s},L{L},nsplit{nsplit}]")


@pytest.mark.parametrize("L", LONG_L)
@pytest.mark.parametrize("S", [386, 100])
def test_prefill_attention_long_cache_vs_fp32(dev, L, S):
    """configs[1]/[3]: a streaming chunk (386 new rows at 392x728, 100+ at 280x280) attending to L keys, bottom-right causal,
    16- and 32-row tiles, key splits 1 / 2 / 4 / 8 with the combine kernel, all three attention variants."""
    from livecc_amd import ops
    Hq, Hkv, D = 28, 4, 128
    kv = _filled_arena(dev, Hkv, 32768, seed=5)
    past = L - S
    q = _rand((S, Hq * D), dev, 1.5, 13)
    K = kv.k_view(0, 0)[:, :L].float().transpose(0, 1)
    V = kv.v_view(0, 0)[:, :L].float().transpose(0, 1)
    ref = rb(_ref_attn_chunked(q.float().view(S, Hq, D), K, V, past))
    try:
        for variant in (2, 3, 1, 0):
            ops.set_attn_variant(variant)
            tall = ops.attn_tile_rows(Hq, Hkv)      # 36 under variant 3 at 28 / 4 heads (the engine's tiles), else 32
            unsplit = {}
            for tr, ns in ((16, 1), (16, 4), (32, 2), (16, 8), (32, 8), (32, 1), (32, 5)) + (((tall, 1), (tall, 5)) if tall > 32 else ()):
                if variant == 3 and tr < 32:
                    continue            # variant 3 = the 32x32x16-MFMA kernel (attn32.hip): tiles of 32 rows and more
                if variant in (0, 1) and (tr, ns) not in ((16, 4), (32, 8)):
                    continue
                got = ops.attn_prefill(q, kv, 0, [(0, S, past)], Hq, tile_rows=tr, nsplit=ns)
                _check_attn(got.view(S, Hq, D), ref, f"attn_prefill_long[L{L},S{S},v{variant},rows{tr},split{ns}]")
                if variant == 3 and ns == 1:
                    unsplit[tr] = got
            if variant == 3 and tall > 32:           # a (row, head) pair sees the same key tiles in the same order whatever tile it sits in
                assert torch.equal(unsplit[32], unsplit[tall]), "36-row tiles must reproduce the 32-row tiles bit for bit"
    finally:
        ops.set_attn_variant(ops.ATTN_DEFAULT_VARIANT)


def test_fused_decode_attention_long_caches_vs_fp32(dev):
    """The multi-stream fused kernel (bias + M-RoPE + KV append + attention) at 4k / 12k / 32k keys against fp32 attention over
    the cache it appended to (its append itself is pinned bit-exact in test_gpu_ops)."""
    from livecc_amd import ops
    from tests.test_gpu_ops import _hf_mrope_ref
    Hq, Hkv, D, Lmax = 28, 4, 128, 32768
    lens = [4095, 12287, 32735]
    B = len(lens)
    kv = _filled_arena(dev, Hkv, Lmax, n_slots=B, seed=7)
    qkv_dim = (Hq + 2 * Hkv) * D
    g = torch.Generator().manual_seed(9)
    part = (torch.randn(2, B, qkv_dim, generator=g) * 0.7).to(dev)
    bias = _rand((qkv_dim,), dev, 0.2, 98)
    slots = torch.arange(B, dtype=torch.int32, device=dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos3 = torch.tensor([[l + 2 for l in lens]] * 3, dtype=torch.int32)
    _, _, inv = _hf_mrope_ref(pos3)
    c, s_ = ops.mrope_table(pos3.to(dev), inv.to(dev), [16, 24, 24])
    for nsplit in (8, 32):
        got, counters = ops.attn_decode_fused(part, bias, c, s_, kv, 0, slots, kv_len, Hq, nsplit)
        assert int(counters.abs().sum()) == 0
        # the rotated q of the separate kernel (bit-exact vs HF's op sequence, test_gpu_ops) is the reference query
        kv2 = ops.KvArena(B, 1, Hkv, Lmax, dev)
        kv2.buf.copy_(kv.buf)
        q_ref = ops.rope_kv_append(None, c, s_, slots, None, kv2, 0, Hq, partial=part, bias=bias, kv_len=kv_len)
        assert torch.equal(kv2.buf, kv.buf)
        for b, L in enumerate(lens):
            K = kv.k_view(b, 0)[:, :L + 1].float().transpose(0, 1)
            V = kv.v_view(b, 0)[:, :L + 1].float().transpose(0, 1)
            ref = rb(_ref_attn_causal(q_ref[b:b + 1].float().view(1, Hq, D), K, V, L))
            _check_attn(got[b:b + 1].view(1, Hq, D), ref, f"attn_decode_fused_long[L{L + 1},nsplit{nsplit}]")


def test_one_shot_12k_history_then_streaming_turn_matches_hf(dev):
    """configs[3] end to end at `small` shapes (GQA 7:1 like the 7B): a 240-frame 280x280 clip in ONE turn (12,000 visual tokens:
    the video_qa / MCQ prefill; grid_t = 120 >> max(h,w)/2, where the M-RoPE text offset rule matters) served by the native path in
    pieces (max_new_rows 2,048 -> six prefill calls over the carried KV; max_patches 8,192 -> six ViT calls), then a 2-frame
    streaming turn on the 12k-key cache.  HF (bf16 and fp32, CPU) prefills the same prompt in one pass.  Same logits bound, fp32
    error ratio and margin-aware token identity as every other end-to-end test."""
    from livecc_amd import protocol
    from livecc_amd.config import small
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from oracle import hf_oracle as O
    from tests.test_gpu_e2e import _compare_stream
    cfg = small()
    hf16 = O.build_hf_model(cfg, dtype=torch.bfloat16, seed=3, init_scale=1.5)
    hf32 = O.build_hf_model(cfg, dtype=torch.float32, seed=3, init_scale=1.5)
    native = LiveCCForConditionalGeneration.from_hf_model(hf16, cfg, dev, max_streams=1, max_kv_len=16384, max_new_rows=2048,
                                                          max_patches=8192, max_history=16)
    T, H, W, n_tok = 242, 280, 280, 6
    frames = torch.from_numpy(protocol.synth_frames(T, H, W, seed=21, layout="TCHW"))
    builder = protocol.TurnBuilder(cfg, seed=21)
    turns, state, past = [], None, None
    for ti, (a, b) in enumerate([(0, 240), (240, 242)]):
        grid = protocol.grid_of(b - a, H, W, cfg)
        new = builder.turn_ids(ti, protocol.num_video_tokens(grid, cfg))
        ids = new if past is None else np.concatenate([past, new])
        r = native.generate(input_ids=torch.from_numpy(ids).view(1, -1), frames=frames[a:b].to(dev), frames_layout="TCHW", past_key_values=state,
                            do_sample=False, repetition_penalty=1.05, max_new_tokens=n_tok, min_new_tokens=n_tok, output_logits=True)
        state = r.past_key_values
        seq = r.sequences[0].cpu().numpy()
        past = seq[:-1]
        turns.append(dict(turn_ids=new, grid=grid, new_tokens=seq[len(ids):].tolist(), logits=r.logits.float().cpu(), frames=(a, b)))
    assert state.get_seq_length() == len(past) and len(past) > 12000
    state.release()
    _compare_stream(cfg, hf16, hf32, turns, frames, "one_shot_12k_history_small", 1.05)


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: 72B shapes, fp8 weights
# ---------------------------------------------------------------------------------------------------------------------
SHAPES_72B = [(2 * 29568, 8192, "gate_up"), (8192, 29568, "down"), (10240, 8192, "qkv"), (8192, 8192, "o")]


def _w8_case(dev, N, K, seed):
    from livecc_amd.weights import dequantize_fp8_rows, quantize_fp8_rows
    g = torch.Generator(device=dev).manual_seed(seed)
    w = torch.randn((N, K), generator=g, device=dev, dtype=torch.float32) * 0.02
    q, sc = quantize_fp8_rows(w)
    return w, q, sc, dequantize_fp8_rows(q, sc)


@pytest.mark.parametrize("N,K,name", SHAPES_72B)
@pytest.mark.parametrize("M", [1, 8])
def test_gemv_w8_at_72b_shapes(dev, N, K, name, M):
    from livecc_amd import ops
    x = _rand((M, K), dev, 1.0, 1)
    w, q, sc, wd = _w8_case(dev, N, K, 2)
    w8, sc2 = ops.quantize_fp8(w)
    assert torch.equal(sc2, sc)
    if name == "gate_up":
        got = ops.linear_w8(x, w8, sc, None, ops.EPI_SWIGLU)
        ref, atol = _ref_linear(x, wd, None, 4, with_atol=True)
    else:
        b = _rand((N,), dev, 0.1, 3)
        got = ops.linear_w8(x, w8, sc, b)
        ref, atol = _ref_linear(x, wd, b, with_atol=True)
    assert_bf16_close(got, ref, f"gemv_w8_72b[{name},M{M}]", max_ulp=1.0, max_frac=5e-3, atol=atol)
    S = ops.gemv_num_splits(N, K)
    if name != "gate_up" and S <= K // 64:
        part = ops.linear_w8(x, w8, sc, nsplit=S)
        r32 = x.float() @ wd.t()
        assert (part.sum(0) - r32).abs().max().item() <= 2e-5 * float(r32.abs().max()) * math.sqrt(K / 256) + 1e-5


@pytest.mark.parametrize("N,K,name", SHAPES_72B)
def test_gemm_w8_prefill_at_72b_shapes(dev, N, K, name):
    """M = 386 (one streaming chunk): the shipped variant choice and the forced 8-wave fp8 kernels."""
    from livecc_amd import ops
    M = 386
    x = _rand((M, K), dev, 1.0, 1)
    w, q, sc, wd = _w8_case(dev, N, K, 4)
    w8, _ = ops.quantize_fp8(w)
    for variant in (2, 4):
        ops.set_gemm_variant(variant)
        try:
            if name == "gate_up":
                got = ops.linear_w8(x, w8, sc, None, ops.EPI_SWIGLU)
                ref, atol = _ref_linear(x, wd, None, 4, with_atol=True)
            elif name in ("down", "o"):
                res = _rand((M, N), dev, 1.0, 5)
                got = ops.linear_w8(x, w8, sc, None, ops.EPI_RESIDUAL, res)
                ref, atol = _ref_linear(x, wd, None, 3, res, with_atol=True)
            else:
                b = _rand((N,), dev, 0.1, 3)
                got = ops.linear_w8(x, w8, sc, b)
                ref, atol = _ref_linear(x, wd, b, with_atol=True)
        finally:
            ops.set_gemm_variant(ops.GEMM_DEFAULT_VARIANT)
        assert_bf16_close(got, ref, f"gemm_w8_72b[{name},v{variant}]", max_ulp=1.0, max_frac=5e-3, atol=atol)


# ---------------------------------------------------------------------------------------------------------------------
# sampling kernel vs HF's warpers
# ---------------------------------------------------------------------------------------------------------------------
def _seen_bitmap(hist, V):
    B = len(hist)
    seen_np = np.zeros((B, V // 32), dtype=np.uint32)
    for b in range(B):
        for t in hist[b].tolist():
            seen_np[b, t >> 5] |= np.uint32(1 << (t & 31))
    return torch.from_numpy(seen_np.view(np.int32))


@pytest.mark.parametrize("V", [2048, 152064])
@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 50, 1.0), (0.7, 0, 0.9), (1.3, 20, 0.8), (0.01, 5, 0.001), (1.0, 0, 1.0),
                                                     (0.5, 1000, 0.95)])
def test_sampling_processed_distribution_matches_hf_warpers(dev, V, temperature, top_k, top_p):
    """do_sample=True: the processed scores (what HF hands to softmax + multinomial) equal RepetitionPenalty ->
    ThresholdLogitsProcessor -> TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper to 1e-6, the kept sets are
    identical, the drawn token lies in the kept set, and draws are reproducible for a seed."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                                                        TopPLogitsWarper)
    from livecc_amd import ops
    from oracle.hf_oracle import ThresholdLogitsProcessor
    B = 3
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(B, V, generator=g) * 3).to(torch.bfloat16)
    hist = [torch.randint(0, V, (60,), generator=g) for _ in range(B)]
    seen = _seen_bitmap(hist, V).to(dev)
    slots = torch.arange(B, dtype=torch.int32, device=dev)
    pen = 1.1
    thr_tok = int(logits[0].float().argmax())
    tok, scores = ops.sample_topk_topp(logits.to(dev), seen, slots, temperature, top_k, top_p, seed=123, repetition_penalty=pen,
                                       thr_token=thr_tok, thr_value=0.5, want_scores=True)
    tok2, _ = ops.sample_topk_topp(logits.to(dev), seen, slots, temperature, top_k, top_p, seed=123, repetition_penalty=pen,
                                   thr_token=thr_tok, thr_value=0.5)
    assert torch.equal(tok, tok2), "same seed, same counters -> same draws"
    for b in range(B):
        sc = logits[b:b + 1].float().clone()
        sc = RepetitionPenaltyLogitsProcessor(pen)(hist[b].view(1, -1), sc)
        sc = ThresholdLogitsProcessor(thr_tok, 0.5, 0.0)(hist[b].view(1, -1), sc)
        if temperature != 1.0:
            sc = TemperatureLogitsWarper(temperature)(None, sc)
        if top_k:
            sc = TopKLogitsWarper(top_k)(None, sc)
        pre_top_p = sc[0].clone()
        if top_p < 1.0:
            sc = TopPLogitsWarper(top_p)(None, sc)
        got = scores[b].cpu()
        keep_ref, keep_got = torch.isfinite(sc[0]), torch.isfinite(got)
        # The top-p cut falls INSIDE a group of equal scores whenever the boundary value is tied (bf16 logits: ~50 ids per distinct
        # value at V = 152k).  HF then keeps whichever members of the group its (unstable) sort happened to place last; the kernel
        # cuts by value and keeps the whole group -- the only difference allowed: ids of ONE tied value, all kept by the kernel.
        diff = keep_ref != keep_got
        if bool(diff.any()):
            assert pre_top_p[diff].unique().numel() == 1, f"kept sets differ beyond one tie group: {int(keep_ref.sum())} vs {int(keep_got.sum())}"
            assert bool(keep_got[diff].all()), "the kernel keeps the whole boundary tie group"
            boundary = pre_top_p[diff][0]
            assert bool((keep_ref & (pre_top_p == boundary)).any()) or int(diff.sum()) <= 1, "HF cut inside that tie group"
        both = keep_ref & keep_got
        assert torch.allclose(got[both], sc[0][both], rtol=1e-6, atol=1e-6)
        assert bool(keep_got[int(tok[b])]), "drawn token must be in the kept set"


def test_sampling_frequencies_follow_the_distribution_and_topk1_is_greedy(dev):
    from livecc_amd import ops
    V, B = 2048, 1
    g = torch.Generator().manual_seed(2)
    logits = (torch.randn(B, V, generator=g) * 2).to(torch.bfloat16).to(dev)
    seen = torch.zeros(B, V // 32, dtype=torch.int32, device=dev)
    slots = torch.zeros(1, dtype=torch.int32, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    n = 4000
    counts = torch.zeros(V)
    for _ in range(n):
        tok, _ = ops.sample_topk_topp(logits, seen, slots, 1.0, 8, 1.0, seed=7, rng_ctr=ctr)
        counts[int(tok[0])] += 1
    assert int(ctr[0]) == n, "one draw counter tick per sample"
    top = logits[0].float().topk(8)
    p = torch.softmax(top.values, -1).cpu()
    f = counts[top.indices.cpu()] / n
    assert counts.sum() == f.sum() * n, "only the top-8 ids are ever drawn"
    assert (f - p).abs().max().item() < 4 * math.sqrt(0.25 / n), f"empirical {f.tolist()} vs softmax {p.tolist()}"
    # top_k = 1: the draw is the argmax, identical to the greedy kernel
    t1, _ = ops.sample_topk_topp(logits, seen, slots, 0.01, 1, 0.001, seed=5)
    t0, _ = ops.sample_greedy(logits, seen, slots)
    assert int(t1[0]) == int(t0[0]) == int(logits[0].float().argmax())


# ---------------------------------------------------------------------------------------------------------------------
# configs[1]: LiveCC-7B shapes against the HF CPU path on identical seeded weights
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(os.environ.get("LCC_SKIP_SLOW") == "1", reason="LCC_SKIP_SLOW=1")
def test_livecc_7b_turns_match_hf_cpu_path_on_identical_weights(dev, slow_budget):
    """BASELINE.json configs[1] at the REAL shapes: the 6-frame turn (1,114-token prefill incl. the ViT on 4,368 patches) and a
    2-frame turn on the carried KV, 16 tokens each.  Native logits vs the HF CPU path on the same weights, teacher-forced along the
    native tokens, bf16 (the reference's dtype) and -- with LCC_PARITY_FP32=1, +4.5 minutes of host time -- fp32 (the truth).  The GPU
    tier runs serially under a 20-minute limit, so by default the fp32 side of THIS 32-step comparison is what `python bench.py` prints
    (`parity.err_ratio_vs_fp32`, `rms_err_ratio_vs_fp32_*`, `tokens_equal_where_decided`: driver-visible in BENCH_rNN.json), and the
    unconditional fp32 asserts of the tier are the per-layer and first-token checks of tests/test_gpu_layer_parity.py:
        |native - HF_bf16| <= 6e-2 * max|logit| at every step                                    (same bound as test_gpu_e2e)
        native token == HF's own argmax on >= 60 % of the steps (random weights: sub-ulp top-1/top-2 margins are common -- 27-30 of
        32 agree; the hard token test is test_greedy_tokens_are_exact_on_decisive_weights below)
        |native - HF_fp32| <= 1.5 * |HF_bf16 - HF_fp32| + 1e-3 * scale  per step, worst logit   (as close to the truth as the reference)
        rms over the vocabulary of (native - HF_fp32) <= 1.15 * rms(HF_bf16 - HF_fp32) at every step, <= 1.08 over all steps
        native token == fp32 argmax wherever the fp32 margin exceeds twice the bf16 reference's own error."""
    import sys
    import tempfile
    full = os.environ.get("LCC_PARITY_FP32") == "1"
    slow_budget(430 if full else 170)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    cfg = get_config("livecc-7b")
    args = bench.parse(["--cpu-baseline", "on"])
    arena = WeightArena(cfg, dev).fill_tiled(seed=0)
    model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=4096, max_new_rows=1280, max_patches=4608,
                                           max_history=16)
    ntok, nlog = bench.native_parity_turns(model, cfg, args, protocol, 2, dev)
    del model, arena
    torch.cuda.empty_cache()
    with tempfile.TemporaryDirectory() as tmp:
        teacher, o16, o32 = os.path.join(tmp, "t.npy"), os.path.join(tmp, "r16.npz"), os.path.join(tmp, "r32.npz")
        np.save(teacher, ntok)
        ev, cut = bench.run_cpu_leg("livecc-7b", args, 900.0, 2, teacher, o16)
        assert not cut and os.path.exists(o16), f"CPU reference leg did not finish: {ev[-3:]}"
        r32 = None
        if full:
            ev, cut = bench.run_cpu_leg("livecc-7b", args, 1800.0, 2, teacher, o32, dtype="float32")
            assert not cut and os.path.exists(o32), f"fp32 CPU leg did not finish: {ev[-3:]}"
            r32 = dict(np.load(o32))
        rep = bench.parity_report(ntok, nlog, dict(np.load(o16)), r32)
    record("livecc7b_vs_hf_cpu", rep)
    assert rep["turns_compared"] == 2 and rep["steps"] == 32
    assert rep["rel_dlogit_vs_bf16"] <= 6e-2, rep
    # random weights: most top-1/top-2 margins of the bf16 oracle are below its own rounding noise (27-30 of 32 steps agree, depending
    # on the fp32 summation order of the day) -- the HARD token test is the decisive-weight test below; here: a loose majority
    # (`tokens_equal_where_decided_vs_bf16` / `tokens_decided_vs_bf16` are recorded: agreement on the steps the oracle's margin decides)
    assert rep["tokens_equal"] >= 0.6 * rep["tokens_total"], rep
    if full:
        assert rep["err_ratio_vs_fp32"] <= 1.5, rep
        assert rep["rms_err_ratio_vs_fp32_worst_step"] <= 1.15 and rep["rms_err_ratio_vs_fp32_all_steps"] <= 1.08, rep
        assert rep["tokens_equal_where_decided"] == rep["tokens_decided_by_margin"], rep


@pytest.mark.skipif(os.environ.get("LCC_SKIP_SLOW") == "1", reason="LCC_SKIP_SLOW=1")
@pytest.mark.parametrize("preset", ["livecc-7b", "qwen2vl-2b-untied"])
def test_greedy_tokens_are_exact_on_decisive_weights(dev, preset, slow_budget):
    """north_star: "token-id exact under greedy".  With i.i.d. Gaussian weights the top-1/top-2 logit gap is the size of bf16's own
    noise and the clause is undecidable; on the `decisive` synthetic weights (livecc_amd/weights.py: lm_head rows = a permutation of
    the scaled embedding rows) HF's top-1 margin is tens of noise units on every step, so the clause is a hard assertion: the HF bf16
    CPU path, teacher-forced along the native tokens of the 6-frame turn + a 2-frame turn (16 tokens each), prefers the native token
    at ALL 32 steps, at LiveCC-7B and at Qwen2-VL-2B shapes (2B with an untied lm_head: the tied checkpoint cannot carry a
    permutation).
    WHAT IT DOES NOT SHOW: with a margin of 30-60 % of the logit scale this proves embedding -> residual stream ->
    final norm -> lm_head and that the layers do not blow up; a mid-network error of several x the bf16 noise would still pass.  The
    sensitive tests are the rms-ratio ones (per layer, first token, 32-step fixtures).  Opt-in since round 4 (LCC_LIVE_ORACLE=1): its
    fixture twin test_livecc7b_greedy_tokens_equal_the_committed_hf_tokens_on_decisive_weights always runs."""
    import sys
    import tempfile
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from livecc_amd import protocol
    from livecc_amd.config import get_config
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.weights import WeightArena
    slow_budget(120 if preset == "livecc-7b" else 70)
    cfg = get_config(preset)
    args = bench.parse(["--cpu-baseline", "on", "--config", preset])
    arena = WeightArena(cfg, dev).fill_tiled(seed=0, variant="decisive")
    model = LiveCCForConditionalGeneration(cfg, arena, dev, max_streams=1, max_kv_len=4096, max_new_rows=1280, max_patches=4608,
                                           max_history=16)
    ntok, nlog = bench.native_parity_turns(model, cfg, args, protocol, 2, dev)
    del model, arena
    torch.cuda.empty_cache()
    with tempfile.TemporaryDirectory() as tmp:
        teacher, o16 = os.path.join(tmp, "t.npy"), os.path.join(tmp, "r16.npz")
        np.save(teacher, ntok)
        ev, cut = bench.run_cpu_leg(preset, args, 900.0, 2, teacher, o16, weights="decisive:0")
        assert not cut and os.path.exists(o16), f"CPU reference leg did not finish: {ev[-3:]}"
        rep = bench.decisive_report(ntok, nlog, dict(np.load(o16)))
    record(f"greedy_tokens_decisive[{preset}]", rep)
    assert rep["tokens_total"] == 32 and rep["tokens_equal"] == 32, rep
    assert rep["steps_with_margin_over_10x_noise"] >= 29, rep           # >= 90 % of the steps decided by >= 10 x the measured noise
    assert len(set(ntok.reshape(-1).tolist())) >= 30, "the decisive model walks a permutation: (almost) no repeated ids"
