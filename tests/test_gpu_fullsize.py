"""-m gpu: BASELINE.json's full LiveCC-7B shapes (random weights of the real architecture; the CPU oracle cannot run them in
test time) checked through size-independent properties of the path:

  * determinism: the same stream replayed twice gives the same tokens and bit-identical logits;
  * KV carry: prefilling a prompt in one call or in two calls (cache reuse) gives the same next-token logits -- the in-call
    causal attention and the attention over the carried cache must agree;
  * decode == prefill: the logits of step k computed by the weight-streaming decode path (skinny GEMVs, decode attention)
    equal those of a prefill call that feeds the same token as a 1-row prompt extension (tiled GEMM path is not used for one
    row, but the positions / cache bookkeeping are) and, for a >16-row extension, the tiled GEMMs + prefill attention;
  * batch invariance: two streams advanced in one batched call give the logits each gives alone.
Tolerance: 6 % of the logit scale, the bound the end-to-end tests use between the native path and the bf16 oracle -- two
bf16 evaluation orders of a 28-layer network (skinny GEMV + decode attention vs tiled GEMM + prefill attention) differ by about
twice one path's own rounding error (measured: KV carry in one vs two calls bit-identical, decode vs prefill 4.0 %, batched (480-row prefill) vs alone 3.3 %)."""
import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(__import__("os").environ.get("LCC_SKIP_SLOW") == "1", reason="LCC_SKIP_SLOW=1")]


@pytest.fixture(scope="module")
def m7b(dev):
    from livecc_amd.config import livecc_7b
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    cfg = livecc_7b()
    return cfg, LiveCCForConditionalGeneration.from_config(cfg, dev, seed=0, max_streams=2, max_kv_len=4096, max_new_rows=2048,
                                                           max_patches=8192, max_history=16)


def _close(a, b, tol=0.06):
    a, b = a.float(), b.float()
    scale = b.abs().max().item()
    return (a - b).abs().max().item() <= tol * scale, (a - b).abs().max().item() / scale


def test_determinism_and_kv_carry_at_7b_shapes(dev, m7b):
    from livecc_amd import protocol
    cfg, model = m7b
    frames = torch.from_numpy(protocol.synth_frames(6, 392, 728, seed=1234, layout="THWC")).to(dev)
    grid = protocol.grid_of(6, 392, 728, cfg)
    ids = protocol.TurnBuilder(cfg, seed=1234).turn_ids(0, protocol.num_video_tokens(grid, cfg))
    assert len(ids) > 1100
    kw = dict(frames=frames, frames_layout="THWC", repetition_penalty=1.05, max_new_tokens=6, min_new_tokens=6, output_logits=True)
    a = model.generate(input_ids=torch.from_numpy(ids).view(1, -1), **kw)
    a.past_key_values.release()
    b = model.generate(input_ids=torch.from_numpy(ids).view(1, -1), **kw)
    assert torch.equal(a.sequences, b.sequences) and torch.equal(a.logits, b.logits), "same input, different output"
    toks = b.sequences[0, len(ids):].cpu().numpy()
    # KV carry: extend the finished turn by a text-only suffix in ONE call ...
    suffix = np.asarray([cfg.eos_token_id] + list(range(100, 140)), dtype=np.int64)        # 41 rows: tiled GEMMs + prefill attention
    past = np.concatenate([ids, toks[:-1]])
    full = np.concatenate([past, suffix])
    one = model.generate(input_ids=torch.from_numpy(full).view(1, -1), past_key_values=b.past_key_values, repetition_penalty=1.0,
                         max_new_tokens=1, output_logits=True)
    lg_one = one.logits[0].clone()
    # ... and in TWO calls on a second stream that replays the same history (first 20 suffix rows, then the other 21)
    c = model.generate(input_ids=torch.from_numpy(ids).view(1, -1), **kw)
    assert torch.equal(c.sequences, b.sequences)
    h1 = np.concatenate([past, suffix[:20]])
    two_a = model.generate(input_ids=torch.from_numpy(h1).view(1, -1), past_key_values=c.past_key_values, repetition_penalty=1.0,
                           max_new_tokens=1, output_logits=True)
    # roll the generated token back (the reference's past_ids = sequences[:, :-1]) and extend with the rest
    two_b = model.generate(input_ids=torch.from_numpy(full).view(1, -1), past_key_values=two_a.past_key_values, repetition_penalty=1.0,
                           max_new_tokens=1, output_logits=True)
    ok, err = _close(two_b.logits[0], lg_one)
    record("fullsize_7b_kv_carry_one_vs_two_calls", dict(rel_dlogit=err))
    assert ok, f"one-call vs two-call prefill logits differ by {err:.4f} of the scale"
    one.past_key_values.release()
    two_b.past_key_values.release()


def test_decode_step_equals_single_row_prefill_and_batch_invariance_at_7b_shapes(dev, m7b):
    from livecc_amd import protocol
    cfg, model = m7b
    rng = np.random.RandomState(3)
    ids_a = rng.randint(0, 150000, size=300).astype(np.int64)
    ids_b = rng.randint(0, 150000, size=180).astype(np.int64)
    kw = dict(repetition_penalty=1.0, output_logits=True)
    ra = model.generate(input_ids=torch.from_numpy(ids_a).view(1, -1), max_new_tokens=4, min_new_tokens=4, **kw)
    seq = ra.sequences[0].cpu().numpy()
    # step 2's logits came from the decode path (token 1 fed back).  Recompute them with a prefill call on a fresh stream that
    # gets prompt + token 1 as its prompt.
    rb = model.generate(input_ids=torch.from_numpy(seq[:len(ids_a) + 1]).view(1, -1), max_new_tokens=1, **kw)
    ok, err = _close(ra.logits[1], rb.logits[0])
    record("fullsize_7b_decode_step_vs_prefill", dict(rel_dlogit=err))
    assert ok, f"decode-step logits vs prefill logits differ by {err:.4f} of the scale"
    assert int(torch.argmax(ra.logits[1])) == int(torch.argmax(rb.logits[0])) or \
        (torch.topk(rb.logits[0].float(), 2).values.diff().abs().item() < 2 * err * rb.logits[0].float().abs().max().item())
    ra.past_key_values.release()
    rb.past_key_values.release()
    # batch invariance: both prompts in one batched call vs alone
    alone_a = model.generate(input_ids=torch.from_numpy(ids_a).view(1, -1), max_new_tokens=3, min_new_tokens=3, **kw)
    alone_a.past_key_values.release()
    alone_b = model.generate(input_ids=torch.from_numpy(ids_b).view(1, -1), max_new_tokens=3, min_new_tokens=3, **kw)
    alone_b.past_key_values.release()
    oa, ob = model.generate_batch([dict(input_ids=torch.from_numpy(ids_a)), dict(input_ids=torch.from_numpy(ids_b))], repetition_penalty=1.0,
                                  max_new_tokens=3, force_length=True, output_logits=True)
    for got, ref in ((oa, alone_a), (ob, alone_b)):
        ok, err = _close(got.logits[0], ref.logits[0])
        record("fullsize_7b_batched_vs_alone", dict(rel_dlogit=err))
        assert ok, f"batched vs alone first-token logits differ by {err:.4f}"
    oa.past_key_values.release()
    ob.past_key_values.release()
