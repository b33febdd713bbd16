"""The drop-in boundary, demonstrated by EXECUTING the reference (VERDICT r3 missing #2): `ref demo/infer.py` itself -- loaded from
/root/reference by path in a subprocess with liger / decord / torchvision / qwen_vl_utils stood in for (oracle/ref_infer_harness.py) --
runs `live_cc_once_for_evaluation` (ref :244-310) and the `demo/cli.py` loop over `live_cc` (ref :61-180) on a recording model, and
`livecc_amd.infer.LiveCCDemoInfer` runs the same two flows over the same model, tokenizer and synthetic video: the two orchestrators
must make IDENTICAL `generate` calls and return identical responses.  (CPU tier, build container only: /root/reference does not exist
on the GPU box -- there the committed trace of the executed reference over HF's model is replayed through the native engine,
tests/test_gpu_golden.py::test_reference_orchestrator_trace_through_the_native_engine.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_infer_harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not H.available(), reason="needs /root/reference (the build container)")


def run_reference(mode, out):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_infer_harness.py"), "--mode", mode, "--out", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with open(out) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def reference_trace(tmp_path_factory):
    return run_reference("recording", tmp_path_factory.mktemp("ref") / "recording.json")


@pytest.fixture()
def ours(monkeypatch):
    """`livecc_amd.infer.LiveCCDemoInfer` over the SAME recording model class; its GPU resize is replaced by the CPU reference of the
    same arithmetic (oracle/resize_ref.py; the HIP kernel is bit-exact against it: tests/test_gpu_resize.py)."""
    from livecc_amd import infer as I, resize as R, video as V
    from livecc_amd.text import TextFrontEnd
    from oracle.resize_ref import resize_ref
    tk = H.load_tokenizer()
    cfg = H.model_config(tk)

    def cpu_resize(clip, h, w, layout="TCHW", device=None):
        c = clip if layout == "TCHW" else clip.permute(0, 3, 1, 2)
        return resize_ref(c.contiguous(), h, w)
    monkeypatch.setattr(R, "resize_bicubic_aa", cpu_resize)
    model = H.RecordingModel(cfg)
    inf = I.LiveCCDemoInfer(model=model, text=TextFrontEnd(tk, cfg))
    V.register_video(H.VIDEO_NAME, H.make_video())
    yield inf, model
    V.unregister_video(H.VIDEO_NAME)


def assert_same_calls(ref_calls, our_calls):
    assert len(ref_calls) == len(our_calls), (len(ref_calls), len(our_calls))
    for i, (r, o) in enumerate(zip(ref_calls, our_calls)):
        assert r["input_ids"] == o["input_ids"], f"call {i}: input_ids differ ({len(r['input_ids'])} vs {len(o['input_ids'])} ids)"
        assert r["grid"] == o["grid"], f"call {i}: {r['grid']} vs {o['grid']}"
        assert r["pixel_sha"] == o["pixel_sha"] and r["pixel_rows"] == o["pixel_rows"], f"call {i}: pixel values differ"
        assert r["past_len"] == o["past_len"], f"call {i}: cache hand-over"
        for k in ("repetition_penalty", "max_new_tokens", "pad_token_id", "logits_processor", "tokens"):
            assert r[k] == o[k], f"call {i}: {k}: {r[k]} vs {o[k]}"
        # do_sample: the reference passes nothing in live_cc_once_for_evaluation (generation_config decides) = None on both sides
        assert r["do_sample"] == o["do_sample"], f"call {i}: do_sample {r['do_sample']} vs {o['do_sample']}"


def test_reference_constructor_calls_what_livecc_amd_serves(reference_trace):
    """ref demo/infer.py:35-59 executed: the from_pretrained arguments the native `from_pretrained` accepts, the ' ...' token and the
    system-prompt offset."""
    import inspect
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    from livecc_amd.text import TextFrontEnd
    c = reference_trace["created"]
    assert c["torch_dtype"] == "auto" and c["device_map"] == "cpu" and c["use_fast"] == "False"
    inspect.signature(LiveCCForConditionalGeneration.from_pretrained).bind("path", torch_dtype="auto", device_map="cuda", attn_implementation="flash_attention_2")
    fe = TextFrontEnd(H.load_tokenizer())
    assert reference_trace["streaming_eos_token_id"] == fe.streaming_eos_token_id
    assert reference_trace["system_prompt_offset"] == fe.system_prompt_offset


def test_live_cc_once_for_evaluation_makes_the_reference_calls(reference_trace, ours):
    inf, model = ours
    responses = H.scenario_once(inf)
    ref = reference_trace["once"]
    assert len(ref["calls"]) >= 4
    assert_same_calls(ref["calls"], model.calls)
    assert [[float(a), float(b), t] for a, b, t in responses] == ref["responses"]
    assert ref["calls"][0]["past_len"] == 0 and all(c["past_len"] > 0 for c in ref["calls"][1:])


def test_live_cc_cli_loop_makes_the_reference_calls(reference_trace, ours):
    inf, model = ours
    responses, state = H.scenario_live(inf)
    ref = reference_trace["live"]
    assert len(ref["calls"]) >= 8
    assert_same_calls(ref["calls"], model.calls)
    assert responses == ref["responses"]
    assert {k: (float(v) if isinstance(v, (int, float, np.floating)) and not isinstance(v, bool) else v) for k, v in state.items()} == \
           {k: (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v) for k, v in ref["state"].items()}
    # the threshold processor object is fresh per call (ref :161-164), the query changes once mid-stream (ref :144-146)
    assert all(c["logits_processor"] and c["logits_processor"][0]["count"] == 0 for c in ref["calls"])
    texts = H.load_tokenizer().decode(ref["calls"][-1]["input_ids"], skip_special_tokens=False)
    assert texts.count(H.QUERY) == 1 and texts.count("and now?") == 1 and texts.count("<|im_start|>system") == 1


def test_generate_signature_binds_every_argument_the_reference_passes(reference_trace):
    """What ref demo/infer.py:165-172, 230-236, 297-302 pass to `generate` binds to the native model's signature (no **kwargs catch-all is
    relied upon for these names)."""
    import inspect
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    sig = inspect.signature(LiveCCForConditionalGeneration.generate)
    for name in ("input_ids", "pixel_values_videos", "video_grid_thw", "attention_mask", "past_key_values", "return_dict_in_generate", "do_sample",
                 "repetition_penalty", "logits_processor", "max_new_tokens", "pad_token_id"):
        assert name in sig.parameters, name
    assert all(c["extra"] == [] for c in reference_trace["once"]["calls"] + reference_trace["live"]["calls"])


def test_committed_trace_of_the_reference_over_hf_is_reproducible(tmp_path):
    """tests/golden/ref_infer_trace.json = the executed reference over HF's own model (tiny shapes, CPU, bf16): regenerated here and
    compared call by call (ids, tokens, responses) -- pins the fixture the GPU tier replays through the native engine."""
    with open(H.TRACE_PATH) as f:
        want = json.load(f)
    got = run_reference("hf", tmp_path / "hf.json")
    for sc in ("once", "live"):
        assert got[sc]["responses"] == want[sc]["responses"], sc
        assert [c["input_ids"] for c in got[sc]["calls"]] == [c["input_ids"] for c in want[sc]["calls"]], sc
        assert [c["tokens"] for c in got[sc]["calls"]] == [c["tokens"] for c in want[sc]["calls"]], sc
        margins = [min((a - b) / s for (a, b), s in zip(c["top2_vals"], c["scale"])) for c in want[sc]["calls"]]
        assert min(margins) >= 0.05, f"{sc}: HF's own top-1 margin must decide every token of the fixture ({min(margins):.3f} of the scale)"
