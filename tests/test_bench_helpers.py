"""CPU: the arithmetic behind bench.py's extra objects (no GPU, no model): algorithmic flops of the MFMA-bound phases (SURVEY 8d formulas),
the reference-noise-floor report and the decode-step roofline."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from livecc_amd import protocol  # noqa: E402
from livecc_amd.config import get_config  # noqa: E402


def test_stream_phase_flops_follow_the_survey_formulas():
    """One 6-frame turn = the first-turn numbers of SURVEY 8a: ~2.18 TFLOP of vision tower per 2-frame slice x 3, 5.04 TFLOP + attention for a
    1,131-row prefill; a one-shot clip counted as ONE turn equals `oneshot_flops` (same formulas, different call site)."""
    cfg = get_config("livecc-7b")
    f = bench.stream_phase_flops(cfg, 6, 392, 728, 15, protocol)
    n = 28 * 52
    vit_slice = 2 * n * (1.505e6 + 32 * 19.67e6) + 32 * 4 * n * n * 1280 + 2 * (n // 4) * 44.57e6
    assert abs(f["vit_flops"] / (3 * vit_slice) - 1.0) < 0.01
    S = f["prefill_rows"]
    assert S == 1131 and f["final_kv"] == 1131 + 15
    llm = 28 * (2 * 233.06e6 * S + 4 * S * (S / 2) * 3584) + 2 * 152064 * 3584
    assert abs(f["llm_prefill_flops"] / llm - 1.0) < 0.01
    # 60 frames: the cache grows by the prompt rows + 15 kept tokens per turn; flops grow with it
    g = bench.stream_phase_flops(cfg, 60, 392, 728, 15, protocol)
    assert g["final_kv"] == g["prefill_rows"] + 28 * 15 and g["llm_prefill_flops"] > 28 * f["llm_prefill_flops"] / 3
    g0 = bench.stream_phase_flops(cfg, 60, 392, 728, 0, protocol)
    assert g0["vit_flops"] == g["vit_flops"] and g0["llm_prefill_flops"] < g["llm_prefill_flops"]


def test_reference_noise_floor_report():
    rng = np.random.default_rng(0)
    la = rng.normal(size=(2, 4, 1000)).astype(np.float32)
    lb = la + rng.normal(scale=0.01, size=la.shape).astype(np.float32)
    own_a, own_b = la.argmax(-1), lb.argmax(-1)
    r = bench.hf_vs_hf_report(dict(logits=la, own_argmax=own_a), dict(logits=lb, own_argmax=own_b), own_a.copy(), "test")
    assert r["steps"] == 8 and 0 < r["rel_dlogit_hf_vs_hf"] < 0.05 and r["tokens_total"] == 8
    assert r["tokens_equal_hf_vs_hf"] == int((own_a == own_b).sum()) and r["native_tokens_equal_run_a"] == 8
    same = bench.hf_vs_hf_report(dict(logits=la, own_argmax=own_a), dict(logits=la, own_argmax=own_a), own_a, "identical")
    assert same["rel_dlogit_hf_vs_hf"] == 0.0 and same["tokens_equal_hf_vs_hf"] == 8


def test_decode_step_roofline_bytes():
    """bytes per step = weights once + per stream (mean KV length + 1 new row) x 57,344 + fp32 logits."""
    cfg = get_config("livecc-7b")

    class Eng:
        def profile_read_steps(self, n):
            return np.asarray([3.0, 3.0, 2.0, 2.0], dtype=np.float32)          # ms

        def profile_read_step_index(self, n):
            return np.asarray([0, 4, 8, 12])
    r = bench.decode_step_roofline(cfg, Eng(), 8, [1000, 3000], False)
    want = cfg.decode_weight_bytes() + 8 * (2000 * cfg.kv_bytes_per_token + cfg.kv_bytes_per_token + cfg.vocab_size * 4)
    assert r["algorithmic_bytes_per_step"] == want and r["avg_step_us"] == 2500.0
    assert abs(r["frac"] - want / 2.5e-3 / 8e12) < 1e-3
    assert r["late_steps_without_vision_tower_overlap"]["avg_step_us"] == 2000.0
    assert cfg.kv_bytes_per_token == 57344 and abs(cfg.decode_weight_bytes() / 14.14e9 - 1.0) < 0.01
