"""Shared helpers of the parity tests (reference arithmetic in fp32 torch with HF's bf16 rounding points)."""
import json
import os

import torch

# LCC_PARITY_OUT: bench.py runs single fixture tests as a subprocess and reads their record from a scratch directory
OUT = os.environ.get("LCC_PARITY_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def rb(x: torch.Tensor) -> torch.Tensor:
    """round an fp32 tensor to bf16 and back (one bf16 rounding point)."""
    return x.to(torch.bfloat16).to(torch.float32)


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a-b| in units of the bf16 ulp of max(|a|,|b|) (elements are bf16 values held in any float dtype)."""
    a, b = a.float(), b.float()
    mag = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    return (a - b).abs() / ulp


def assert_bf16_close(got: torch.Tensor, ref: torch.Tensor, name: str, max_ulp: float = 1.0, max_frac: float = 2e-3,
                      atol=0.0):
    """Two bf16 tensors computed with fp32 accumulation in different orders agree except for rare 1-ulp rounding
    flips: every element within `max_ulp` bf16 ulps (or within `atol`, a float or a per-element tensor -- for dot
    products the fp32 summation-order noise is ~eps32 * sum|a_k b_k|, which exceeds one bf16 ulp of the result when
    the sum cancels to nearly zero), and at most `max_frac` of the elements differ at all."""
    got, ref = got.float().cpu(), ref.float().cpu()
    if torch.is_tensor(atol):
        atol = atol.float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    d = bf16_ulp_diff(got, ref)
    bad = (d > max_ulp) & ((got - ref).abs() > atol)
    frac = ((got != ref) & ((got - ref).abs() > atol)).float().mean().item()
    record(name, dict(max_ulp=float(d.max()), frac_diff=frac, max_abs=float((got - ref).abs().max())))
    assert not bad.any(), (f"{name}: {int(bad.sum())} elements differ by more than {max_ulp} bf16 ulp; worst "
                           f"{float(d.max()):.2f} ulp, abs {float((got - ref).abs().max()):.4g}")
    assert frac <= max_frac, f"{name}: {frac:.2e} of the elements differ (> {max_frac:.1e})"


_REC = {}


def record(name, d):
    _REC[name] = d
    try:
        os.makedirs(OUT, exist_ok=True)
        w = os.environ.get("PYTEST_XDIST_WORKER")          # pytest -n N: one report per worker process (merged by the caller)
        with open(os.path.join(OUT, f"parity_report_{w}.json" if w else "parity_report.json"), "w") as f:
            json.dump(_REC, f, indent=1, sort_keys=True, default=float)
    except OSError:
        pass
