"""Host logic of the frame fetch -> resize front end against torch's own CPU kernel and the reference's loop:
tap tables bit for bit, the two-pass fp32 accumulation order, the pixel-budget policy, the timestamp -> frame mapping."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from livecc_amd import resize as R
from oracle import resize_ref as O


@pytest.mark.parametrize("n_in,n_out", [(160, 56), (1920, 728), (1080, 392), (60, 84), (720, 392), (1280, 728), (84, 84), (37, 5)])
def test_tap_tables_equal_atens_bit_for_bit(n_in, n_out):
    """A one-hot row through torch's kernel reads its weights out exactly."""
    eye = torch.eye(n_in).view(1, n_in, 1, n_in)
    w_torch = F.interpolate(eye, size=(1, n_out), mode="bicubic", align_corners=False, antialias=True)[0, :, 0, :].numpy()   # [src, out]
    xmin, xsize, W = R.aa_bicubic_taps(n_in, n_out)
    mine = np.zeros((n_in, n_out), np.float32)
    for i in range(n_out):
        mine[xmin[i]:xmin[i] + xsize[i], i] = W[i, :xsize[i]]
        assert xsize[i] <= W.shape[1]
    diff = mine != w_torch
    # identical bit for bit; the only tolerated deviation is a tap at the very edge of the support, where the cubic cancels to
    # ~1e-15 (it cannot change an fp32 sum of 8-bit samples): seen once, for the extreme 37 -> 5 reduction
    assert diff.sum() <= 1 and np.abs(mine - w_torch).max() <= 1e-12, f"{int(diff.sum())} weights differ, max {np.abs(mine - w_torch).max()}"
    if (n_in, n_out) != (37, 5):
        assert not diff.any()


def _pass(x, xmin, xsize, W):
    out = np.zeros(x.shape[:-1] + (len(xmin),), np.float32)
    for i in range(len(xmin)):
        t = x[..., xmin[i]] * W[i, 0]
        for j in range(1, xsize[i]):
            t = (x[..., xmin[i] + j].astype(np.float64) * np.float64(W[i, j]) + t.astype(np.float64)).astype(np.float32)   # fma
        out[..., i] = t
    return out


@pytest.mark.parametrize("hi,wi,ho,wo", [(90, 160, 28, 56), (40, 60, 56, 84), (108, 192, 56, 84)])
def test_width_then_height_fma_accumulation_is_atens_order(hi, wi, ho, wo):
    """The arithmetic the HIP kernels implement (width pass, then height pass, t = s0*w0 then fma) is bit-identical to torch's
    float result -- so the uint8 outputs cannot differ even at rounding ties."""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (2, 3, hi, wi), dtype=torch.uint8, generator=g)
    ref = O.resize_float_ref(x, ho, wo).numpy()
    a = _pass(x.numpy().astype(np.float32), *R.aa_bicubic_taps(wi, wo))
    b = _pass(np.ascontiguousarray(a.transpose(0, 1, 3, 2)), *R.aa_bicubic_taps(hi, ho)).transpose(0, 1, 3, 2)
    assert np.array_equal(b, ref)
    assert np.array_equal(np.rint(np.clip(b, 0, 255)).astype(np.uint8), O.resize_ref(x, ho, wo).numpy())


def test_pixel_budget_policy():
    # demo/cli.py default max_pixels = 384*28*28 on a 1080p video -> the G1 geometry of SURVEY section 8
    assert R.smart_resized_hw(1080, 1920, 3000, max_pixels=384 * 28 * 28) == (392, 728)
    # README's 480-frame regime: budget max(min(768*784, 24576*784/480*2), 1.05*100*784) = 82320 px per frame
    h, w = R.smart_resized_hw(1080, 1920, 480)
    assert h % 28 == 0 and w % 28 == 0 and h * w <= 82320 < (h + 28) * (w + 28) * 1.3
    assert R.smart_resized_hw(280, 280, 480) == (280, 280)
    # short clips: the per-frame cap is VIDEO_MAX_PIXELS = 768*28*28
    h, w = R.smart_resized_hw(2160, 3840, 8)
    assert h * w <= 768 * 28 * 28


def test_timestamp_to_frame_mapping_follows_the_reference_loop():
    rng = np.random.RandomState(0)
    for trial in range(200):
        n = rng.randint(1, 60)
        pts = np.cumsum(rng.uniform(0.01, 0.2, size=n))
        t0 = rng.uniform(0, pts[-1] * 1.1)
        k = rng.randint(1, 9)
        ts = t0 + 0.5 * np.arange(k)
        start = rng.randint(0, n)
        got = R.select_clip_frames(ts, pts, start)
        ref = O.select_clip_frames_ref(ts, pts, start)
        assert got[0] == ref[0] and np.allclose(got[1], ref[1]), (trial, got, ref)
        assert len(got[0]) % 2 == 0
    # exactly 30 fps, the common case: timestamps 0.0, 0.5, ... hit frames 0, 15, ...
    pts = np.arange(300) / 30.0
    assert R.select_clip_frames([0.0, 0.5, 1.0, 1.5, 2.0, 2.5], pts, 0)[0] == [0, 15, 30, 45, 60, 75]
    assert R.select_clip_frames([3.0], pts, 76)[0] == [90, 105]                 # padded to an even count with +1/FPS


def test_golden_fixture_pins_the_oracle():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_small.npz"))
    got = O.resize_ref(torch.from_numpy(z["frames"]), int(z["out_hw"][0]), int(z["out_hw"][1])).numpy()
    assert np.array_equal(got, z["resized"])
