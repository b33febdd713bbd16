"""-m gpu: the HIP antialias-bicubic resize (lcc_resize_bicubic_aa_u8) against torch's own CPU kernel = the torchvision call of
ref livecc_utils/video_process_patch.py:150-155 (oracle/resize_ref.py).  uint8 in, uint8 out: BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,hi,wi,ho,wo", [(4, 270, 480, 112, 196), (2, 1080, 1920, 392, 728), (2, 40, 60, 56, 84), (3, 56, 84, 56, 84),
                                          (1, 720, 1280, 280, 504), (2, 113, 77, 28, 28)])
@pytest.mark.parametrize("layout", ["THWC", "TCHW"])
def test_resize_matches_torch_cpu_bit_for_bit(dev, T, hi, wi, ho, wo, layout):
    from livecc_amd import resize as R
    from oracle import resize_ref as O
    g = torch.Generator().manual_seed(hi * 7 + wi)
    tchw = torch.randint(0, 256, (T, 3, hi, wi), dtype=torch.uint8, generator=g)
    tchw[0, :, : hi // 4] = 255                       # saturated + black bands: the clamp of the bicubic overshoot
    tchw[0, :, hi // 4: hi // 2, : wi // 3] = 0
    src = tchw.permute(0, 2, 3, 1).contiguous() if layout == "THWC" else tchw
    got = R.resize_bicubic_aa(src.to(dev), ho, wo, layout).cpu()
    ref = O.resize_ref(tchw, ho, wo)
    assert got.shape == ref.shape == (T, 3, ho, wo) and got.dtype == torch.uint8
    nbad = int((got != ref).sum())
    assert nbad == 0, f"{nbad} of {ref.numel()} bytes differ (max {int((got.int() - ref.int()).abs().max())})"


def test_resize_golden_fixture(dev):
    from livecc_amd import resize as R
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_small.npz"))
    got = R.resize_bicubic_aa(torch.from_numpy(z["frames"]).to(dev), int(z["out_hw"][0]), int(z["out_hw"][1]), "TCHW").cpu().numpy()
    assert np.array_equal(got, z["resized"])


def test_smart_resized_clip_feeds_the_model_like_the_reference(dev):
    """get_smart_resized_clip on a GPU-resident 'decoded video' (uint8 THWC, 30 fps pts): frame selection as the reference's
    loop, resize bit-exact, and the clip goes straight into the ViT front end (frames path == pixel_values path of the oracle
    patchify on the same resized frames)."""
    from livecc_amd import resize as R
    from oracle import resize_ref as O
    g = torch.Generator().manual_seed(11)
    video = torch.randint(0, 256, (120, 135, 240, 3), dtype=torch.uint8, generator=g)          # 4 s at 30 fps, 135x240
    pts = np.arange(120) / 30.0
    h, w = R.smart_resized_hw(135, 240, 120, max_pixels=100 * 28 * 28)
    assert (h, w) == (224, 392)          # 135x240 is below VIDEO_MIN_PIXELS = 100*28*28: scaled UP to the minimum
    ts = np.arange(0.0, 3.0, 0.5)                                                                # the first 6-frame chunk
    clip, kept, idxs = R.get_smart_resized_clip(video.to(dev), h, w, ts, pts, 0)
    ridx, rts = O.select_clip_frames_ref(ts, pts, 0)
    assert idxs == ridx == [0, 15, 30, 45, 60, 75] and np.allclose(kept, rts)
    ref = O.resize_ref(video[ridx].permute(0, 3, 1, 2).contiguous(), h, w)
    assert torch.equal(clip.cpu(), ref)
    # past the end of the video: nothing to fetch
    clip2, _, idxs2 = R.get_smart_resized_clip(video.to(dev), h, w, [10.0, 10.5], pts, 100)
    assert clip2 is None and idxs2 == []


def test_live_cc_from_video_paces_like_the_reference(dev):
    """The reference's due-frame logic on a GPU-resident 30 fps video: nothing before 3 s of video exist, then the 6-frame
    chunk, then one 2-frame chunk per second of wall clock, `video_end` once the stream is exhausted."""
    from livecc_amd.config import tiny
    from livecc_amd.infer import LiveCCDemoInfer
    from livecc_amd.modeling import LiveCCForConditionalGeneration
    cfg = tiny()
    native = LiveCCForConditionalGeneration.from_config(cfg, dev, seed=0, max_streams=1, max_kv_len=2048, max_new_rows=512,
                                                        max_patches=2048, max_history=16)
    infer = LiveCCDemoInfer(model=native)
    g = torch.Generator().manual_seed(5)
    video = torch.randint(0, 256, (180, 60, 90, 3), dtype=torch.uint8, generator=g).to(dev)      # 6 s at 30 fps
    pts = np.arange(180) / 30.0
    from livecc_amd.video import DecodedVideo
    state, seen = {"video_path": DecodedVideo(video, pts, 30.0)}, []
    for wall in (0.4, 3.2, 3.3, 4.1, 5.6, 6.0, 7.0, 8.0):
        state["video_timestamp"] = wall                                  # the reference's calling convention (ref demo/cli.py:13-19)
        for (a, b), text, state in infer.live_cc(None, state, max_pixels=4 * 28 * 28, max_new_tokens=3, force_length=True,
                                                 do_sample=False):
            seen.append((wall, a, b))
    # wall 0.4: the first call waits for initial_time_interval (3 s) of video -> frames 0..2.5 s (ref infer.py:107-110)
    assert seen[0][1:] == (0.0, 3.0) and seen[0][0] == 0.4
    spans = [s[1:] for s in seen]
    assert spans[:4] == [(0.0, 3.0), (3.0, 4.0), (4.0, 5.0), (5.0, 6.0)][:len(spans[:4])]
    assert all(b - a in (1.0, 3.0) for a, b in spans)
    assert state.get("video_end") is True
    state["past_key_values"].release()
