"""Frame selection in front of the hot path (ref livecc_utils/video_process_patch.py:24-156) against golden
vectors produced by EXECUTING the reference's own functions with stubbed decoders (oracle/make_frame_selection_golden.py ->
tests/golden/frame_selection.json).  Host integer logic only: runs without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from livecc_amd import resize as R, video as V
from oracle.make_frame_selection_golden import digest, make_videos

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_selection.json")))
VIDEOS = make_videos()


def test_constants_match_the_reference_configuration():
    c = GOLD["constants"]
    assert (R.FPS_MAX_FRAMES, R.VIDEO_MIN_PIXELS, R.VIDEO_MAX_PIXELS, R.VIDEO_TOTAL_PIXELS, R.FRAME_FACTOR, R.FPS) == (
        c["FPS_MAX_FRAMES"], c["VIDEO_MIN_PIXELS"], c["VIDEO_MAX_PIXELS"], c["VIDEO_TOTAL_PIXELS"], c["FRAME_FACTOR"], c["FPS"])


@pytest.mark.parametrize("case", GOLD["read"], ids=lambda c: f"{c['video']}-{sorted(c['ele'].items())}-strict{int(c['strict_fps'])}-drop{int(c['drop_last'])}")
def test_read_video_decord_plus_index_selection(case):
    v = VIDEOS[case["video"]]
    vid = V.DecodedVideo(torch.zeros(len(v["pts"]), 2, 2, 3, dtype=torch.uint8), v["pts"], v["fps"])
    ele = dict(case["ele"], video=vid)
    if "error" in case:
        with pytest.raises(Exception) as ei:
            V.read_video_decord_plus(ele, strict_fps=case["strict_fps"], drop_last=case["drop_last"], return_pts=True)
        assert type(ei.value).__name__ == case["error"]
        return
    idxs, pts, total = V.select_video_frames(vid.pts, len(vid), vid.avg_fps, ele, case["strict_fps"], case["drop_last"])
    assert idxs == case["idxs"]
    clip, sample_fps, pts2 = V.read_video_decord_plus(ele, strict_fps=case["strict_fps"], drop_last=case["drop_last"], return_pts=True)
    assert clip.shape[0] == case["n"] and clip.shape[1] == 3
    assert sample_fps == pytest.approx(case["sample_fps"], rel=1e-12)
    if case["strict_fps"] or case["pts"] is None:
        assert digest(pts2) == case["pts"]
    # (the reference's linspace branch returns the un-sampled pts of the cut range; nothing in the hot path consumes them)


@pytest.mark.parametrize("case", GOLD["spatial"], ids=lambda c: f"{c['h']}x{c['w']}-n{c['nframes']}")
def test_spatial_resize_budget(case):
    n = case["nframes"] if case["nframes"] is not None else case["T"]
    assert list(V.spatial_resize_hw(case["h"], case["w"], n)) == case["size"]


@pytest.mark.parametrize("case", GOLD["reader"], ids=lambda c: f"{c['video']}-{c['max_pixels']}")
def test_smart_resized_video_reader_size(case):
    v = VIDEOS[case["video"]]
    vid = V.DecodedVideo(torch.zeros(len(v["pts"]), v["h"], v["w"], 3, dtype=torch.uint8).expand(len(v["pts"]), -1, -1, -1)
                         if False else torch.empty(len(v["pts"]), 0, 0, 3, dtype=torch.uint8), v["pts"], v["fps"])
    # the size only depends on (h, w, number of frames): avoid allocating the pixels
    assert list(R.smart_resized_hw(v["h"], v["w"], len(vid), case["max_pixels"])) == case["size"]


@pytest.mark.parametrize("case", GOLD["clip"], ids=lambda c: f"{c['video']}-{c['timestamps'][0]}-{len(c['timestamps'])}")
def test_streaming_clip_timestamp_scan(case):
    idxs, ts = R.select_clip_frames(case["timestamps"], VIDEOS[case["video"]]["pts"], case["index_from"])
    assert idxs == case["idxs"]
    assert ts == pytest.approx(case["timestamps_out"], abs=0)


def test_registry_and_loader():
    vid = V.DecodedVideo(torch.zeros(4, 2, 2, 3, dtype=torch.uint8), [0.0, 0.5, 1.0, 1.5])
    assert vid.avg_fps == pytest.approx(2.0)
    V.register_video("mem://x.mp4", vid)
    try:
        assert V.open_video("mem://x.mp4") is vid
        with pytest.raises(ValueError, match="not found"):
            V.open_video("mem://missing.mp4")
        V.set_video_loader(lambda p: vid)
        assert V.open_video("mem://missing.mp4") is vid
    finally:
        V.set_video_loader(None)
        V.unregister_video("mem://x.mp4")


def test_smart_nframes_rules():
    assert V.smart_nframes({}, total_frames=300, video_fps=30.0) == 20
    assert V.smart_nframes({"nframes": 15}, 300, 30.0) == 16
    assert V.smart_nframes({}, 18000, 29.97) == R.FPS_MAX_FRAMES
    assert V.smart_nframes({}, 36, 30.0) == 4                     # FPS_MIN_FRAMES
    with pytest.raises(ValueError):
        V.smart_nframes({"fps": 2.0, "nframes": 4}, 300, 30.0)
    with pytest.raises(ValueError):
        V.smart_nframes({}, 1, 30.0)
