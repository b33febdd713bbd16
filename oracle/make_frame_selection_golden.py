"""ORACLE / TEST INFRASTRUCTURE: golden vectors of the reference's frame selection, produced by EXECUTING the reference's
own `livecc_utils/video_process_patch.py` (imported from /root/reference by path) with its C++/third-party dependencies
stubbed:

  * `decord.VideoReader`              -> a fake reader over synthetic pts tables that records the frame indices requested
  * `torchvision.transforms`          -> `functional.resize` records the requested [h, w] and returns zeros
  * `qwen_vl_utils.vision_process`    -> absent here and un-vendored; its constants / `smart_nframes` / `smart_resize` are the
                                         restatements of livecc_amd (video.py / protocol.py), so the golden vectors pin the
                                         REFERENCE's control flow around them (range cut by pts, linspace / strict-fps sampling,
                                         FPS_MAX_FRAMES cap, even-count padding, per-nframes pixel budget, timestamp -> frame scan)

Run in the build container (needs /root/reference):  python oracle/make_frame_selection_golden.py
Writes tests/golden/frame_selection.json, checked by tests/test_video.py on every machine (no reference needed there).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/livecc-utils/src/livecc_utils/video_process_patch.py"

VIDEOS = {}          # path -> dict(pts, fps, h, w)
LOG = {}             # last get_batch indices / resize sizes


def make_videos():
    rng = np.random.RandomState(7)
    v = {}
    v["a_30fps_10s"] = dict(pts=np.arange(300) / 30.0, fps=30.0, h=1080, w=1920)
    p = np.arange(250) / 25.0 + rng.uniform(0, 0.01, 250)
    v["b_25fps_jitter"] = dict(pts=np.sort(p), fps=25.0, h=720, w=1280)
    v["c_60fps_20s"] = dict(pts=np.arange(1200) / 60.0 + 0.033, fps=60.0, h=360, w=640)
    v["d_30fps_600s"] = dict(pts=np.arange(18000) / 29.97, fps=29.97, h=1080, w=1920)
    v["e_short"] = dict(pts=np.arange(36) / 30.0, fps=30.0, h=480, w=480)
    v["f_2fps"] = dict(pts=np.arange(121) / 2.0, fps=2.0, h=392, w=728)
    return v


class _Batch:
    def __init__(self, arr):
        self._a = arr

    def asnumpy(self):
        return self._a


class FakeVideoReader:
    def __init__(self, path, num_threads=0):
        self.spec = VIDEOS[path]
        self._frame_pts = None

    def __len__(self):
        return len(self.spec["pts"])

    def get_avg_fps(self):
        return self.spec["fps"]

    def get_frame_timestamp(self, i):
        p = self.spec["pts"]
        self._frame_pts = np.stack([p, p], axis=1)
        return self._frame_pts[i]

    def next(self):
        return np.zeros((self.spec["h"], self.spec["w"], 3), np.uint8)

    def get_batch(self, idxs):
        LOG["batch"] = [int(i) for i in idxs]
        return _Batch(np.zeros((len(idxs), 4, 4, 3), np.uint8) if not LOG.get("real_hw") else
                      np.zeros((len(idxs), self.spec["h"], self.spec["w"], 3), np.uint8))


def digest(pts):
    """pts lists can be the whole cut range (the reference does not re-index them on the linspace branch): keep a digest."""
    if pts is None:
        return None
    a = np.asarray(pts, dtype=np.float64).reshape(-1)
    if len(a) <= 64:
        return [float(x) for x in a]
    return dict(n=int(len(a)), first=float(a[0]), last=float(a[-1]), sum=float(a.sum()))


def install_stubs():
    from livecc_amd import protocol, resize as R, video as V
    decord = types.ModuleType("decord")
    decord.VideoReader = FakeVideoReader
    sys.modules["decord"] = decord

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode:
        BICUBIC = "bicubic"

    def resize(video, size, interpolation=None, antialias=None):
        LOG["resize"] = [int(size[0]), int(size[1])]
        return torch.zeros(video.shape[0], video.shape[1], 1, 1, dtype=video.dtype)
    fn.resize = resize
    tr.functional = fn
    tr.InterpolationMode = InterpolationMode
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn})

    # qwen_vl_utils reads the VIDEO_MAX_PIXELS environment variable WHEN IT IS IMPORTED, i.e. after the reference has set it
    # (video_process_patch.py:10-11): build the stand-in lazily, at the reference's own import statement
    import importlib.abc
    import importlib.machinery

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            if name in ("qwen_vl_utils", "qwen_vl_utils.vision_process"):
                return importlib.machinery.ModuleSpec(name, self, is_package=(name == "qwen_vl_utils"))
            return None

        def create_module(self, spec):
            return None

        def exec_module(self, mod):
            if mod.__name__ == "qwen_vl_utils":
                mod.__path__ = []
                return
            vp = mod
            vp.FORCE_QWENVL_VIDEO_READER = os.environ.get("FORCE_QWENVL_VIDEO_READER")
            vp.IMAGE_FACTOR, vp.FRAME_FACTOR, vp.FPS = R.IMAGE_FACTOR, R.FRAME_FACTOR, R.FPS
            vp.VIDEO_MIN_PIXELS, vp.VIDEO_MAX_PIXELS = 128 * 28 * 28, 768 * 28 * 28      # MIN is overwritten by the reference (:12)
            vp.VIDEO_TOTAL_PIXELS = int(float(os.environ.get("VIDEO_MAX_PIXELS", 128000 * 28 * 28 * 0.9)))
            vp.FPS_MAX_FRAMES, vp.FPS_MIN_FRAMES = 768, V.FPS_MIN_FRAMES               # MAX is overwritten by the reference (:13)
            vp.VIDEO_READER_BACKENDS = {}

            def smart_nframes(ele, total_frames, video_fps):
                # the function body reads the module global the reference has just overwritten (FPS_MAX_FRAMES)
                saved = R.FPS_MAX_FRAMES
                R.FPS_MAX_FRAMES = vp.FPS_MAX_FRAMES
                try:
                    return V.smart_nframes(ele, total_frames, video_fps)
                finally:
                    R.FPS_MAX_FRAMES = saved
            vp.smart_nframes = smart_nframes
            vp.smart_resize = protocol.smart_resize
    sys.meta_path.insert(0, _Finder())


def main():
    global VIDEOS
    VIDEOS = make_videos()
    from transformers import logging as _hf_logging  # noqa: F401  (before the stubs: transformers probes torchvision by find_spec)
    real_exists = os.path.exists
    os.path.exists = lambda p: True if p in VIDEOS else real_exists(p)
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_video_process_patch", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    out = {"constants": dict(FPS_MAX_FRAMES=ref.FPS_MAX_FRAMES, VIDEO_MIN_PIXELS=ref.VIDEO_MIN_PIXELS, VIDEO_MAX_PIXELS=ref.VIDEO_MAX_PIXELS,
                             VIDEO_TOTAL_PIXELS=ref.VIDEO_TOTAL_PIXELS, FRAME_FACTOR=ref.FRAME_FACTOR, FPS=ref.FPS),
           "videos": "regenerated by make_videos() of this script (seeded)",
           "read": [], "spatial": [], "reader": [], "clip": []}

    eles = [dict(), dict(video_start=2.0, video_end=7.5), dict(video_start=None, video_end=3.0), dict(video_start=1.25),
            dict(nframes=16), dict(fps=1.0), dict(video_start=0.5, video_end=9.0, fps=4.0), dict(video_start=3.0, video_end=3.4)]
    for name in VIDEOS:
        for ele in eles:
            for strict, drop_last in ((False, True), (True, True), (True, False)):
                e = dict(ele, video=name, remote_loader=None)
                try:
                    clip, sample_fps, pts = ref._read_video_decord_plus(dict(e), strict_fps=strict, drop_last=drop_last, return_pts=True)
                    rec = dict(idxs=LOG["batch"], sample_fps=float(sample_fps), n=int(clip.shape[0]), pts=digest(pts))
                except Exception as ex:   # the reference raises for degenerate ranges (smart_nframes interval check ...)
                    rec = dict(error=type(ex).__name__)
                out["read"].append(dict(video=name, ele={k: v for k, v in ele.items()}, strict_fps=strict, drop_last=drop_last, **rec))

    for (h, w) in ((1080, 1920), (720, 1280), (360, 640), (480, 480), (2160, 3840), (240, 426)):
        for nframes in (2, 6, 8, 60, 120, 480):
            ref._spatial_resize_video(torch.zeros(2, 3, h, w, dtype=torch.uint8), nframes=nframes)
            out["spatial"].append(dict(h=h, w=w, nframes=nframes, size=LOG["resize"]))
        ref._spatial_resize_video(torch.zeros(4, 3, h, w, dtype=torch.uint8))
        out["spatial"].append(dict(h=h, w=w, nframes=None, T=4, size=LOG["resize"]))

    for name in VIDEOS:
        for mp in (None, 384 * 28 * 28, 100 * 28 * 28):
            _, rh, rw = ref.get_smart_resized_video_reader(name, mp)
            out["reader"].append(dict(video=name, max_pixels=mp, size=[int(rh), int(rw)]))

    cases = [("a_30fps_10s", [0.0, 0.5, 1.0, 1.5, 2.0, 2.5], 0), ("a_30fps_10s", [3.0, 3.5], 91), ("a_30fps_10s", [3.0], 91),
             ("a_30fps_10s", [9.5], 280), ("a_30fps_10s", [9.5, 10.0, 10.5], 280), ("b_25fps_jitter", [0.0, 0.5, 1.0], 0),
             ("e_short", [0.0, 0.5, 1.0, 1.5, 2.0, 2.5], 0), ("f_2fps", [58.0, 58.5, 59.0, 59.5, 60.0], 110), ("f_2fps", [60.0], 120),
             ("c_60fps_20s", [0.0, 0.5], 0)]
    for name, ts, frm in cases:
        vr = FakeVideoReader(name)
        _, ts_out, idxs = ref.get_smart_resized_clip(vr, 28, 28, torch.tensor(ts, dtype=torch.float64), VIDEOS[name]["pts"], frm)
        out["clip"].append(dict(video=name, timestamps=ts, index_from=frm, idxs=[int(i) for i in idxs],
                                timestamps_out=[float(x) for x in ts_out]))

    path = os.path.join(ROOT, "tests", "golden", "frame_selection.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"wrote {path}: {len(out['read'])} read cases, {len(out['spatial'])} resize budgets, {len(out['reader'])} reader sizes, "
          f"{len(out['clip'])} clip scans")


if __name__ == "__main__":
    main()
