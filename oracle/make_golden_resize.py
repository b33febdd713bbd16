"""TEST INFRASTRUCTURE: writes tests/golden/resize_small.npz -- seeded uint8 frames and their antialias-bicubic resize by
torch's own CPU kernel (oracle/resize_ref.py = the torchvision call of ref livecc_utils/video_process_patch.py:150-155).
Run from the repo root:  python -m oracle.make_golden_resize"""
import os

import numpy as np
import torch

from oracle import resize_ref as R


def main():
    g = torch.Generator().manual_seed(2024)
    frames = torch.randint(0, 256, (2, 3, 54, 96), dtype=torch.uint8, generator=g)
    frames[0, :, :10] = 255          # saturated regions exercise the clamp of the bicubic overshoot
    frames[0, :, 10:20] = 0
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resize_small.npz")
    np.savez_compressed(path, frames=frames.numpy(), out_hw=np.asarray([28, 56]), resized=R.resize_ref(frames, 28, 56).numpy())
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
