"""In-tree build of the HIP extension: hipcc --offload-arch=gfx950 -> livecc_amd/_C/liblivecc_amd.so.
hipcc cross-compiles without a GPU, so this runs in the build container as well as on the MI355X box."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = True, force: bool = False) -> str:
    csrc = os.path.join(HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building liblivecc_amd.so failed")
    if verbose and r.stdout.strip():
        print(r.stdout.strip().splitlines()[-1])
    so = os.path.join(HERE, "_C", "liblivecc_amd.so")
    assert os.path.exists(so), so
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
