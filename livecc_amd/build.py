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
    build_torch_ops(verbose=verbose, force=force)
    return so


def build_torch_ops(verbose: bool = True, force: bool = False) -> str:
    """csrc/torch_ops.cpp -> _C/liblivecc_torch_ops.so: the TORCH_LIBRARY registration of the operator-level entry points (host code only:
    g++ against the installed PyTorch-ROCm's headers, linked to liblivecc_amd.so through rpath $ORIGIN)."""
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(HERE, "csrc", "torch_ops.cpp")
    out = os.path.join(HERE, "_C", "liblivecc_torch_ops.so")
    deps = [src, os.path.join(os.path.dirname(HERE), "include", "livecc_amd.h"), os.path.join(HERE, "_C", "liblivecc_amd.so")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps[:2]):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include", src, "-o", out, f"-L{os.path.join(HERE, '_C')}", "-llivecc_amd",
                                                       f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
                                                       "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building liblivecc_torch_ops.so failed")
    if verbose:
        print(out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
