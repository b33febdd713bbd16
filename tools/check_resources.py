#!/usr/bin/env python3
"""Compile the HIP sources with -Rpass-analysis=kernel-resource-usage and print VGPR / scratch / occupancy per kernel.
Exits non-zero when any kernel of this library spills to scratch (a silent 3-5x slowdown: it happened once to the 8-wave
GEMM when a per-16-row predicate made hipcc if-convert the accumulators).  No GPU needed.

    python tools/check_resources.py [file.hip ...] [--filter substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "livecc_amd", "csrc")


def analyse(src):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}",
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = dict(name=m.group(1))
            out.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and key not in cur:
                cur[key] = int(m.group(1))
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = p.stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    if flt in args:
        args.remove(flt)
    srcs = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for src in srcs:
        rows = analyse(src)
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*", "", n)
            if flt and flt not in n:
                continue
            spill = r.get("scratch", 0) > 0
            bad += spill
            print(f"{os.path.basename(src):16s} {n[:70]:70s} vgpr={r.get('vgpr', -1):3d} scratch={r.get('scratch', 0):4d} occ={r.get('occ', -1)}"
                  + ("   <-- SPILLS" if spill else ""))
    if bad:
        print(f"{bad} kernel(s) spill to scratch", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
