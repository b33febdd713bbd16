#!/usr/bin/env python3
"""Compile every HIP source to gfx950 assembly and list the kernels that move MFMA accumulators between AGPRs and VGPRs (v_accvgpr_read /
v_accvgpr_write), with the basic blocks that hold both an MFMA and such copies -- i.e. copies INSIDE a matrix loop.  hipcc picks the AGPR form
of the MFMAs whenever a kernel may have one wave per SIMD (__launch_bounds__ <= 256 threads without a minimum of 2 waves per SIMD): harmless
for a GEMM (the accumulators are zeroed once and read once), expensive for attention, whose softmax reads the scores and rescales O in the
vector registers every key tile.  No GPU needed.

    python tools/audit_agpr_copies.py [file.hip ...] [--all]        (default: every source; only kernels with copies inside an MFMA block)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "livecc_amd", "csrc")


def asm_of(src, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}", "-S",
                    "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
    return open(out).read()


def audit(text):
    rows = []
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        body = m.group(2)
        rd, wr = body.count("v_accvgpr_read"), body.count("v_accvgpr_write")
        if rd + wr == 0:
            continue
        blocks, cur = collections.OrderedDict(entry=[]), "entry"
        for line in body.split("\n"):
            line = line.strip()
            if not line or line.startswith(";"):
                continue
            if re.match(r"^\.LBB\d+_\d+:", line):
                cur = line.split(":")[0]
                blocks[cur] = []
            else:
                blocks[cur].append(line)
        hot = []
        for k, v in blocks.items():
            nm = sum("v_mfma" in x for x in v)
            nc = sum(x.startswith("v_accvgpr") for x in v)
            if nm and nc:
                hot.append(dict(block=k, instructions=len(v), mfma=nm, agpr_copies=nc))
        rows.append(dict(symbol=m.group(1), mfma=body.count("v_mfma"), accvgpr_read=rd, accvgpr_write=wr, mfma_blocks_with_copies=hot))
    return rows


def main():
    show_all = "--all" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    srcs = [os.path.join(CSRC, a) if not os.path.isabs(a) and not os.path.exists(a) else a for a in args] or \
        sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as td, ThreadPoolExecutor(max_workers=4) as pool:
        texts = list(pool.map(lambda src: asm_of(src, os.path.join(td, os.path.basename(src) + ".s")), srcs))
        for src, text in zip(srcs, texts):
            rows = audit(text)
            names = subprocess.run(["c++filt"], input="\n".join(r["symbol"] for r in rows), capture_output=True, text=True).stdout.splitlines()
            for r, n in zip(rows, names):
                n = re.sub(r"\(.*", "", n)
                # copies in the entry block (accumulators zeroed before the loop) are expected; a GEMM's drain block holds no MFMA
                hot = [h for h in r["mfma_blocks_with_copies"] if h["block"] != "entry"]
                if hot:
                    bad += 1
                if hot or show_all:
                    print(f"{os.path.basename(src):16s} {n[:70]:70s} mfma={r['mfma']:4d} read={r['accvgpr_read']:4d} write={r['accvgpr_write']:4d}"
                          + ("".join(f"\n    {h['block']}: {h['instructions']} instructions, {h['mfma']} MFMAs, {h['agpr_copies']} AGPR copies" for h in hot)))
    print(f"{bad} kernel(s) with AGPR copies inside an MFMA block")


if __name__ == "__main__":
    main()
