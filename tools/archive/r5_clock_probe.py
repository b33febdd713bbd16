#!/usr/bin/env python
"""What clock and power does the chip give the 8-wave GEMM?  Back-to-back gate/up launches (M = 3088) for ~5 s per operand kind while a
thread polls the amdgpu hwmon / pp_dpm files (falling back to `rocm-smi --json`): random operands vs zero-filled ones, and the
HBM-bound decode GEMV for comparison.  One JSON line per phase: launches/s, median shader clock, median socket power."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livecc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def read_sysfs():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key, scale in (("freq1_input", "sclk_mhz", 1e-6), ("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6)):
            f = os.path.join(hw, name)
            if os.path.exists(f):
                try:
                    out.setdefault(key, float(open(f).read().strip()) * scale)
                except (OSError, ValueError):
                    pass
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if line.strip().endswith("*"):
                    out.setdefault("dpm_sclk_mhz", float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")))
        except (OSError, ValueError, IndexError):
            pass
    return out


def read_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
        j = json.loads(r.stdout)
        card = next(iter(j.values()))
        out = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "clock" in kl:
                out["smi_sclk"] = v
            if "power" in kl and ("socket" in kl or "average" in kl or "package" in kl):
                out["smi_power"] = v
        return out
    except Exception as e:      # noqa: BLE001
        return dict(smi_error=str(e)[:80])


class Poller(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        use_smi = not read_sysfs()
        while not self.stop:
            self.samples.append(read_smi() if use_smi else read_sysfs())
            time.sleep(0.05 if not use_smi else 0.2)


def med(xs):
    xs = sorted(x for x in xs if isinstance(x, (int, float)))
    return round(xs[len(xs) // 2], 1) if xs else None


def phase(name, fn, seconds=5.0):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    p = Poller()
    p.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        n += 40
    dt = time.perf_counter() - t0
    p.stop = True
    p.join()
    keys = sorted({k for s in p.samples for k in s})
    rec = dict(phase=name, us_per_launch=round(dt / n * 1e6, 1), samples=len(p.samples))
    for k in keys:
        vals = [s.get(k) for s in p.samples[len(p.samples) // 4:]]        # skip the ramp
        rec[k] = med(vals) if all(isinstance(v, (int, float)) or v is None for v in vals) else vals[-1]
    print(json.dumps(rec), flush=True)


H, I = 3584, 18944
N, K, M = 2 * I, H, 3088
w_rand = [ops.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(2)]
w_zero = [torch.zeros_like(w_rand[0]) for _ in range(2)]
x_rand = torch.randn(M, K, device=dev).to(torch.bfloat16)
x_zero = torch.zeros_like(x_rand)
x1 = torch.randn(1, K, device=dev).to(torch.bfloat16)
cnt = [0]


def gemm(x, ws):
    cnt[0] += 1
    ops.linear(x, ws[cnt[0] & 1], None, ops.EPI_SWIGLU, packed_shape=(N, K))


print(json.dumps(dict(idle=read_sysfs() or read_smi())), flush=True)
for rep in range(2):
    phase("gemm_gate_up_M3088_random", lambda: gemm(x_rand, w_rand))
    phase("gemm_gate_up_M3088_zeros", lambda: gemm(x_zero, w_zero))
    phase("gemv_gate_up_M1_random (HBM-bound)", lambda: gemm(x1, w_rand), seconds=3.0)
