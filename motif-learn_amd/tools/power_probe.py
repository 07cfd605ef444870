#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi, read-only) while one kernel family runs back to back for a few seconds: the dense strip kernel
(FP64 vector pipe fed from scalar registers), the matrix-core plain sum, the HBM-bound batch kernel.  Context for the clock figures in
profiles/r04_strip_clock_series.txt: which kernels the chip's power management slows, and at what draw.   power_probe.py [seconds]"""
import os, subprocess, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as exc:  # noqa: BLE001
        return f"rocm-smi failed: {exc}"
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("Power", "sclk", "Temperature (Sensor junction)", "Temperature (Sensor edge)"))]
    return " | ".join(keep)


def zps(n, k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ZPs(n, k)


torch.cuda.set_device(0)
f = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
print("idle:", smi(), flush=True)
cases = []
z8 = zps(8, 32); p8 = z8._device_plan(); o8 = D.frame_moments_device(p8, f)
cases.append(("dense strip kernel (32, 8), FP64 vector pipe", lambda: D.frame_moments_device(p8, f, out=o8), p8))
z20 = zps(20, 40); p20 = z20._device_plan(); o20 = D.frame_moments_device(p20, f, row0=256, n_rows=512)
cases.append(("dense plain sum (40, 20), FP64 matrix cores", lambda: D.frame_moments_device(p20, f, row0=256, n_rows=512, out=o20), p20))
pt = f.unfold(0, 32, 1).unfold(1, 32, 1).reshape(-1, 32, 32)[:1 << 21].contiguous(); ob = D.patch_moments_device(p8, pt)
cases.append(("batch kernel (32, 8), HBM-bound", lambda: D.patch_moments_device(p8, pt, out=ob), p8))
for label, fn, plan in cases:
    stop = threading.Event(); samples = []

    def sampler():
        time.sleep(1.0)
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.7)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < SECS - 1.5:
        for _ in range(20):
            fn()
        torch.cuda.current_stream().synchronize(); n += 20
    with _native.ClockMonitor(0, 1900.0) as mon:          # (the monitor's wave lives at most 2 s: the last 1.5 s of the run)
        t1 = time.time()
        while time.time() - t1 < 1.5:
            for _ in range(20):
                fn()
            torch.cuda.current_stream().synchronize(); n += 20
    stop.set(); th.join()
    print(f"== {label}: {n} launches in {time.time() - t0:.1f} s, shader clock by the resident-wave monitor {mon.ghz:.3f} GHz")
    for s in samples[:4]:
        print("   ", s)
    sys.stdout.flush()
    time.sleep(2.0)
