#!/usr/bin/env python3
"""Does a resident monitor wave see the clock the loaded kernel's waves see?  Ground truth: the instrumented strip kernel
(lib/libzernike_hip_trace.so, per-wave s_memtime / s_memrealtime); monitor: zk_clock_monitor in its three forms
(ZK_CLOCK_MONITOR_MODE 0 naps, 1 spins, 2 does FP64 work between its samples)."""
import ctypes, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
os.environ["MTFLEARN_AMD_LIB"] = os.path.join(ROOT, "motif-learn_amd", "mtflearn_amd", "lib", "libzernike_hip_trace.so")
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame

torch.cuda.set_device(0)
lib = _native.load()
lib.zk_debug_strip_trace.argtypes = [ctypes.c_void_p]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    z = ZPs(8, 32)
plan = z._device_plan()
H = 2048
f = torch.from_numpy(honeycomb_frame(H, seed=0)).cuda()
buf = torch.zeros((H // 64) * (H // 8) * 4 * 10, dtype=torch.int64, device="cuda")
out = D.frame_moments_device(plan, f)
torch.cuda.synchronize()
lib.zk_debug_strip_trace(ctypes.c_void_p(buf.data_ptr()))
for mode in (0, 1, 2):
    os.environ["ZK_CLOCK_MONITOR_MODE"] = str(mode)
    for _ in range(40):
        D.frame_moments_device(plan, f, out=out)
    with _native.ClockMonitor(0) as m:
        for _ in range(20):
            D.frame_moments_device(plan, f, out=out)
        torch.cuda.current_stream().synchronize()      # (a DEVICE synchronisation would wait for the monitor itself)
    t = buf.cpu().numpy().reshape(-1, 10).astype(np.int64)
    ghz = (t[:, 7] - t[:, 4]) / ((t[:, 3] - t[:, 0]) * 10.0)
    print(f"monitor mode {mode}: monitor {m.ghz:.3f} GHz over {m.ms:.2f} ms   |   the kernel's own waves (last launch): median {np.median(ghz):.3f} GHz")
