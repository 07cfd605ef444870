#!/usr/bin/env python3
"""Where the time of zk_frame_strip2_kernel goes that no wave accounts for: every workgroup of the
instrumented build (make -C csrc trace -> lib/libzernike_hip_trace.so) records the 100-MHz constant clock (s_memrealtime) and
the shader clock (s_memtime) at its start, after staging, after arithmetic + store issue and after its stores have drained,
plus HW_ID / XCC_ID.  From that: the shader clock DURING the kernel, the occupancy of the wave slots over the kernel's
duration, the phases of a workgroup's life, the start-up ramp and the drain tail.

    strip_trace.py [--n-max 8] [--size 32] [--frame 2048] [--reps 3]
"""
import argparse, ctypes, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
os.environ["MTFLEARN_AMD_LIB"] = os.path.join(ROOT, "motif-learn_amd", "mtflearn_amd", "lib",
                                              os.environ.get("ZK_TRACE_LIB", "libzernike_hip_trace.so"))
import numpy as np
import torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-max", type=int, default=8)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--frame", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--idle-ms", type=float, default=0.0, help="sleep before the traced launch (clock ramp from idle)")
    ap.add_argument("--series", default="", help="comma list of back-to-back launch counts: clock and time of the LAST launch of each run")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    lib = _native.load()
    lib.zk_debug_strip_trace.argtypes = [ctypes.c_void_p]
    lib.zk_debug_strip_trace.restype = ctypes.c_int
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z = ZPs(a.n_max, a.size)
    plan = z._device_plan()
    H = a.frame
    f = torch.from_numpy(honeycomb_frame(H, seed=0)).cuda()
    gx, gy = (H + 63) // 64, (H + 7) // 8
    buf = torch.zeros(gx * gy * 4 * 10, dtype=torch.int64, device="cuda")
    out = D.frame_moments_device(plan, f)
    torch.cuda.synchronize()
    if a.series:
        import time
        assert lib.zk_debug_strip_trace(ctypes.c_void_p(buf.data_ptr())) == 0
        print(f"zk_frame_strip2_kernel<{a.n_max}> K {a.size} frame {H}^2: shader clock and duration of the last of k back-to-back launches "
              f"(each run starts after {a.idle_ms:.0f} ms of idle)")
        for k in [int(x) for x in a.series.split(",")]:
            torch.cuda.synchronize()
            time.sleep(a.idle_ms / 1e3)
            for _ in range(k):
                D.frame_moments_device(plan, f, out=out)
            torch.cuda.synchronize()
            t = buf.cpu().numpy().reshape(-1, 10).astype(np.int64)
            t = t[t[:, 0] > 0]                                  # (kernels with taller workgroups leave the rest of the buffer untouched)
            buf.zero_()
            rt, ck = t[:, 0:4], t[:, 4:8]
            ghz = (ck[:, 3] - ck[:, 0]) / ((rt[:, 3] - rt[:, 0]) * 10.0)
            print(f"  k = {k:4d}: {np.median(ghz):.3f} GHz   {(rt[:, 3].max() - rt[:, 0].min()) / 100.0:8.1f} us   wave life {np.mean(ck[:, 3] - ck[:, 0]):.0f} clocks   ({len(t)} waves)", flush=True)
        lib.zk_debug_strip_trace(ctypes.c_void_p(0))
        return
    for _ in range(a.reps):                       # warm: clocks up, the traced launch is the last of a back-to-back run
        D.frame_moments_device(plan, f, out=out)
    if a.idle_ms > 0:
        torch.cuda.synchronize()
        import time
        time.sleep(a.idle_ms / 1e3)
    assert lib.zk_debug_strip_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    plan.profile(True)
    D.frame_moments_device(plan, f, out=out)
    torch.cuda.synchronize()
    _, ms = plan.profile_read()
    plan.profile(False)
    lib.zk_debug_strip_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 10).astype(np.int64)
    t = t[t[:, 0] > 0]
    rt, ck, hw, xcc = t[:, 0:4], t[:, 4:8], t[:, 8], t[:, 9] & 0xf
    t0, t1 = rt[:, 0].min(), rt[:, 3].max()
    dur_us = (t1 - t0) / 100.0
    print(f"zk_frame_strip2_kernel<{a.n_max}> K {a.size} frame {H}^2: {gx * gy} workgroups x 4 waves; HIP-event time {ms * 1e3:.1f} us, "
          f"first wave start -> last wave drained {dur_us:.1f} us (100-MHz clock)")
    life_rt = (rt[:, 3] - rt[:, 0]).astype(float)          # 10-ns ticks
    life_ck = (ck[:, 3] - ck[:, 0]).astype(float)
    ghz = life_ck / (life_rt * 10.0)
    print(f"shader clock during the kernel (s_memtime / s_memrealtime per wave life): median {np.median(ghz):.3f} GHz, "
          f"5 % {np.percentile(ghz, 5):.3f}, 95 % {np.percentile(ghz, 95):.3f}")
    # by time of the kernel: clock of the waves that start in each tenth
    tenth = np.minimum(((rt[:, 0] - t0) * 10 // max(1, t1 - t0)).astype(int), 9)
    print("  by tenth of the kernel's duration:", " ".join(f"{np.median(ghz[tenth == k]):.2f}" if np.any(tenth == k) else "-" for k in range(10)))
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    n_cu = len(np.unique(cu_key))
    print(f"compute units seen: {n_cu}; XCCs {len(np.unique(xcc))}; waves per CU min / mean / max: "
          f"{np.bincount(np.unique(cu_key, return_inverse=True)[1]).min()} / {len(cu_key) / n_cu:.1f} / {np.bincount(np.unique(cu_key, return_inverse=True)[1]).max()}")
    slots = n_cu * 4 * 2
    occ = life_rt.sum() / (slots * (t1 - t0))
    print(f"wave-slot occupancy over the kernel's duration (sum of wave lives / ({n_cu} CUs x 4 SIMDs x 2 slots x duration)): {occ:.3f}")
    ph = np.stack([rt[:, 1] - rt[:, 0], rt[:, 2] - rt[:, 1], rt[:, 3] - rt[:, 2]], 1) / 100.0
    print("a wave's life (us): staging + barrier {:.2f}  arithmetic + store issue {:.2f}  store drain {:.2f}  total {:.2f}".format(
        *ph.mean(0), ph.sum(1).mean()))
    phc = np.stack([ck[:, 1] - ck[:, 0], ck[:, 2] - ck[:, 1], ck[:, 3] - ck[:, 2]], 1)
    print("a wave's life (shader clocks): {:.0f} + {:.0f} + {:.0f} = {:.0f}".format(*phc.mean(0), phc.sum(1).mean()))
    # residency over time: waves resident on the chip in each 1-us bin
    nb = int((t1 - t0) // 100) + 1
    res = np.zeros(nb + 1)
    np.add.at(res, ((rt[:, 0] - t0) // 100).astype(int), 1)
    np.add.at(res, np.minimum(((rt[:, 3] - t0) // 100).astype(int) + 1, nb), -1)
    res = np.cumsum(res)[:nb]
    comp = np.zeros(nb + 1)
    np.add.at(comp, ((rt[:, 1] - t0) // 100).astype(int), 1)
    np.add.at(comp, np.minimum(((rt[:, 2] - t0) // 100).astype(int) + 1, nb), -1)
    comp = np.cumsum(comp)[:nb]
    print(f"resident waves per microsecond bin (of {slots} slots): mean {res.mean() / slots:.3f} of the slots; in their arithmetic phase: {comp.mean() / slots:.3f}")
    edges = np.linspace(0, nb, 11).astype(int)
    print("  resident by tenth:  ", " ".join(f"{res[edges[k]:edges[k + 1]].mean() / slots:.2f}" for k in range(10)))
    print("  arithmetic by tenth:", " ".join(f"{comp[edges[k]:edges[k + 1]].mean() / slots:.2f}" for k in range(10)))
    # start-up ramp and tail per CU
    first = np.array([rt[cu_key == c, 0].min() for c in np.unique(cu_key)])
    last = np.array([rt[cu_key == c, 3].max() for c in np.unique(cu_key)])
    print(f"first wave of a CU after the kernel's start: median {np.median(first - t0) / 100:.2f} us, max {np.max(first - t0) / 100:.2f}; "
          f"last wave of a CU before the kernel's end: median {np.median(t1 - last) / 100:.2f} us, max {np.max(t1 - last) / 100:.2f}")
    # gaps: per (CU, SIMD), idle time between a wave's end and the start of the next wave in the same slot stream
    gaps = []
    for c in np.unique(cu_key):
        m = cu_key == c
        for sd in range(4):
            mm = m & (simd == sd)
            if mm.sum() < 2:
                continue
            ev = sorted([(x, 1) for x in rt[mm, 0]] + [(x, -1) for x in rt[mm, 3]])
            level, last_t, acc = 0, t0, {0: 0, 1: 0, 2: 0}
            for x, d in ev:
                acc[min(level, 2)] += x - last_t
                level += d
                last_t = x
            acc[0] += t1 - last_t
            gaps.append([acc[0], acc[1], acc[2]])
    g = np.array(gaps, float) / (t1 - t0)
    print(f"a SIMD over the kernel's duration: no wave {g[:, 0].mean():.3f}, one wave {g[:, 1].mean():.3f}, two waves {g[:, 2].mean():.3f}")


if __name__ == "__main__":
    main()
