#!/usr/bin/env python3
"""Soak: large random batches (98 k - 220 k patches) through ZK_PATH_AUTO (stream / row-pair kernels) against the
generic kernel, on device buffers."""
import sys, os, warnings, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "motif-learn_amd")); sys.path.insert(0, R)
from mtflearn_amd import ZPs, _native as native
rng = np.random.default_rng(5); bad = 0
gen = torch.Generator(device="cuda").manual_seed(1)
for it in range(24):
    size = int(rng.integers(8, 81)); n_max = int(min(size, rng.integers(2, 17)))
    f64 = rng.random() < 0.35
    n = int(rng.integers(98304, 220000))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore"); z = ZPs(n_max, size)
    plan = z._device_plan(); code = native.ZK_F64 if f64 else native.ZK_F32
    src = torch.rand((n, size, size), dtype=torch.float64 if f64 else torch.float32, device="cuda", generator=gen) - 0.3
    outs = {}
    for name, path in (("auto", 0), ("generic", 1)):
        plan.set_path(path)
        o = torch.empty((n, len(z.n)), dtype=torch.float64, device="cuda")
        plan.transform_patches_dev(src.data_ptr(), code, n, o.data_ptr(), 0); torch.cuda.synchronize(); outs[name] = o
    plan.set_path(0)
    floor = 1e-10 if n_max > 12 else 1e-11 if n_max > 10 else 1e-12
    err = float((outs["auto"] - outs["generic"]).abs().max() / outs["generic"].abs().max())
    stream = plan.has_path(0, code, native.PATH_STREAM)
    ok = err <= floor
    bad += not ok
    print(f"K={size:3d} n_max={n_max:2d} {'f64' if f64 else 'f32'} N={n:7d} stream_available={stream} rel err {err:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    del src, outs; torch.cuda.empty_cache()
print("bad", bad)
