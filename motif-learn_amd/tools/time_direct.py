import os, sys, warnings
sys.path.insert(0, "/root/repo/motif-learn_amd")
import numpy as np, torch
from mtflearn_amd import ZPs, _native, distributed as D
from mtflearn_amd.synthetic import honeycomb_frame
torch.cuda.set_device(0)
frame = torch.from_numpy(honeycomb_frame(2048, seed=0)).cuda()
for n_max, K in ((18, 40), (20, 40), (24, 48), (28, 56), (36, 72)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore"); z = ZPs(n_max, K)
    plan = z._device_plan(); plan.set_path(_native.PATH_DIRECT)
    n_t = 1 << 18 if K <= 48 else 1 << 17
    pt = frame.unfold(0, K, 5).unfold(1, K, 5).reshape(-1, K, K)[:n_t].contiguous()
    out = D.patch_moments_device(plan, pt); torch.cuda.synchronize()
    for _ in range(10): D.patch_moments_device(plan, pt, out=out)
    npx = int(np.count_nonzero(z.polynomials[0])); fl = 2.0 * npx * len(z.n)
    for label, env in (("rolling re-arm", None), ("whole-slab re-arm (round 3)", "1"), ("rolling re-arm", None)):
        os.environ.pop("ZK_DIRECT_NO_ROLL", None)
        if env: os.environ["ZK_DIRECT_NO_ROLL"] = env
        for _ in range(10): D.patch_moments_device(plan, pt, out=out)
        plan.profile(True)
        for _ in range(10): D.patch_moments_device(plan, pt, out=out)
        torch.cuda.synchronize(); _, ms = plan.profile_read(); plan.profile(False)
        print(f"batch direct ({K}, {n_max}) {label:28s}: {n_t / (ms / 10) / 1e3:7.1f} M patches/s  {n_t * fl / (ms / 10) / 1e9:6.1f} TFLOP/s = {n_t * fl / (ms / 10) / 1e9 / 78.6:.2f} of peak", flush=True)
    os.environ.pop("ZK_DIRECT_NO_ROLL", None)
