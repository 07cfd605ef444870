#!/usr/bin/env python3
"""Soak: seeded random (size, n_max, dtype, shapes); every kernel family a plan offers against the generic kernel.

  python motif-learn_amd/tools/soak_random_plans.py [seed] [iterations] [n_max below this bound, default 25; 41 reaches the
  matrix-core plain sums of n_max 25-40]
"""
import sys, numpy as np, warnings, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "motif-learn_amd")); sys.path.insert(0, R)
from mtflearn_amd import ZPs, _native as native
def zps(n,k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore"); return ZPs(n,k)
def both(z, array, mode):
    plan=z._device_plan(); run=plan.transform_patches if mode==0 else plan.transform_frame; out={}
    for path,name in native.PATH_NAMES.items():
        if plan.has_path(mode, native.dtype_code(array.dtype), path):
            plan.set_path(path); out[name]=run(array)
    plan.set_path(native.PATH_AUTO); return out
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 123)
bad=0; n_cmp=0
NMAX_BOUND=int(sys.argv[3]) if len(sys.argv)>3 else 25
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 300):
    size=int(rng.integers(8,97)); n_max=int(min(size, rng.integers(0,NMAX_BOUND)))
    dtype=np.float32 if rng.random()<0.6 else np.float64
    z=zps(n_max,size)
    p=(rng.random((int(rng.integers(1,400)),size,size))-0.4).astype(dtype)
    img=(rng.random((int(rng.integers(size,size+60)),int(rng.integers(size,size+200))))-0.4).astype(dtype)
    floor=3e-7 if n_max>20 else 1e-8 if n_max>16 else 1e-10 if n_max>12 else 1e-11 if n_max>10 else 1e-12
    for mode,arr in ((0,p),(1,img)):
        out=both(z,arr,mode); ref=out["generic"]
        for name,got in out.items():
            if name!="generic":
                n_cmp+=1
                err=np.abs(got-ref).max()/np.abs(ref).max()
                if not err<=floor:
                    bad+=1; print("MISMATCH",size,n_max,dtype.__name__,mode,name,err,flush=True)
    if it%10==9: print("iter",it+1,"comparisons",n_cmp,"bad",bad,flush=True)
print("done comparisons",n_cmp,"bad",bad)
