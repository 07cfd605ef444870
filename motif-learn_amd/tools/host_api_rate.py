import sys, time, os
sys.path.insert(0, "motif-learn_amd")
import numpy as np
from mtflearn_amd import ZPs
z = ZPs(8, 32)
rng = np.random.default_rng(0)
p = rng.random((500000, 32, 32), dtype=np.float32)
z.transform(p[:1000])
for _ in range(3):
    t = time.perf_counter(); zm = z.transform(p); dt = time.perf_counter() - t
    print(f"host->moments {p.shape[0]} patches: {dt*1e3:.1f} ms  {p.shape[0]/dt/1e6:.1f} M patches/s  {p.nbytes/dt/1e9:.1f} GB/s in")
img = rng.random((4096, 4096), dtype=np.float32)
z.transform(img[:64, :64])
for _ in range(2):
    t = time.perf_counter(); zm = z.transform(img); dt = time.perf_counter() - t
    print(f"host frame 4096^2 -> (45,4096,4096): {dt*1e3:.1f} ms  out {zm.data.nbytes/dt/1e9:.1f} GB/s")
