import sys, time, warnings
sys.path.insert(0, "motif-learn_amd")
import numpy as np
from mtflearn_amd import ZPs
warnings.simplefilter("ignore")
for n_max, size in ((8, 32), (12, 64), (10, 32), (24, 48), (36, 72)):
    t0 = time.perf_counter(); z = ZPs(n_max, size); t1 = time.perf_counter(); p = z._device_plan(); t2 = time.perf_counter()
    z2 = ZPs(n_max, size); t3 = time.perf_counter(); p2 = z2._device_plan(); t4 = time.perf_counter()
    print(f"({size}, {n_max}): basis {1e3*(t1-t0):.1f} ms, plan {1e3*(t2-t1):.1f} ms; second object: basis {1e3*(t3-t2):.1f} ms, plan {1e3*(t4-t3):.1f} ms")
