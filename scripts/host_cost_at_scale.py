import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
from swimmer_amd._lib import numpy_global_uniform_pm1
for N, md in ((512, 16), (4096, 16), (2048, 70)):
    a = torch.empty((N, md), dtype=torch.float64).pin_memory()
    an = a.numpy()
    d = torch.empty((N, md), dtype=torch.float64, device="cuda:0")
    np.random.seed(0)
    numpy_global_uniform_pm1(an)
    t = time.perf_counter()
    for _ in range(200): numpy_global_uniform_pm1(an)
    rng = (time.perf_counter() - t) / 200
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        with torch.cuda.stream(s):
            d.copy_(a, non_blocking=True)
        s.synchronize()
    h2d = (time.perf_counter() - t) / 200
    print(f"N={N} md={md}: native RNG {rng*1e6:.1f} us, H2D + wait {h2d*1e6:.1f} us", flush=True)
