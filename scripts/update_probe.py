"""Time the ARS update kernel in isolation (design aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
dev = "cuda:0"
p = sw.SwParams.make(3)
N = 512
f64 = dict(dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
ret = torch.as_tensor(rng.standard_normal(2 * N), **f64)
deltas = torch.as_tensor(rng.uniform(-1, 1, (N, 2, 8)), **f64)
pol = torch.zeros((2, 8), **f64)
rows = sw.kernels.moments_blocks(2 * N)
mom = torch.as_tensor(rng.standard_normal((rows, 16)) ** 2, **f64)
running = torch.zeros(17, **f64); mean = torch.zeros(8, **f64); inv = torch.ones(8, **f64)
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("V1 (no stats block)      : %.2f us" % t(lambda: sw.kernels.ars_update(p, ret, deltas, pol, 0.01, N)))
print("V2 rows=%d               : %.2f us" % (rows, t(lambda: sw.kernels.ars_update(p, ret, deltas, pol, 0.01, N, moments=mom, running=running, n_new_states=1000, mean=mean, inv_std=inv))))
mom4 = mom[:4].contiguous()
print("V2 rows=4                : %.2f us" % t(lambda: sw.kernels.ars_update(p, ret, deltas, pol, 0.01, N, moments=mom4, running=running, n_new_states=1000, mean=mean, inv_std=inv)))
print("top_b=64                 : %.2f us" % t(lambda: sw.kernels.ars_update(p, ret, deltas, pol, 0.01, N, top_b=64)))
