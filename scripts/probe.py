"""GPU probe: raw kernel timings that steer the design (not part of the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import swimmer_amd as sw

dev = "cuda:0"
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.median(ts))

print(torch.cuda.get_device_name(0))
for n in (3, 6):
    p = sw.SwParams.make(n)
    d, m = 2*n+2, n-1
    H = 1000
    rng = np.random.default_rng(0)
    for R in (16, 64, 1024, 4096, 16384, 65536, 262144):
        pol = torch.as_tensor(0.05*rng.uniform(-1, 1, (R, m, d)), device=dev)
        mean = torch.zeros(d, dtype=torch.float64, device=dev); inv = torch.ones(d, dtype=torch.float64, device=dev)
        ret = torch.empty(R, dtype=torch.float64, device=dev)
        mom = torch.zeros((sw.kernels.moments_blocks(R), 2*d), dtype=torch.float64, device=dev)
        t_plain = timeit(lambda: sw.kernels.rollout(p, H, pol, mean=mean, inv_std=inv, returns=ret))
        t_mom = timeit(lambda: sw.kernels.rollout(p, H, pol, mean=mean, inv_std=inv, returns=ret, moments=mom))
        line = f"n={n} rollout R={R:7d} H={H}: plain {t_plain[0]:8.3f} ms ({R*H/t_plain[0]/1e6:9.1f} Msteps/s)  +moments {t_mom[0]:8.3f} ms"
        if R*H*d*8 < 20e9:
            traj = torch.empty((H, d, R), dtype=torch.float64, device=dev)
            t_tr = timeit(lambda: sw.kernels.rollout(p, H, pol, mean=mean, inv_std=inv, returns=ret, moments=mom, traj=traj))
            line += f"  +traj {t_tr[0]:8.3f} ms ({R*H/t_tr[0]/1e6:9.1f} Msteps/s)"
            t_cov = timeit(lambda: sw.kernels.traj_moments(p, traj))
            line += f"  cov-pass {t_cov[0]:7.3f} ms ({R*H*d*8/t_cov[0]/1e6:8.1f} GB/s)"
            del traj
        print(line, flush=True)
    for B in (8192, 1 << 16, 1 << 20, 1 << 22, 1 << 24):
        st = torch.as_tensor(rng.uniform(-1, 1, (d, B)), device=dev)
        ac = torch.as_tensor(rng.uniform(-1, 1, (m, B)), device=dev)
        out = torch.empty_like(st); rew = torch.empty(B, dtype=torch.float64, device=dev)
        def f():
            for _ in range(20): sw.kernels.step(p, st, ac, out=out, reward=rew)
        t = timeit(f)
        per = t[0]/20
        bytes_ = (2*d + m + 1)*8*B
        print(f"n={n} step B={B:9d}: {per*1e3:9.2f} us/launch  {B/per/1e3:10.1f} Msteps/s  {bytes_/per/1e6:8.1f} GB/s algorithmic", flush=True)
