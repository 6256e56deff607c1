"""What the covariance pass costs at the size where a batch fills every SIMD (design aid):
(1) the standalone pass (sw_traj_moments_f64) over one iteration's trajectories, alone on the chip;
(2) the rollout launch with capture only; (3) the rollout launch with the pass riding along."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
from swimmer_amd import kernels

torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
H = 1000
for n, N in ((6, 2048), (6, 256), (3, 2048), (3, 512)):
    p = sw.SwParams.make(n)
    R = 2 * N
    traj = torch.randn((H, p.d, R), dtype=torch.float64, device="cuda:0")
    acc = kernels.new_cov_acc(p, R, H, "cuda:0")
    for _ in range(3):
        kernels.traj_moments(p, traj, acc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        kernels.traj_moments(p, traj, acc)
    e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 10
    del traj, acc
    res = {}
    for tag, kw in (("capture only", dict(full_covariance=False, record_trajectories=True)),
                    ("capture + ride-along pass", dict(full_covariance=True))):
        ep = sw.EnvParam("B", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
        ap = sw.ARSParam("B", V1=False, n_iter=0, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                         threshold=0, initial_w="Zero")
        a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", **kw)
        # capture only: keep the trajectory stores, drop the per-iteration clone into the store (a 0.46 GB copy
        # between two launches lets the chip's power budget recover and flatters the next launch by ~10 %)
        a.record_trajectories = False
        for _ in range(30):
            a.run_iteration_async(want_returns=False)
            a.database._device_batches.clear()
        torch.cuda.synchronize()
        a._pipe.timing(1)
        for _ in range(10):
            a.run_iteration_async(want_returns=False)
            a.database._device_batches.clear()
        torch.cuda.synchronize()
        res[tag] = a._pipe.rollout_ms()[0]
        del a
    byts = R * H * p.d * 8
    print(f"n={n} N={N:5d}: standalone pass {alone * 1e3:7.1f} us ({byts / alone / 1e6:6.0f} GB/s of states); rollout launch "
          + ", ".join(f"{k} {v:.4f} ms" for k, v in res.items())
          + f"; riding along costs {1e3 * (res['capture + ride-along pass'] - res['capture only']):+.1f} us", flush=True)
