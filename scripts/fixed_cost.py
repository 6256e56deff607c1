"""Fixed cost of a rollout launch: kernel time vs horizon (design aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw

dev = "cuda:0"
for n, N in ((3, 512), (6, 256)):
    p = sw.SwParams.make(n)
    d, m = 2 * n + 2, n - 1
    rng = np.random.default_rng(0)
    deltas = torch.as_tensor(rng.uniform(-1, 1, (N, m, d)), device=dev)
    pol = torch.zeros((m, d), dtype=torch.float64, device=dev)
    mean = torch.zeros(d, dtype=torch.float64, device=dev)
    inv = torch.ones(d, dtype=torch.float64, device=dev)
    for H in (1, 2, 10, 100, 1000):
        ret = torch.empty(2 * N, dtype=torch.float64, device=dev)
        traj = torch.empty((H, d, 2 * N), dtype=torch.float64, device=dev)
        mom = torch.zeros((sw.kernels.moments_blocks(2 * N), 2 * d), dtype=torch.float64, device=dev)
        status = torch.zeros(2 * N, dtype=torch.int32, device=dev)
        def f():
            sw.kernels.ars_rollouts(p, H, pol, deltas, 0.01, 0, N, mean=mean, inv_std=inv, returns=ret,
                                    traj=traj, moments=mom, status=status)
        for _ in range(5): f()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"n={n} rollouts={2*N} H={H:5d}: {best*1e3:8.2f} us per launch", flush=True)
