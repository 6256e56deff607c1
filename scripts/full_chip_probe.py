"""Rollout launch time when the batch puts a wave on every SIMD (2048 directions, n = 3 and 6),
with and without the covariance pass riding along (design aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
for n in ((3, 6) if not os.environ.get('ONLY_N') else (int(os.environ['ONLY_N']),)):
    for N in (512, 2048):
        for cov in (True,):
            ep = sw.EnvParam("B", n=n, H=1000, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
            ap = sw.ARSParam("B", V1=False, n_iter=0, H=1000, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                             threshold=0, initial_w="Zero")
            a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", full_covariance=(cov is True),
                            record_trajectories=(cov == "traj-only"))
            for _ in range(4):
                a.run_iteration_async(want_returns=False)
            torch.cuda.synchronize()
            a._pipe.timing(1)
            t0 = time.perf_counter()
            for _ in range(12):
                a.run_iteration_async(want_returns=False)
                a.database._device_batches.clear()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 12
            ms, k = a._pipe.rollout_ms()
            print(f"n={n} N={N:5d} capture/covariance={cov!s:9}: rollout launch {ms:.4f} ms, iteration {dt*1e3:.4f} ms", flush=True)
            del a
