"""Rollout launch time with the zero policy of the benchmark (no segment ever leaves its quadrant
centre) against a TRAINED policy (joints swinging tens of degrees: the reduced angles cross the
+-pi/4 re-normalisation boundaries all the time).  Design aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
ep = sw.EnvParam("B", n=3, H=1000, l_i=.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
ap = sw.ARSParam("B", V1=False, n_iter=0, H=1000, N=512, b=512, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", full_covariance=True)
def timed(tag, k=16):
    torch.cuda.synchronize()
    a._pipe.timing(1)
    rets = None
    for _ in range(k):
        rets = a.run_iteration_async()
    torch.cuda.synchronize()
    ms, n = a._pipe.rollout_ms()
    a._pipe.timing(0)
    traj = a._traj
    th = traj[:, 2::2, :]
    print(f"{tag}: rollout launch {ms:.4f} ms over {n} launches; mean return {float(rets.mean()):.2f}; "
          f"theta range over the last batch [{float(th.min()):.2f}, {float(th.max()):.2f}] rad", flush=True)
timed("iteration   0-15 (zero policy)")
for block in range(3):
    for _ in range(100):
        a.run_iteration_async(want_returns=False)
    timed(f"iteration {100 * (block + 1) + 16 * block:3d}+ (training)")
