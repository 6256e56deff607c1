"""End-to-end sanity: ARS V2 training on the swimmer with the GPU path (SEGMENTS=3 by default; N directions,
ITERS iterations from the environment).  Prints the learning curve (mean of the 2N returns per iteration,
ars_agent.py:199-201).  profiles/r03_j_train_demo_n6.log: SEGMENTS=6 N=256 ITERS=200."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
N, H, iters = int(os.environ.get("N", 512)), 1000, int(os.environ.get("ITERS", 300))
n = int(os.environ.get("SEGMENTS", 3))
ep = sw.EnvParam("LeonSwimmer-RealWorld", n=n, H=H, l_i=.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)   # ars/plot_graph.py:14-16
ap = sw.ARSParam("RLControl", V1=False, n_iter=iters, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
agent = sw.ARSAgent(ep, ap, seed=0, full_covariance=False)
t0 = time.perf_counter()
curve = []
for j in range(iters + 1):
    r = agent.run_iteration_async()
    if j % 25 == 0:
        curve.append((j, float(r.mean().item()), float(r.max().item())))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for j, m, mx in curve:
    print(f"iteration {j:4d}: mean return {m:10.4f}  best {mx:10.4f}")
bad = int((agent._status != 0).sum().item())
print(f"{iters+1} iterations of {2*N} rollouts x {H} steps (n = {n}) in {dt:.2f} s = {(iters+1)*2*N*H/dt:.3e} env-steps/s; "
      f"|P|_F = {np.linalg.norm(agent.policy):.4f}; bad rollouts {bad}")
