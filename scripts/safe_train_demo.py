"""End-to-end run of the safe-exploration experiment the reference describes (safe_ars/experiment.py:28-70) on the GPU
path: a real swimmer (m, l, k) = (1, 1, 10), a simulator whose parameters are off by EPSILON in a random direction, the
cost "maximum speed angle" max_i |thetadot_i| (:45), Basic_ARS and Safe_ARS trained from the same seed; prints both
learning curves, the worst cost either agent ever reached in the real world, and how many rollouts the gate stopped.
Plots, argparse and the seeds loop of the reference script are not reproduced (design aid).
    N=64 B=32 ITERS=60 H=500 THRESH=3.0 EPSILON=0.05 python scripts/safe_train_demo.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import swimmer_amd as sw

N, b, iters, H = (int(os.environ.get(k, v)) for k, v in (("N", 64), ("B", 32), ("ITERS", 60), ("H", 500)))
thresh, eps = float(os.environ.get("THRESH", 3.0)), float(os.environ.get("EPSILON", 0.05))
alpha, nu, seed, n = 0.02, 0.03, 7, 3
theta_real = np.array([1.0, 1.0, 10.0])
rs = np.random.RandomState(1)
delta = rs.rand(3)
theta_sim = theta_real + delta / np.linalg.norm(delta) * eps                        # experiment.py:36-38
real = sw.SwimmerEnv("RealWorld", n=n, m_i=theta_real[0], l_i=theta_real[1], k=theta_real[2])
sim = sw.SwimmerEnv("Simulator", n=n, m_i=theta_sim[0], l_i=theta_sim[1], k=theta_sim[2])
cost = sw.safe_ars.MaxAbsThetaDot()


def worst_cost(states):
    return float(np.abs(states[:, :, 3::2]).max())


out = {}
for name, agent in (("basic", sw.safe_ars.Basic_ARS()), ("safe", sw.safe_ars.Safe_ARS(cost, thresh, thresh - 1.0, sim))):
    np.random.seed(seed)
    t0 = time.perf_counter()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        curve, states = agent.train(iters, real, N, b, alpha, nu, H)
    dt = time.perf_counter() - t0
    out[name] = (curve, worst_cost(states), dt)
    extra = ""
    if name == "safe":
        same = np.all(states[:, 1:] == states[:, :-1], axis=2)
        stopped = int(same[:, -1].sum())
        extra = f"; rollouts the gate stopped: {stopped} of {len(states)}; real steps over the threshold: {agent.real_violations}"
    print(f"{name:5s}: {iters} iterations of {2 * N} rollouts x {H} steps in {dt:.2f} s; worst cost reached {out[name][1]:.3f} "
          f"(threshold {thresh}){extra}")
for j in range(0, iters, max(1, iters // 12)):
    print(f"iteration {j:4d}: mean return  basic {out['basic'][0][j]:10.5f}   safe {out['safe'][0][j]:10.5f}")
print(f"last        : mean return  basic {out['basic'][0][-1]:10.5f}   safe {out['safe'][0][-1]:10.5f}")
