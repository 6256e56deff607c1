"""Where does an ARS iteration's wall time go: host enqueue time vs GPU time (design aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw

N, H = 512, 1000
ep = sw.EnvParam("x", n=3, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
ap = sw.ARSParam("x", V1=False, n_iter=1, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
agent = sw.ARSAgent(ep, ap, seed=0)
for _ in range(5): agent.run_iteration_async()
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): agent.run_iteration_async()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.3f} ms/iter, total {1e3*(t2-t0)/K:.3f} ms/iter")
# pieces
t0 = time.perf_counter()
for _ in range(K):
    h = agent._deltas_host_np[0]; h[...] = np.random.rand(N, 2, 8); h *= 2; h -= 1
t1 = time.perf_counter()
print(f"host RNG + scale into pinned: {1e6*(t1-t0)/K:.1f} us")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): agent._pipe.host_slot_wait(0)
t1 = time.perf_counter()
print(f"host_slot_wait: {1e6*(t1-t0)/K:.1f} us")

# per-call host cost inside the iteration
import time as _t
ap_ = agent.agent_param
acc = {"fill": 0.0, "rollouts": 0.0, "update": 0.0}
torch.cuda.synchronize()
for _ in range(K):
    i = agent._it & 1; agent._it += 1
    a = _t.perf_counter()
    agent._pipe.host_slot_wait(i)
    host = agent._deltas_host_np[i]; host[...] = np.random.rand(ap_.N, 2, 8); host *= 2; host -= 1
    b = _t.perf_counter()
    agent._pipe.rollouts(i, agent.params, ap_.N, 0, ap_.N, ap_.H, agent._deltas_host[i], agent._deltas2[i], agent._policy, ap_.nu,
                         agent._mean, agent._inv_std, agent._returns_local, agent._traj2[i], agent._moments_local, agent._cov_acc, agent._status)
    c = _t.perf_counter()
    agent._pipe.update(i, agent.params, ap_.N, agent._returns_local, agent._deltas2[i], agent._policy, ap_.alpha, ap_.b, 0,
                       agent._moments_local, agent._running, 2 * ap_.N * ap_.H, agent._mean, agent._inv_std, agent._sigma)
    d = _t.perf_counter()
    acc["fill"] += b - a; acc["rollouts"] += c - b; acc["update"] += d - c
torch.cuda.synchronize()
print({k: f"{1e6 * v / K:.1f} us" for k, v in acc.items()})
