"""Per-iteration wall time of the ARS pipeline under different options (design aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw

def run(N=512, H=1000, K=40, **kw):
    ep = sw.EnvParam("x", n=3, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("x", V1=kw.pop("V1", False), n_iter=1, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=0, **kw)
    fixed = agent.sample_deltas() if os.environ.get("FIXED_DELTAS") else None
    for _ in range(5): agent.run_iteration_async(fixed)
    torch.cuda.synchronize()
    agent._pipe.timing(0 if os.environ.get("NO_TIMING") else int(os.environ.get("TIME_EVERY", "1")))
    t0 = time.perf_counter()
    for _ in range(K): agent.run_iteration_async(fixed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    ms, n = agent._pipe.rollout_ms() if not os.environ.get("NO_TIMING") else (0.0, 0)
    return dt * 1e3, ms

MODES = (("v2 full cov", {}), ("v2 diag only", dict(full_covariance=False)), ("v1", dict(V1=True)),
                 ("v2 full cov, lane kernel", dict(rollout_kernel="lane")))
sel = os.environ.get("MODE")
for name, kw in MODES:
    if sel and sel != name:
        continue
    tot, k = run(**dict(kw))
    print(f"{name:28s}: {tot:.4f} ms/iter, rollout kernel {k:.4f} ms, other {1e3*(tot-k):.1f} us")
