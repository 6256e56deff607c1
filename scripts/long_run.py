import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
ep = sw.EnvParam("LeonSwimmer-Bench", n=3, H=1000, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
ap = sw.ARSParam("Bench", V1=False, n_iter=1, H=1000, N=512, b=512, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
agent = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", full_covariance=True)
for _ in range(5): agent.run_iteration_async(want_returns=False)
torch.cuda.synchronize()
for chunk in range(6):
    t0 = time.perf_counter()
    for _ in range(1000): agent.run_iteration_async(want_returns=False)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"chunk {chunk}: {dt/1000*1e3:.4f} ms/iter (host enqueue done after {t_host/1000*1e3:.4f} ms/iter)", flush=True)
