"""Host-side cost of one pipelined ARS iteration: with H = 10 the GPU needs ~15 us per iteration,
so the loop time IS the host time (noise generation, pinned H2D + wait, two launches, Python)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw

torch.cuda.set_stream(torch.cuda.Stream())
for n, N in ((3, 512), (3, 4096), (6, 2048)):
    ep = sw.EnvParam("x", n=n, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("x", V1=False, n_iter=1, H=10, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                     threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=0, full_covariance=True)
    for _ in range(1200): agent.run_iteration_async(want_returns=False)   # past the one-time ~35 ms runtime stall near launch 1000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): agent.run_iteration_async(want_returns=False)
    host = (time.perf_counter() - t0) / 500
    torch.cuda.synchronize()
    print(f"n={n} directions={N}: {host*1e6:.1f} us of host time per iteration", flush=True)
    del agent
