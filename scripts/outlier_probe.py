"""Where does `n = 6, N = 2048, no capture / no covariance` lose 4.4 ms per iteration
(profiles/r02_e_capture_and_covariance_cost.log)?  Host time of every stage of
run_iteration_async, per iteration, with and without the pipeline's launch timing (design aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import swimmer_amd as sw
from swimmer_amd import kernels

torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
n, N, H = int(os.environ.get("PN", 6)), int(os.environ.get("PNDIR", 2048)), 1000
for cov, timing in ((False, 0), (False, 1), (True, 1), (True, 0)):
    ep = sw.EnvParam("B", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("B", V1=False, n_iter=0, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                     threshold=0, initial_w="Zero")
    a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", full_covariance=cov)
    for _ in range(4):
        a.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    a._pipe.timing(timing)
    stages = {}
    pipe = a._pipe
    orig = {k: getattr(pipe, k) for k in ("host_slot_wait", "rollouts", "update")}

    def wrap(name, fn):
        def inner(*args, **kw):
            t = time.perf_counter()
            r = fn(*args, **kw)
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - t
            return r
        return inner
    for k, fn in orig.items():
        setattr(pipe, k, wrap(k, fn))
    iters = 12
    t0 = time.perf_counter()
    for _ in range(iters):
        a.run_iteration_async(want_returns=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms, k = a._pipe.rollout_ms() if timing else (float("nan"), 0)
    bad = int((a._status != 0).sum().item())
    print(f"n={n} N={N} covariance={cov!s:5} launch-timing={timing}: iteration {(t2 - t0) / iters * 1e3:.4f} ms "
          f"(enqueue loop {(t1 - t0) / iters * 1e3:.4f}, final sync {(t2 - t1) * 1e3:.3f} ms); rollout launch {ms:.4f} ms x{k}; "
          + ", ".join(f"{k} {v / iters * 1e3:.4f}" for k, v in stages.items()) + f"; bad rollouts {bad}", flush=True)
    del a
