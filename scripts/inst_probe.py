"""Rollout launch time of the OTHER template instantiations of the n = PN rollout kernel (the headline one is
ARS + capture + moments): V2 without capture, V1 without capture, V1 with capture.  For loop-placement sweeps
(SWIMMER_HIP_LIB=... over builds with -DSW_OCT_LOOP_PAD=k ...).  Design aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
n, N, H = int(os.environ.get('PN', 3)), int(os.environ.get('PNDIR', 512)), 1000
out = []
for tag, v1, kw in (("V2 no capture", False, dict(full_covariance=False)),
                    ("V1 no capture", True, dict()),
                    ("V1 capture", True, dict(record_trajectories=True))):
    ep = sw.EnvParam("B", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("B", V1=v1, n_iter=0, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False, threshold=0, initial_w="Zero")
    a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", **kw)
    a.record_trajectories = False          # keep the stores, drop the per-iteration clone (see cov_probe.py)
    for _ in range(30):
        a.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    a._pipe.timing(1)
    for _ in range(30):
        a.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    out.append(f"{tag} {a._pipe.rollout_ms()[0]:.4f}")
    del a
print(f"n={n} N={N}: " + "; ".join(out) + " ms", flush=True)
