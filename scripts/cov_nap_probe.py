"""n = 6, 2048 directions on one GPU (a rollout wave on every SIMD): rollout launch with the covariance
pass riding along, for the load pacing given by SWIMMER_COV_NAP (read once per process).  Design aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
n, N, H = int(os.environ.get('PN', 6)), int(os.environ.get('PNDIR', 2048)), 1000
res = {}
which = os.environ.get("PMODE", "both")   # capture | ride | both (one agent per process avoids order effects)
for tag, kw in (("capture only", dict(full_covariance=False, record_trajectories=True)),
                ("capture + ride-along pass", dict(full_covariance=True))):
    if which != "both" and (which == "capture") != (tag == "capture only"):
        continue
    ep = sw.EnvParam("B", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("B", V1=False, n_iter=0, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                     threshold=0, initial_w="Zero")
    a = sw.ARSAgent(ep, ap, seed=0, device="cuda:0", **kw)
    # capture only: keep the trajectory stores, drop the per-iteration clone into the store (a 0.46 GB copy
    # between two launches lets the chip's power budget recover and flatters the next launch by ~10 %)
    a.record_trajectories = False
    for _ in range(30):
        a.run_iteration_async(want_returns=False)
        a.database._device_batches.clear()
    torch.cuda.synchronize()
    a._pipe.timing(1)
    for _ in range(30):
        a.run_iteration_async(want_returns=False)
        a.database._device_batches.clear()
    torch.cuda.synchronize()
    res[tag] = a._pipe.rollout_ms()[0]
    del a
print(f"SWIMMER_COV_NAP={os.environ.get('SWIMMER_COV_NAP', '0'):>3} SWIMMER_COV_PRIO={os.environ.get('SWIMMER_COV_PRIO', '0')} n={n} N={N}: " + ", ".join(f"{k} {v:.4f} ms" for k, v in res.items())
      + (f"; riding along costs {1e3 * (res['capture + ride-along pass'] - res['capture only']):+.1f} us" if len(res) == 2 else ""), flush=True)
