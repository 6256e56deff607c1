"""Step-kernel sweep of SURVEY 8(d) C2: n = 3 (and 6), batches 2^13 ... 2^24 envs, 1000
back-to-back launches at the small sizes.  Locates the launch-bound / cache-resident /
HBM-bound regimes (design aid; writes JSON lines to stdout)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw

rng = np.random.default_rng(0)
for n in (3, 6):
    p = sw.SwParams.make(n)
    d, m = 2 * n + 2, n - 1
    for e in range(int(os.environ.get('SWEEP_FROM', 13)), 25):
        B = 1 << e
        st = torch.as_tensor(rng.uniform(-1, 1, (d, B)), device="cuda:0")
        ac = torch.as_tensor(rng.uniform(-1, 1, (m, B)), device="cuda:0")
        out = torch.empty_like(st); rew = torch.empty(B, dtype=torch.float64, device="cuda:0")
        plan = sw.kernels.StepPlan(p, st, ac, out, rew)
        reps = 1000 if e <= 18 else 50
        best = 1e9
        for rep in range(3):
            for _ in range(5): plan.launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): plan.launch()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        byts = (2 * d + m + 1) * 8 * B
        print(json.dumps({"n": n, "envs": B, "launches": reps, "us_per_launch": round(best * 1e3, 3),
                          "env_steps_per_s": round(B / best * 1e3), "bytes_per_launch": byts,
                          "algorithmic_GBps": round(byts / best / 1e6, 1)}), flush=True)
        del st, ac, out, rew, plan
