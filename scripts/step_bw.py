"""Step-kernel bandwidth at large batches (design aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
p = sw.SwParams.make(3)
rng = np.random.default_rng(0)
for B in (1 << 22, 1 << 24):
    st = torch.as_tensor(rng.uniform(-1, 1, (8, B)), device="cuda:0")
    ac = torch.as_tensor(rng.uniform(-1, 1, (2, B)), device="cuda:0")
    out = torch.empty_like(st); rew = torch.empty(B, dtype=torch.float64, device="cuda:0")
    plan = sw.kernels.StepPlan(p, st, ac, out, rew)
    best = 1e9
    for rep in range(3):
        for _ in range(5): plan.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.launch()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"B={B}: {best*1e3:.1f} us  {152*B/best/1e6:.0f} GB/s")
