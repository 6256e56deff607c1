"""Run the physics-step kernel a few times at a bandwidth-bound batch (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import swimmer_amd as sw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
n = 3
p = sw.SwParams.make(n)
rng = np.random.default_rng(0)
st = torch.as_tensor(rng.uniform(-1, 1, (8, B)), device="cuda:0")
ac = torch.as_tensor(rng.uniform(-1, 1, (2, B)), device="cuda:0")
out = torch.empty_like(st); rew = torch.empty(B, dtype=torch.float64, device="cuda:0")
for _ in range(10):
    sw.kernels.step(p, st, ac, out=out, reward=rew)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (2 * 8 + 2 + 1) * 8 * B)
