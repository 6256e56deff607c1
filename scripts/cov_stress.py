"""Stress of the covariance pass's cross-workgroup hand-over (agent-scope row stores, ticket, merge by the last tile):
many passes over fresh random trajectories of many shapes; every pass is run twice into fresh accumulators (bits must
agree) and compared with the sums torch computes in fp64.  A stale row in the merge would show as a gross error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import swimmer_amd as sw
from swimmer_amd import kernels

SHAPES = [(3, 1024, 1000), (3, 4096, 250), (6, 512, 1000), (6, 4096, 125), (8, 256, 400), (5, 333, 77), (2, 64, 1000),
          (4, 1000, 129), (7, 65, 513)]


def run(reps, dev="cuda:0", shapes=SHAPES):
    torch.manual_seed(0)
    worst, passes = 0.0, 0
    for rep in range(reps):
        for n, R, H in shapes:
            worst = max(worst, one_shape(n, R, H, dev))
            passes += 4
    return passes, worst


def one_shape(n, R, H, dev):
    p = sw.SwParams.make(n)
    d = p.d
    traj = torch.randn((H, d, R), dtype=torch.float64, device=dev)
    traj[:, 2::2, :] += 1.5707963267948966           # angles live around pi / 2 (the pass's pivot)
    accs = []
    for _ in range(2):
        acc = kernels.new_cov_acc(p, R, H, dev)
        kernels.traj_moments(p, traj, acc)
        kernels.traj_moments(p, traj, acc)          # a second pass into the same accumulator (ticket reset)
        accs.append(acc[:1 + d + d * d].clone())
    torch.cuda.synchronize()
    assert torch.equal(accs[0], accs[1]), ("bits differ", n, R, H)
    x = traj.clone()
    x[:, 2::2, :] -= 1.5707963267948966
    x = x.permute(1, 0, 2).reshape(d, -1)
    s1 = 2.0 * x.sum(dim=1)
    s2 = 2.0 * (x @ x.T)
    got = accs[0]
    assert got[0].item() == 2.0 * R * H
    e1 = ((got[1:1 + d] - s1).abs() / (1.0 + s1.abs())).max().item()
    e2 = ((got[1 + d:].view(d, d) - s2).abs() / (1.0 + s2.abs())).max().item()
    assert e1 < 1e-9 and e2 < 1e-9, ("sums differ", n, R, H, e1, e2)
    return max(e1, e2)


if __name__ == "__main__":
    torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
    passes, worst = run(int(os.environ.get("REPS", 30)))
    print(f"{passes} passes over {len(SHAPES)} shapes: bit-reproducible, worst relative deviation from torch fp64 {worst:.2e}")
