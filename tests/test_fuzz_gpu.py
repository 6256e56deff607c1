"""Larger-than-usual policies (actions of tens instead of tenths): the segment-per-lane kernels
against the oracle where the dynamics are violent and the solve is less comfortable."""
import numpy as np
import pytest
import torch

import oracle
from conftest import PARAM_SETS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [3, 4, 6, 8])
@pytest.mark.parametrize("scale", [0.3, 1.0, 3.0])
def test_violent_policies_stay_close_to_the_oracle(n, scale):
    import swimmer_amd as sw
    rng = np.random.default_rng(100 * n + int(10 * scale))
    d, m, R, H = 2 * n + 2, n - 1, 48, 150
    l, mm, k, h = PARAM_SETS["realworld"]
    pol = scale * rng.uniform(-1, 1, (R, m, d))
    op = oracle.OracleParams.make(n, l, mm, k, h)
    ref_ret, ref_traj = oracle.rollout_batch(op, H, pol, None, None, want_traj=True)
    worst = {}
    for kernel in ("lane", "quad"):
        p = sw.SwParams.make(n, l, mm, k, h, flags=sw._lib.kernel_flags(kernel))
        traj = torch.empty((H, d, R), dtype=torch.float64, device="cuda:0")
        status = torch.zeros(R, dtype=torch.int32, device="cuda:0")
        ret = sw.kernels.rollout(p, H, torch.as_tensor(pol, device="cuda:0"), traj=traj, status=status)
        tr = traj.permute(2, 0, 1).cpu().numpy()
        assert int(status.abs().sum()) == 0
        mag = max(1.0, np.abs(ref_traj).max())
        worst[kernel] = np.abs(tr - ref_traj).max() / mag
        assert np.abs(ret.cpu().numpy() - ref_ret).max() <= 1e-7 * max(1.0, np.abs(ref_ret).max())
    print(f"n={n} scale={scale}: |state| up to {np.abs(ref_traj).max():.1f}, "
          f"relative deviation lane {worst['lane']:.2e} quad/row {worst['quad']:.2e}")
    assert max(worst.values()) <= 1e-8
