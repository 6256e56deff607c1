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


@pytest.mark.parametrize("n", [3, 5, 6])
def test_angles_carried_in_reduced_form_across_many_quadrants(n):
    """The segment-per-lane kernels carry theta = r + K pi/2 and re-normalise r on a rare path
    written in assembly (swimmer_device.h, angle_keep_reduced).  Start states spread over +-50 rad
    (negative angles, every quadrant) with angular velocities up to +-30 rad/s (0.03 rad per step:
    every segment crosses several re-normalisation boundaries, in both directions, in 120 steps)
    against the oracle, which evaluates libm sin / cos of the full angle every step."""
    import swimmer_amd as sw
    rng = np.random.default_rng(7 + n)
    d, m, R, H = 2 * n + 2, n - 1, 64, 120
    l, mm, k, h = PARAM_SETS["default"]
    st = np.empty((R, d))
    st[:, 0:2] = rng.uniform(-0.5, 0.5, (R, 2))
    st[:, 2::2] = rng.uniform(-50.0, 50.0, (R, n))
    st[:, 3::2] = rng.uniform(-30.0, 30.0, (R, n))
    st[:8, 2::2] = np.round(st[:8, 2::2] / (np.pi / 4)) * (np.pi / 4)     # on the re-normalisation boundaries
    pol = 0.05 * rng.uniform(-1, 1, (R, m, d))
    op = oracle.OracleParams.make(n, l, mm, k, h)
    ref = [oracle.rollout(op, H, pol[r], state0=st[r]) for r in range(R)]
    ref_ret = np.array([x[0] for x in ref])
    ref_traj = np.array([x[1] for x in ref])
    s0 = torch.as_tensor(np.ascontiguousarray(st.T), device="cuda:0")
    for kernel in ("lane", "quad"):
        p = sw.SwParams.make(n, l, mm, k, h, flags=sw._lib.kernel_flags(kernel))
        traj = torch.empty((H, d, R), dtype=torch.float64, device="cuda:0")
        fin = torch.empty((d, R), dtype=torch.float64, device="cuda:0")
        status = torch.zeros(R, dtype=torch.int32, device="cuda:0")
        ret = sw.kernels.rollout(p, H, torch.as_tensor(pol, device="cuda:0"), state0=s0, traj=traj,
                                 final_state=fin, status=status)
        tr = traj.permute(2, 0, 1).cpu().numpy()
        assert int(status.abs().sum()) == 0
        dev = np.abs(tr - ref_traj).max() / np.abs(ref_traj).max()
        print(f"n={n} {kernel}: |theta| up to {np.abs(ref_traj[:, :, 2::2]).max():.0f} rad, "
              f"relative deviation {dev:.2e}")
        assert dev <= 1e-9
        assert np.abs(ret.cpu().numpy() - ref_ret).max() <= 1e-8 * max(1.0, np.abs(ref_ret).max())
        assert np.array_equal(fin.T.cpu().numpy(), tr[:, -1, :])


@pytest.mark.parametrize("kernel", ["lane", "quad"])
@pytest.mark.parametrize("n", [3, 6])
def test_rollout_status_bits_for_hopeless_inputs(n, kernel):
    """Start angle beyond the supported range -> SW_STATUS_RANGE and a NaN return; a policy that
    blows the swimmer up -> flagged (range or non-finite), never silent garbage, never a hang."""
    import swimmer_amd as sw
    d, m, R, H = 2 * n + 2, n - 1, 20, 300
    p = sw.SwParams.make(n, flags=sw._lib.kernel_flags(kernel))
    st = sw.kernels.reset(p, R)
    st[2, 3] = 5.0e9                                   # rollout 3 starts out of range
    pol = torch.zeros((R, m, d), dtype=torch.float64, device="cuda:0")
    pol[7] = 1.0e7                                     # rollout 7: positive feedback on everything
    status = torch.zeros(R, dtype=torch.int32, device="cuda:0")
    ret = sw.kernels.rollout(p, H, pol, state0=st, status=status)
    torch.cuda.synchronize()
    s = status.cpu().numpy()
    assert s[3] & 4 and bool(torch.isnan(ret[3]))
    assert s[7] & (2 | 4), s[7]
    ok = np.ones(R, dtype=bool)
    ok[[3, 7]] = False
    assert (s[ok] == 0).all() and bool(torch.isfinite(ret[torch.as_tensor(ok)]).all())


@pytest.mark.parametrize("kernel", ["lane", "quad"])
@pytest.mark.parametrize("n", [3, 6])
def test_mirror_symmetry_of_whole_rollouts_at_baseline_size(n, kernel):
    """A size-independent property at BASELINE size (1024 rollouts x H = 1000), no oracle needed:
    reflecting the swimmer about the x axis, M = diag(1, -1, -1, -1, ...) on (Gdx, Gdy, th_i,
    thd_i), flips the sign of every joint torque, so the V1 policy P' = -P M started from the
    mirrored state runs the mirrored trajectory and collects the SAME return (reward = Gdot_x).
    Exercises negative angles (start at -pi/2), the reduced-angle tracking in both directions and
    every sign in the dynamics."""
    import swimmer_amd as sw
    rng = np.random.default_rng(31 + n)
    d, m, R, H = 2 * n + 2, n - 1, 1024, 1000
    p = sw.SwParams.make(n, 0.8, 1.2, 10.2, 1e-3, flags=sw._lib.kernel_flags(kernel))
    M = -np.ones(d)
    M[0] = 1.0
    pol = 0.3 * rng.uniform(-1, 1, (R, m, d))
    st = np.zeros((d, R))
    st[2::2] = np.pi / 2 + rng.uniform(-0.3, 0.3, (n, R))
    st[3::2] = rng.uniform(-1, 1, (n, R))
    st[0:2] = rng.uniform(-0.2, 0.2, (2, R))
    dev = "cuda:0"
    out = []
    for sign, s0, pl in ((1, st, pol), (-1, M[:, None] * st, -pol * M[None, None, :])):
        traj = torch.empty((H, d, R), dtype=torch.float64, device=dev)
        status = torch.zeros(R, dtype=torch.int32, device=dev)
        ret = sw.kernels.rollout(p, H, torch.as_tensor(np.ascontiguousarray(pl), device=dev),
                                 state0=torch.as_tensor(np.ascontiguousarray(s0), device=dev), traj=traj,
                                 status=status)
        assert int(status.abs().sum()) == 0
        out.append((ret.cpu().numpy(), traj.cpu().numpy()))
    (r1, t1), (r2, t2) = out
    scale = np.abs(t1).max()
    dev_t = np.abs(t1 - M[None, :, None] * t2).max() / scale
    dev_r = np.abs(r1 - r2).max() / max(1.0, np.abs(r1).max())
    print(f"n={n} {kernel}: mirrored rollouts deviate by {dev_t:.2e} (states), {dev_r:.2e} (returns)")
    assert dev_t <= 1e-9 and dev_r <= 1e-9
