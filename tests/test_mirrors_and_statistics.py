"""Public mirrors of the reference that the ARS loop does not exercise in the form a user may
call them (against reference-generated goldens, tests/golden/mirrors.npz), the update kernel
beyond its LDS capacity, and the V2 running statistics over LONG runs (tests/golden/long.npz:
a reference run that learns; plus 300 iterations against the oracle)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sw():
    import swimmer_amd
    return swimmer_amd


@pytest.fixture(scope="module")
def mirrors():
    return np.load(os.path.join(GOLDEN, "mirrors.npz"), allow_pickle=False)


@pytest.mark.parametrize("n", [3, 6])
def test_select_action_v1_and_v2(sw, mirrors, n):
    """Environment.select_action (ars/environment.py:19-35)."""
    g = mirrors
    ep = sw.EnvParam("LeonSwimmer-Test", n=n, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    env = sw.Environment(ep)
    P, mean, cov = g[f"sel_n{n}_policy"], g[f"sel_n{n}_mean"], g[f"sel_n{n}_cov"]
    for o, a1, a2 in zip(g[f"sel_n{n}_obs"], g[f"sel_n{n}_v1"], g[f"sel_n{n}_v2"]):
        got1 = env.select_action(P, o.tolist())
        got2 = env.select_action(P, o.tolist(), cov, mean)
        assert isinstance(got1, np.ndarray) and got1.shape == (n - 1,)
        assert np.abs(got1 - a1).max() <= 1e-13 * max(1.0, np.abs(a1).max())
        assert np.abs(got2 - a2).max() <= 1e-12 * max(1.0, np.abs(a2).max())
        # one of covariance / mean missing -> V1, like the reference's `or`
        assert np.array_equal(env.select_action(P, o.tolist(), cov, None), got1)


def test_update_policy_with_an_order_subset(sw, mirrors):
    """ARSAgent.update_policy(deltas, rewards, order) (ars_agent.py:110-130) with an order that
    is 3 of the 7 directions, not sorted: only those enter sigma_R and the step."""
    g = mirrors
    alpha, b = g["upd_alpha_b"]
    ep = sw.EnvParam("LeonSwimmer-Test", n=3, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("Test", V1=True, n_iter=1, H=10, N=7, b=int(b), alpha=float(alpha), nu=0.01,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=0)
    deltas = [x for x in g["upd_deltas"]]
    rewards = g["upd_rewards"].tolist()
    assert agent.sort_directions(deltas, rewards) == g["upd_sorted"].tolist()
    agent.policy = g["upd_policy0"]
    agent.update_policy(deltas, rewards, g["upd_order"].tolist())
    assert np.abs(agent.policy - g["upd_policy1"]).max() <= 1e-14
    # the full sorted order gives a different step (so the subset really was honoured)
    agent.policy = g["upd_policy0"]
    agent.update_policy(deltas, rewards, agent.sort_directions(deltas, rewards))
    assert np.abs(agent.policy - g["upd_policy1"]).max() > 1e-4


def test_warm_start_from_a_policy_file(sw, mirrors, tmp_path):
    """initial_w=<path>.npy (ars_agent.py:74-81), then two V2 iterations."""
    g = mirrors
    path = str(tmp_path / "w0.npy")
    np.save(path, g["warm_w0"])
    ep = sw.EnvParam("LeonSwimmer-Test", n=3, H=80, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
    ap = sw.ARSParam("Test", V1=False, n_iter=2, H=80, N=4, b=4, alpha=0.0075, nu=0.01,
                     safe=False, threshold=0, initial_w=path)
    agent = sw.ARSAgent(ep, ap, seed=8)
    assert np.array_equal(agent.policy, g["warm_w0"])
    for it in range(2):
        r = np.array(agent.runOneIteration())
        assert np.abs(r - g["warm_rewards"][it]).max() <= 1e-9 * max(1.0, np.abs(g["warm_rewards"][it]).max())
        assert np.abs(agent.policy - g["warm_policies"][it]).max() <= 1e-9
    assert np.abs(agent.mean - g["warm_mean"]).max() <= 1e-10
    sd = np.sqrt(np.diag(g["warm_cov"]))
    assert (np.abs(agent.covariance - g["warm_cov"]) <= 1e-7 * np.outer(sd, sd)).all()
    with pytest.raises(AssertionError):           # wrong shape, like the reference's assert
        np.save(path, np.zeros((3, 8)))
        sw.ARSAgent(ep, ap, seed=8)


@pytest.mark.parametrize("top_b", [0, 5000])
def test_update_beyond_the_lds_capacity(sw, top_b):
    """n_dir = 8192 > 6144 directions: the update kernel's global-memory branch (returns and
    ranking straight from HBM).  Policy, sigma_R and the selection against NumPy."""
    N, n = 8192, 3
    m, d = n - 1, 2 * n + 2
    rs = np.random.RandomState(5)
    deltas = 2 * rs.rand(N, m, d) - 1
    rets = rs.uniform(-3, 8, 2 * N)
    rets[100:120] = rets[100]                  # ties: broken towards the higher index
    P0 = rs.uniform(-0.5, 0.5, (m, d))
    alpha, b = 0.02, 64.0
    dev = "cuda:0"
    p = sw.SwParams.make(n)
    pol = torch.as_tensor(P0.copy(), device=dev)
    sig = torch.zeros(1, dtype=torch.float64, device=dev)
    sw.kernels.ars_update(p, torch.as_tensor(rets, device=dev), torch.as_tensor(deltas, device=dev),
                          pol, alpha, b, top_b, sigma_out=sig)
    r = rets.reshape(N, 2)
    if top_b:
        order = np.argsort(r.max(axis=1), kind="stable")[::-1][:top_b]   # ars_agent.py:105-108
    else:
        order = np.arange(N)
    used = r[order].reshape(-1)
    sigma = np.std(used)
    # divisor: b when every direction is used (ars_agent.py:128), len(order) with a true top-b
    # truncation (safe_ars/ars.py:64)
    grad = ((r[order, 0] - r[order, 1])[:, None, None] * deltas[order]).sum(0) / ((top_b or b) * sigma)
    want = P0 + alpha * grad
    assert abs(sig.item() - sigma) <= 1e-13 * sigma
    assert np.abs(pol.cpu().numpy() - want).max() <= 1e-11 * np.abs(want).max()


# ---- V2 statistics over long runs ---------------------------------------------------------
def test_long_learning_run_vs_reference(sw):
    """tests/golden/long.npz: the REFERENCE ran 60 ARS V2 iterations (n = 3, realworld
    parameters, N = 8, H = 200, alpha 0.02, nu 0.03, seed 4) and recorded policy / mean /
    diag(cov) / returns every 10 iterations.  The mean return climbs from -7e-6 to 9e-4, i.e. the
    state distribution moves away from the reset pivot while the statistics keep every state
    since iteration 0 (ars_agent.py:179-182)."""
    g = np.load(os.path.join(GOLDEN, "long.npz"), allow_pickle=False)
    n, N, H, iters, every, seed = (int(x) for x in g["long_cfg"])
    l_i, m_i, k, h, alpha, nu = (float(x) for x in g["long_phys"])
    ep = sw.EnvParam("LeonSwimmer-Test", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
    ap = sw.ARSParam("Test", V1=False, n_iter=iters, H=H, N=N, b=N, alpha=alpha, nu=nu,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=seed)
    j = 0
    worst = {"curve": 0.0, "ret": 0.0, "pol": 0.0, "mean": 0.0, "inv_std": 0.0}
    cscale = np.abs(g["long_curve"]).max()
    for it in range(iters):
        r = np.array(agent.runOneIteration())
        worst["curve"] = max(worst["curve"], abs(r.mean() - g["long_curve"][it]) / cscale)
        if (it + 1) % every == 0:
            scale = max(1e-3, np.abs(g["long_rewards"][j]).max())
            worst["ret"] = max(worst["ret"], np.abs(r - g["long_rewards"][j]).max() / scale)
            worst["pol"] = max(worst["pol"], np.abs(agent.policy - g["long_policies"][j]).max())
            worst["mean"] = max(worst["mean"], np.abs(agent.mean - g["long_means"][j]).max())
            inv = agent._inv_std.cpu().numpy()
            worst["inv_std"] = max(worst["inv_std"],
                                   np.abs(inv / g["long_diag_covs"][j] ** -0.5 - 1.0).max())
            j += 1
    print("long run vs reference, worst deviations:", worst)
    assert j == iters // every
    # What limits the agreement is not the statistics algorithm (the test below bounds that at
    # 1e-15) but the closed loop: the swimmer barely moves in these 60 iterations (returns
    # 1e-8 .. 1e-3, sums of velocities of 1e-2 that cancel), so rounding-level ABSOLUTE
    # differences of ~1e-13 in a return are ~1e-7 RELATIVE, and every policy step divides by the
    # returns' standard deviation.  Observed on MI355X: curve 3.1e-7, returns 9.4e-7, policy
    # 5.2e-8, mean 1.5e-8, inv_std 1.1e-6.  (The C oracle, which keeps the reference's evaluation
    # order and is 94-99 % bit-identical per step, lands at 6e-9 / 5e-10 / 1e-10 / 1e-8.)
    # Bars: the contract's 1e-5 on returns and whitening, 1e-6 on policy and mean.
    assert worst["curve"] <= 1e-5 and worst["ret"] <= 1e-5
    assert worst["pol"] <= 1e-6 and worst["mean"] <= 1e-6
    assert worst["inv_std"] <= 1e-5
    sd = np.sqrt(np.diag(g["long_full_cov_last"]))
    assert (np.abs(agent.covariance - g["long_full_cov_last"]) <= 1e-5 * np.outer(sd, sd)).all()


def test_statistics_accumulation_error_over_300_iterations(sw):
    """What the device's statistics ALGORITHM loses, separated from trajectory differences:
    every state of 300 iterations (N = 8, H = 200: 960 000 states) is recorded, and the device's
    running mean / inv_std (per-iteration batches merged with Chan's update) are compared with a
    two-pass float128 mean / variance over exactly those states, every 50 iterations.
    The reference's own two-pass float64 np.mean / np.cov is no closer to that than 1e-13."""
    N, H, iters = 8, 200, 300
    ep = sw.EnvParam("LeonSwimmer-Test", n=3, H=H, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
    ap = sw.ARSParam("Test", V1=False, n_iter=iters, H=H, N=N, b=N, alpha=0.02, nu=0.03,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=4, full_covariance=False, record_trajectories=True)
    s1 = np.zeros(8, dtype=np.longdouble)
    chunks = []
    report = []
    for it in range(iters):
        agent.run_iteration_async(want_returns=False)
        traj = agent.database._device_batches.pop()[0]          # [H, d, 2N] of this iteration
        chunks.append(traj.permute(2, 0, 1).reshape(-1, 8).cpu().numpy())
        if (it + 1) % 50 == 0:
            x = np.concatenate(chunks).astype(np.longdouble)
            mu = x.mean(axis=0)
            var = ((x - mu) ** 2).sum(axis=0) / (x.shape[0] - 1)
            dev_mean = agent.mean.astype(np.longdouble)
            dev_inv = agent._inv_std.cpu().numpy().astype(np.longdouble)
            e_mean = float(np.abs(dev_mean - mu).max())
            e_inv = float(np.abs(dev_inv * np.sqrt(var) - 1).max())
            x64 = x.astype(np.float64)
            e_ref = float(np.abs(np.diag(np.cov(x64.T)).astype(np.longdouble) ** -0.5 * np.sqrt(var) - 1).max())
            report.append((it + 1, e_mean, e_inv, e_ref))
    for it, e_mean, e_inv, e_ref in report:
        print(f"iteration {it}: |mean - exact| {e_mean:.2e}, inv_std rel. dev. {e_inv:.2e} "
              f"(float64 two-pass np.cov on the same states: {e_ref:.2e})")
    assert agent.n_saved_states == iters * 2 * N * H
    assert max(r[1] for r in report) <= 1e-13
    assert max(r[2] for r in report) <= 1e-12       # VERDICT asked for a bound below ~1e-7
    # no growth with the length of training
    assert report[-1][2] <= 10 * max(report[0][2], 1e-15)
    assert np.mean(agent.runOneIteration()) > 1e-4   # and it did learn to move
