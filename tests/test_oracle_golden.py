"""Pins the CPU oracle (oracle/) to the reference: every function is compared with the
outputs the reference itself produced (tests/golden/*.npz, see make_golden.py) and with the
reference-authored known answer rlglue/test/acceleration-compare.txt:5-6."""
import os

import numpy as np
import pytest

import oracle
from oracle.ars_oracle import ArsOracle
from conftest import GOLDEN, PARAM_SETS

STEP_TOL = 2e-15      # observed <= 4.5e-16 (94-99 % of the doubles bit-identical)
TRAJ_TOL = 1e-11      # observed <= 2e-13 over 1000 steps


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("pset", list(PARAM_SETS))
def test_step_and_accelerations(golden, n, pset):
    g = golden.steps
    l, m, k, h = PARAM_SETS[pset]
    key = f"n{n}_{pset}"
    p = oracle.OracleParams.make(n, l, m, k, h, g[key + "_dir"])
    nxt, rew = oracle.step_batch(p, g[key + "_state"], g[key + "_action"])
    assert np.abs(nxt - g[key + "_next"]).max() <= STEP_TOL
    assert np.abs(rew - g[key + "_reward"]).max() <= STEP_TOL
    for s, a, gd, td in zip(g[key + "_state"], g[key + "_action"], g[key + "_gdd"], g[key + "_tdd"]):
        G, T = oracle.accelerations(p, s, a)
        assert np.abs(G - gd).max() <= 1e-13 * max(1.0, np.abs(gd).max())
        assert np.abs(T - td).max() <= 1e-13 * max(1.0, np.abs(td).max())


def test_known_answers(golden):
    k = golden.kat
    p = oracle.OracleParams.make(3)
    assert np.array_equal(oracle.reset(p), k["reset"])
    nxt, r = oracle.step(p, k["reset"], [2.5, 2.5])
    assert np.abs(nxt - k["reset_step_u25"]).max() < 1e-17
    for tag in ("u0", "u25", "u5m5"):
        G, T = oracle.accelerations(p, k["kat_state"], k[f"kat_{tag}_u"])
        assert np.abs(G - k[f"kat_{tag}_gdd"]).max() < 1e-14
        assert np.abs(T - k[f"kat_{tag}_tdd"]).max() < 1e-13
    # Coulom's own program, same state (acceleration-compare.txt:6): 0.284343 printed to 6
    # digits; the y component differs only because the recorded 1.5708 != pi/2
    G, _ = oracle.accelerations(p, k["kat_state"], [0.0, 0.0])
    assert abs(G[0] - k["coulom_gdd_printed"][0]) < 1e-6
    assert abs(G[1]) < 1e-5


def cov_close(c, ref, rel):
    """|c_ij - ref_ij| <= rel * sqrt(ref_ii ref_jj): correlation-scaled comparison."""
    sd = np.sqrt(np.diag(ref))
    return bool((np.abs(c - ref) <= rel * np.outer(sd, sd)).all())


def _traj_keys(t):
    return [x[:-len("_return")] for x in t.files if x.endswith("_return")]


def test_rollouts(golden):
    t = golden.trajectories
    for key in _traj_keys(t):
        n = int(key.split("_n")[1][0])
        l, m, k, h = PARAM_SETS[key.split("_")[2]]
        p = oracle.OracleParams.make(n, l, m, k, h)
        H = t[key + "_traj"].shape[0]
        mean = t[key + "_mean"] if key + "_mean" in t.files else None
        cov = t[key + "_cov"] if key + "_cov" in t.files else None
        ret, traj = oracle.rollout(p, H, t[key + "_policy"], mean, cov)
        assert np.abs(traj - t[key + "_traj"]).max() <= TRAJ_TOL, key
        assert abs(ret - float(t[key + "_return"])) <= 1e-10, key


def test_env_module_main_scenario(golden):
    """remy_swimmer_env.py:301-316: seed-23 random state, zero policy, 1000 steps."""
    t = golden.trajectories
    ret, traj = oracle.rollout(oracle.OracleParams.make(3), 1000, np.zeros((2, 8)),
                               state0=t["main23_state0"])
    assert np.abs(traj - t["main23_traj"]).max() <= 1e-11
    assert abs(ret - 756.1082843556578) < 1e-9


ARS_CASES = ["v2_n3_N4_H50", "v2_n3_N8_H50", "v2_n3_N4_H1000", "v2_n3_N6_H200_rw",
             "v1_n3_N4_H100", "v2_n6_N4_H100", "v1_n3_N1_H1000"]
# round 2 (tests/golden/more.npz): the other chain lengths through the reference's ARS loop
MORE_ARS_CASES = ["v1_n2_N4_H400", "v2_n4_N4_H300", "v2_n5_N3_H300", "v1_n7_N2_H200", "v2_n8_N2_H200"]


@pytest.mark.parametrize("tag", ARS_CASES + MORE_ARS_CASES)
def test_ars_iterations(golden, tag):
    a = golden.more if tag in MORE_ARS_CASES else golden.ars
    n, V1, N, b, H, seed, iters = [int(x) for x in a[tag + "_cfg"]]
    l, m, k, h, alpha, nu = a[tag + "_phys"]
    o = ArsOracle(n, l, m, k, h, H, N, b, alpha, nu, bool(V1), seed)
    for it in range(iters):
        r = np.array(o.iteration())
        ref = a[tag + "_rewards"][it]
        assert np.abs(r - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (tag, it)
        # H = 50 from the symmetric reset state leaves returns of ~1e-13 and state variances
        # of ~1e-24: the reference's own update divides by those, so rounding-level
        # differences are amplified (observed 8e-9); longer rollouts agree to <= 2e-12
        ptol = 1e-7 if H <= 50 else 1e-10
        assert np.abs(o.policy - a[tag + "_policies"][it]).max() <= ptol, (tag, it)
        if not V1:
            assert np.abs(o.mean - a[tag + "_means"][it]).max() <= 1e-9
            assert cov_close(o.covariance, a[tag + "_covs"][it], 1e-6)


def test_more_whitened_rollouts(golden):
    """V2-whitened 250..400-step rollouts for n = 2, 4, 7, 8 (tests/golden/more.npz)."""
    t = golden.more
    keys = [x[:-len("_return")] for x in t.files if x.startswith("rollv2_") and x.endswith("_return")]
    assert len(keys) == 4
    for key in keys:
        n = int(key.split("_n")[1][0])
        l, m, k, h = PARAM_SETS[key.split("_")[2]]
        p = oracle.OracleParams.make(n, l, m, k, h)
        ret, traj = oracle.rollout(p, t[key + "_traj"].shape[0], t[key + "_policy"], t[key + "_mean"],
                                   t[key + "_cov"])
        assert np.abs(traj - t[key + "_traj"]).max() <= TRAJ_TOL, key
        assert abs(ret - float(t[key + "_return"])) <= 1e-11 * max(1.0, abs(float(t[key + "_return"]))), key


def test_ars_training_curve(golden):
    a = golden.ars
    o = ArsOracle(3, 1.0, 1.0, 10.0, 1e-3, 60, 3, 3, 0.0075, 0.01, False, 5)
    curve = o.training(4)
    assert np.abs(curve - a["train_v2_n3_N3_H60_curve"]).max() < 1e-15   # returns are ~1e-12
    assert np.abs(o.policy - a["train_v2_n3_N3_H60_policy"]).max() < 1e-7  # short-H amplification


def _coulom_records():
    import json
    with open(os.path.join(GOLDEN, "twin_kat.json")) as f:
        return json.load(f)["coulom"]


def coulom_consistency(accel, rec):
    """Coulom's program printed, for a state, the barycentre acceleration and the three angle
    accelerations, but not the joint torques it used.  G-double-dot does not depend on the
    torques; theta-double-dot is affine in them, so the recorded vector must be reachable:
    exists u (2 unknowns) with thdd(u) = recorded (3 equations).  Returns (|dGx|, u, residual)."""
    st = np.array(rec["state"])
    g0, t0 = accel(st, [0.0, 0.0])
    J = np.stack([accel(st, e)[1] - t0 for e in ([1.0, 0.0], [0.0, 1.0])], axis=1)
    want = np.array(rec["angle_accelerations"])
    u, *_ = np.linalg.lstsq(J, want - t0, rcond=None)
    res = np.abs(t0 + J @ u - want).max()
    return abs(g0[0] - rec["barycenter_acceleration"][0]), abs(g0[1]), u, res


def test_oracle_vs_both_states_of_coulom_program():
    """rlglue/test/acceleration-compare.txt:4-12 -- the only numbers in the reference repository
    that come from OUTSIDE its own code (Coulom's original swimmer), printed to 6 digits for
    states printed to 6 digits.  The Gym model (and so the oracle) reproduces them."""
    p = oracle.OracleParams.make(3)
    for rec, gx_tol in zip(_coulom_records(), (1e-6, 2e-5)):
        dgx, gy, u, res = coulom_consistency(lambda s, a: oracle.accelerations(p, s, a), rec)
        assert dgx < gx_tol          # state 2: 0.282799 from the 6-digit state vs 0.282794 printed
        assert gy < 1e-5             # recorded -8e-11; 1.5708 is not exactly pi / 2
        assert res < 1e-5            # the recorded angle accelerations are consistent (3 eq., 2 unknowns)
        assert np.abs(u - u[0]).max() < 1e-3 and 0.0 < u[0] < 0.1   # a small symmetric torque


# ---- round 3: the "next" rows pinned to the reference itself (tests/golden/next_rows.npz) ----
BASIC_CASES = ("basic_n3_N8_b3", "basic_n3_N4_b6", "basic_n6_N4_b2")


@pytest.mark.parametrize("tag", BASIC_CASES)
def test_basic_ars_top_b_vs_reference(golden, tag):
    """safe_ars/ars.py Basic_ARS.train (:67-98) as the reference ran it: only order[:b] enters
    sigma_R and the step (:57-63, :96) and the divisor is len(order) (:64) -- NOT b (case
    basic_n3_N4_b6 has b = 6 > N = 4: len(order) = 4).  The oracle's top-b branch against it."""
    g = golden.next_rows
    n, N, b, H, seed, iters = (int(v) for v in g[tag + "_cfg"])
    l, m, k, h, alpha, nu = g[tag + "_phys"]
    o = ArsOracle(n, l, m, k, h, H, N, b, alpha, nu, True, seed, top_b=b)
    for it in range(iters):
        r = np.array(o.iteration())
        ref = g[tag + "_returns"][it]
        assert np.abs(r - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
        assert np.mean(r) == pytest.approx(g[tag + "_curve"][it], rel=1e-9, abs=1e-15)
        assert np.abs(o.policy - g[tag + "_policies"][it]).max() <= 1e-9
    # the divisor matters: with b in its place the b > N case lands elsewhere
    if b > N:
        wrong = ArsOracle(n, l, m, k, h, H, N, b, alpha, nu, True, seed, top_b=0)
        wrong.iteration()
        assert np.abs(wrong.policy - g[tag + "_policies"][0]).max() > 1e-6


def test_estimator_objectives_vs_reference(golden):
    """ars/estimator.py:36-87 as the reference evaluated them on a Database of three of its own
    rollouts with the subset its constructor drew (capacity 4 of 3: one rollout counts twice)."""
    from oracle.estimator_oracle import objective_I, objective_J
    g = golden.next_rows
    guess = dict(zip(("m_i", "l_i", "k", "h"), g["est_guess"]))
    args = (g["est_policies"], g["est_trajectories"], g["est_subset"])
    assert len(set(g["est_subset"].tolist())) < len(g["est_subset"])
    for x, want_I, want_J in zip(g["est_x"], g["est_I"], g["est_J"]):
        got_I, got_J = objective_I(3, guess, x, *args), objective_J(3, guess, x, *args)
        if want_I == 0.0:            # true parameters: the reference asserts exactly 0.0 (:138)
            assert got_I < 1e-12 and got_J < 1e-14
        else:
            assert got_I == pytest.approx(want_I, rel=1e-11)
            assert got_J == pytest.approx(want_J, rel=1e-11)


# ---- round 4: the safe-exploration gate (safe_ars/ars.py Safe_ARS), tests/golden/safe_ars.npz ----
def test_safe_ars_oracle_vs_reference(golden):
    """oracle/safe_ars_oracle.py against what the reference's Safe_ARS.rollout / Safe_ARS.train returned: rollouts
    refused by the simulator look-ahead at steps 5 ... 32 and one that is never refused; three training iterations."""
    from oracle import safe_ars_oracle as sao
    g = golden.safe_ars
    n, H = (int(v) for v in g["cfg"])
    sim_thresh, real_thresh = (float(v) for v in g["thresholds"])
    p_real = oracle.OracleParams.make(n, *[float(v) for v in g["real_phys"]])
    p_sim = oracle.OracleParams.make(n, *[float(v) for v in g["sim_phys"]])
    cost = lambda obs: abs(obs[3])      # noqa: E731 -- |thetadot_1|, the generator's cost
    firsts = []
    for P, R_ref, st_ref in zip(g["rollout_policies"], g["rollout_returns"], g["rollout_states"]):
        R, st = sao.safe_rollout(p_real, p_sim, cost, sim_thresh, P, H)
        assert np.abs(st - st_ref).max() <= 1e-12 and abs(R - R_ref) <= 1e-13
        same = np.all(st[1:] == st[:-1], axis=1)
        firsts.append(int(np.argmax(same)) + 1 if same.any() else H)
    assert firsts == g["rollout_first_refused"].tolist()          # the gate closes at the reference's steps
    assert len(set(firsts)) >= 5 and H in firsts                  # the fixture exercises the gate
    # the reference experiment's own cost, max_i |thetadot_i| (safe_ars/experiment.py:45), n = 3 and n = 6
    cost_max = lambda x: np.max([abs(x[3 + 2 * i]) for i in range((len(x) - 2) // 2)])      # noqa: E731
    for tag in ("max3", "max6"):
        nn, HH = (int(v) for v in g[tag + "_cfg"])
        thr = [float(v) for v in g[tag + "_thresholds"]]
        pr = oracle.OracleParams.make(nn, *[float(v) for v in g["real_phys"]])
        ps = oracle.OracleParams.make(nn, *[float(v) for v in g["sim_phys"]])
        for P, R_ref, st_ref, f_ref in zip(g[tag + "_policies"], g[tag + "_returns"], g[tag + "_states"],
                                           g[tag + "_first_refused"]):
            R, st = sao.safe_rollout(pr, ps, cost_max, thr[0], P, HH)
            assert np.abs(st - st_ref).max() <= 1e-12 and abs(R - R_ref) <= 1e-13
            same = np.all(st[1:] == st[:-1], axis=1)
            assert (int(np.argmax(same)) + 1 if same.any() else HH) == int(f_ref)
    N, b, Ht, seed, iters = (int(v) for v in g["train_cfg"])
    alpha, nu = (float(v) for v in g["train_hyper"])
    pols, curve = sao.safe_train(p_real, p_sim, cost, sim_thresh, iters, N, b, alpha, nu, Ht, seed)
    assert np.abs(pols - g["train_policies"]).max() <= 1e-9
    assert np.abs(curve - g["train_curve"]).max() <= 1e-12
