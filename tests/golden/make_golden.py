#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  The reference's Python files are imported *by path* and executed unmodified; the
three packages they import that are not installed here (`gym`, `ray`, `cma`) are replaced
by in-memory stand-ins that carry no arithmetic:

  * gym.Env            -> empty base class (the env only subclasses it)
  * gym.spaces.Box     -> records low/high/shape (bounds are declarative in the reference,
                          remy_swimmer_env.py:36-39; never enforced)
  * ray.remote         -> identity decorator (Ray is process placement only,
                          ars/ars_agent.py:15)
  * cma                -> empty module (only used by Estimator.estimate_real_env_param)

What is written are DATA fixtures only (inputs and the reference's outputs) as .npz files
loadable with numpy.load(allow_pickle=False).  No reference source text is stored.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
"""
import importlib.util
import io
import contextlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _install_standins():
    gym = types.ModuleType("gym")

    class Env(object):
        def close(self):
            return None

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    gym.Env = Env
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces

    ray = types.ModuleType("ray")
    ray.remote = lambda obj: obj
    sys.modules["ray"] = ray
    sys.modules["cma"] = types.ModuleType("cma")

    # the env module, loaded by path and aliased where ars/environment.py:6 expects it
    path = os.path.join(REF, "envs/gym_swimmer/swimmer/remy_swimmer_env.py")
    spec = importlib.util.spec_from_file_location("remy_swimmer_env", path)
    envmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(envmod)
    for name in ("gym.envs", "gym.envs.swimmer"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["gym.envs.swimmer.remy_swimmer_env"] = envmod
    sys.path.insert(0, REF)
    return envmod


envmod = _install_standins()
SwimmerEnv = envmod.SwimmerEnv
from ars.environment import Environment  # noqa: E402  (reference module)
from ars.ars_agent import ARSAgent  # noqa: E402
from ars.parameters import EnvParam, ARSParam  # noqa: E402

PARAM_SETS = {
    # name: (l_i, m_i, k, h)
    "default": (1.0, 1.0, 10.0, 1e-3),          # remy_swimmer_env.py:16-17
    "realworld": (0.8, 1.2, 10.2, 1e-3),        # ars/plot_graph.py:14-16
    "odd": (1.3, 0.7, 4.5, 2.5e-3),             # non-default everything (pivoting differs)
}


def make_env(n, pset, direction=(1.0, 0.0)):
    l_i, m_i, k, h = PARAM_SETS[pset]
    return SwimmerEnv(direction=list(direction), n=n, l_i=l_i, m_i=m_i, k=k, h=h)


def gen_steps():
    """Single-step vectors: random states/actions (SURVEY §8d C2 distributions)."""
    out = {}
    rng = np.random.default_rng(0)
    for n in (2, 3, 4, 5, 6, 8):
        for pset in PARAM_SETS:
            B = 48
            d, m = 2 * n + 2, n - 1
            states = np.empty((B, d))
            states[:, 0:2] = rng.uniform(-0.5, 0.5, (B, 2))
            states[:, 2::2] = rng.uniform(-np.pi, np.pi, (B, n))
            states[:, 3::2] = rng.uniform(-2, 2, (B, n))
            actions = rng.uniform(-5, 5, (B, m))
            direction = (1.0, 0.0) if pset != "odd" else (0.6, -0.8)
            env = make_env(n, pset, direction)
            nxt = np.empty((B, d))
            rew = np.empty(B)
            gdd = np.empty((B, 2))
            tdd = np.empty((B, n))
            for b in range(B):
                env.set_state(states[b].tolist())
                g, t = env.compute_accelerations(actions[b], env.G_dot, env.theta, env.theta_dot)
                gdd[b], tdd[b] = g, t
                ob, r, done, info = env.step(actions[b])
                assert done is False and info == {}
                nxt[b], rew[b] = ob, r
            key = f"n{n}_{pset}"
            out[key + "_state"] = states
            out[key + "_action"] = actions
            out[key + "_next"] = nxt
            out[key + "_reward"] = rew
            out[key + "_gdd"] = gdd
            out[key + "_tdd"] = tdd
            out[key + "_dir"] = np.array(direction)
    np.savez_compressed(os.path.join(OUT, "steps.npz"), **out)
    print("steps.npz", len(out))


def gen_kat():
    """Known-answer state from rlglue/test/acceleration-compare.txt:5 evaluated by the Gym env,
    plus reset / first-step anchors."""
    out = {}
    kat = [-0.0453422, 1.33766e-11, -1.35003, -1.4868, 1.5708, -1.88179e-15, -1.79156, 1.4868]
    env = make_env(3, "default")
    out["reset"] = np.array(env.reset())
    ob, r, _, _ = env.step([2.5, 2.5])
    out["reset_step_u25"] = np.array(ob)
    out["reset_step_u25_reward"] = np.array(r)
    out["kat_state"] = np.array(kat)
    for tag, u in (("u0", [0.0, 0.0]), ("u25", [2.5, 2.5]), ("u5m5", [5.0, -5.0])):
        env.set_state(kat)
        g, t = env.compute_accelerations(np.array(u), env.G_dot, env.theta, env.theta_dot)
        out[f"kat_{tag}_u"] = np.array(u)
        out[f"kat_{tag}_gdd"] = np.array(g)
        out[f"kat_{tag}_tdd"] = np.array(t)
    # Coulom's own recorded barycentre acceleration for that state (acceleration-compare.txt:6)
    out["coulom_gdd_printed"] = np.array([0.284343, -8.38483e-11])
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **out)
    print("kat.npz")


def gen_trajectories():
    """1000-step trajectories through the reference Environment.rollout (V1 and V2 action paths)
    and the env module's own __main__ scenario (remy_swimmer_env.py:301-316)."""
    out = {}
    # (1) module __main__ scenario: seed 23 random state, zero policy, 1000 steps
    np.random.seed(23)
    env = make_env(3, "default")
    env.reset()
    env.G_dot = np.random.rand(2)
    env.theta = np.random.rand(3)
    env.theta_dot = np.random.rand(3)
    s0 = np.array(env.get_state())
    traj = np.empty((1000, 8))
    rews = np.empty(1000)
    for t in range(1000):
        ob, r, _, _ = env.step(np.zeros(2))
        traj[t], rews[t] = ob, r
    out["main23_state0"] = s0
    out["main23_traj"] = traj
    out["main23_rewards"] = rews
    out["main23_total"] = np.array(np.sum(rews))
    tot = 0.0
    for r in rews:
        tot += r
    out["main23_total_seq"] = np.array(tot)

    # (2) rollouts from reset through ars/environment.py (V1 path: covariance=None)
    for n, pset, scale, H in ((3, "default", 0.0, 1000), (3, "default", 0.1, 1000),
                              (3, "realworld", 0.1, 1000), (6, "default", 0.1, 1000),
                              (5, "realworld", 0.05, 400), (3, "default", 1.0, 1000),
                              (6, "odd", 0.3, 300)):
        l_i, m_i, k, h = PARAM_SETS[pset]
        ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
        renv = Environment(ep)
        d, m = 2 * n + 2, n - 1
        P = scale * (2 * np.random.RandomState(0).rand(m, d) - 1)
        ret, states = renv.rollout(P)
        key = f"roll_n{n}_{pset}_s{scale}_H{H}"
        out[key + "_policy"] = P
        out[key + "_return"] = np.array(ret)
        out[key + "_traj"] = np.array(states)

    # (3) V2 action path: whitening with a given mean / covariance (ars/environment.py:31-34)
    for n, pset, H in ((3, "default", 1000), (6, "realworld", 500)):
        l_i, m_i, k, h = PARAM_SETS[pset]
        ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
        renv = Environment(ep)
        d, m = 2 * n + 2, n - 1
        rs = np.random.RandomState(7)
        P = 0.05 * (2 * rs.rand(m, d) - 1)
        mean = 0.1 * rs.randn(d)
        mean[2::2] += np.pi / 2
        A = rs.randn(d, d)
        cov = 0.05 * A @ A.T + np.diag(rs.uniform(0.2, 2.0, d))
        ret, states = renv.rollout(P, covariance=cov, mean=mean)
        key = f"rollv2_n{n}_{pset}_H{H}"
        out[key + "_policy"] = P
        out[key + "_mean"] = mean
        out[key + "_cov"] = cov
        out[key + "_return"] = np.array(ret)
        out[key + "_traj"] = np.array(states)
    np.savez_compressed(os.path.join(OUT, "trajectories.npz"), **out)
    print("trajectories.npz", len(out))


def gen_ars():
    """ARS iterations through the reference ARSAgent (ars/ars_agent.py:132-185)."""
    out = {}
    cases = (
        # tag, n, pset, V1, N, b, H, alpha, nu, seed, iters
        ("v2_n3_N4_H50", 3, "default", False, 4, 4, 50, 0.0075, 0.01, 0, 3),
        ("v2_n3_N8_H50", 3, "default", False, 8, 8, 50, 0.0075, 0.01, 0, 3),
        ("v2_n3_N4_H1000", 3, "default", False, 4, 4, 1000, 0.0075, 0.01, 0, 3),
        ("v2_n3_N6_H200_rw", 3, "realworld", False, 6, 3, 200, 0.0075, 0.01, 3, 4),
        ("v1_n3_N4_H100", 3, "default", True, 4, 4, 100, 0.01, 0.02, 1, 4),
        ("v2_n6_N4_H100", 6, "default", False, 4, 2, 100, 0.0075, 0.01, 2, 3),
        ("v1_n3_N1_H1000", 3, "realworld", True, 1, 1, 1000, 0.0075, 0.01, 0, 5),
    )
    for tag, n, pset, V1, N, b, H, alpha, nu, seed, iters in cases:
        l_i, m_i, k, h = PARAM_SETS[pset]
        ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
        ap = ARSParam("Golden", V1=V1, n_iter=iters, H=H, N=N, b=b, alpha=alpha, nu=nu,
                      safe=False, threshold=0, initial_w="Zero")
        agent = ARSAgent(ep, ap, seed=seed)
        d, m = 2 * n + 2, n - 1
        rewards = np.empty((iters, 2 * N))
        policies = np.empty((iters, m, d))
        means = np.zeros((iters, d))
        covs = np.zeros((iters, d, d))
        for it in range(iters):
            with contextlib.redirect_stdout(io.StringIO()):
                r = agent.runOneIteration()
            rewards[it] = r
            policies[it] = agent.policy
            if not V1:
                means[it] = agent.mean
                covs[it] = agent.covariance
        out[tag + "_cfg"] = np.array([n, int(V1), N, b, H, seed, iters], dtype=np.int64)
        out[tag + "_phys"] = np.array([l_i, m_i, k, h, alpha, nu])
        out[tag + "_rewards"] = rewards
        out[tag + "_policies"] = policies
        out[tag + "_means"] = means
        out[tag + "_covs"] = covs
        out[tag + "_nstates"] = np.array(len(agent.saved_states), dtype=np.int64)
        # trajectory store as the reference keeps it (ars/database.py:31-34): first / last entries
        out[tag + "_db_size"] = np.array(agent.database.size, dtype=np.int64)
        out[tag + "_db_first_policy"] = np.array(agent.database.policies[0])
        out[tag + "_db_first_traj_head"] = np.array(agent.database.trajectories[0][:5])
        out[tag + "_db_last_traj_tail"] = np.array(agent.database.trajectories[-1][-5:])

    # runTraining curve (ars_agent.py:187-220): 1 warm-up + n_iter iterations, mean of 2N returns
    ep = EnvParam("LeonSwimmer-Golden", n=3, H=60, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = ARSParam("Golden", V1=False, n_iter=4, H=60, N=3, b=3, alpha=0.0075, nu=0.01,
                  safe=False, threshold=0, initial_w="Zero")
    agent = ARSAgent(ep, ap, seed=5)
    with contextlib.redirect_stdout(io.StringIO()):
        curve = agent.runTraining()
    out["train_v2_n3_N3_H60_curve"] = np.array(curve)
    out["train_v2_n3_N3_H60_policy"] = np.array(agent.policy)
    np.savez_compressed(os.path.join(OUT, "ars.npz"), **out)
    print("ars.npz", len(out))


def _ars_case(out, tag, n, pset, V1, N, b, H, alpha, nu, seed, iters):
    l_i, m_i, k, h = PARAM_SETS[pset]
    ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
    ap = ARSParam("Golden", V1=V1, n_iter=iters, H=H, N=N, b=b, alpha=alpha, nu=nu,
                  safe=False, threshold=0, initial_w="Zero")
    agent = ARSAgent(ep, ap, seed=seed)
    d, m = 2 * n + 2, n - 1
    rewards = np.empty((iters, 2 * N))
    policies = np.empty((iters, m, d))
    means = np.zeros((iters, d))
    covs = np.zeros((iters, d, d))
    for it in range(iters):
        with contextlib.redirect_stdout(io.StringIO()):
            r = agent.runOneIteration()
        rewards[it] = r
        policies[it] = agent.policy
        if not V1:
            means[it] = agent.mean
            covs[it] = agent.covariance
    out[tag + "_cfg"] = np.array([n, int(V1), N, b, H, seed, iters], dtype=np.int64)
    out[tag + "_phys"] = np.array([l_i, m_i, k, h, alpha, nu])
    out[tag + "_rewards"] = rewards
    out[tag + "_policies"] = policies
    out[tag + "_means"] = means
    out[tag + "_covs"] = covs
    out[tag + "_nstates"] = np.array(len(agent.saved_states), dtype=np.int64)


def gen_more():
    """Round 2: the chain lengths the first fixtures did not run through the reference's ARS loop
    (n = 2: lane kernel only; n = 4, 5, 7, 8: the other instantiations of the row kernel), and
    V2-whitened rollouts for n = 2, 4, 7, 8."""
    out = {}
    for case in (
        # tag, n, pset, V1, N, b, H, alpha, nu, seed, iters
        # n = 2 with V1: by symmetry the 2-segment swimmer never moves in y, so V2 would whiten by
        # the variance of pure rounding noise (2e-40 in the reference) -- not a parity target
        ("v1_n2_N4_H400", 2, "default", True, 4, 4, 400, 0.0075, 0.01, 6, 3),
        ("v2_n4_N4_H300", 4, "realworld", False, 4, 4, 300, 0.0075, 0.01, 7, 3),
        ("v2_n5_N3_H300", 5, "default", False, 3, 2, 300, 0.0075, 0.01, 8, 3),
        ("v1_n7_N2_H200", 7, "odd", True, 2, 2, 200, 0.01, 0.02, 9, 3),
        ("v2_n8_N2_H200", 8, "realworld", False, 2, 2, 200, 0.0075, 0.01, 10, 3),
    ):
        _ars_case(out, *case)
    for n, pset, H in ((2, "realworld", 400), (4, "default", 400), (7, "realworld", 300), (8, "odd", 250)):
        l_i, m_i, k, h = PARAM_SETS[pset]
        ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
        renv = Environment(ep)
        d, m = 2 * n + 2, n - 1
        rs = np.random.RandomState(20 + n)
        P = 0.05 * (2 * rs.rand(m, d) - 1)
        mean = 0.1 * rs.randn(d)
        mean[2::2] += np.pi / 2
        A = rs.randn(d, d)
        cov = 0.05 * A @ A.T + np.diag(rs.uniform(0.2, 2.0, d))
        ret, states = renv.rollout(P, covariance=cov, mean=mean)
        key = f"rollv2_n{n}_{pset}_H{H}"
        out[key + "_policy"], out[key + "_mean"], out[key + "_cov"] = P, mean, cov
        out[key + "_return"], out[key + "_traj"] = np.array(ret), np.array(states)
    np.savez_compressed(os.path.join(OUT, "more.npz"), **out)
    print("more.npz", len(out))


def gen_mirrors():
    """Public methods of the reference classes that the ARS loop does not exercise in the form a
    user may call them: Environment.select_action (V1 and V2, ars/environment.py:19-35),
    ARSAgent.update_policy with an `order` that is a SUBSET in non-sorted order
    (ars_agent.py:110-130), and the initial_w=<file>.npy warm start (ars_agent.py:74-81)."""
    out = {}
    rs = np.random.RandomState(11)
    for n in (3, 6):
        d, m = 2 * n + 2, n - 1
        ep = EnvParam("LeonSwimmer-Golden", n=n, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
        renv = Environment(ep)
        P = rs.uniform(-1, 1, (m, d))
        obs = rs.uniform(-2, 2, (5, d))
        mean = 0.3 * rs.randn(d)
        A = rs.randn(d, d)
        cov = 0.05 * A @ A.T + np.diag(rs.uniform(1e-6, 2.0, d))
        out[f"sel_n{n}_policy"], out[f"sel_n{n}_obs"] = P, obs
        out[f"sel_n{n}_mean"], out[f"sel_n{n}_cov"] = mean, cov
        out[f"sel_n{n}_v1"] = np.array([renv.select_action(P, o.tolist()) for o in obs])
        out[f"sel_n{n}_v2"] = np.array([renv.select_action(P, o.tolist(), cov, mean) for o in obs])

    # update_policy(deltas, rewards, order): order = 3 of 7 directions, not sorted
    n, N = 3, 7
    d, m = 2 * n + 2, n - 1
    ep = EnvParam("LeonSwimmer-Golden", n=n, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = ARSParam("Golden", V1=True, n_iter=1, H=10, N=N, b=3, alpha=0.02, nu=0.01,
                  safe=False, threshold=0, initial_w="Zero")
    agent = ARSAgent(ep, ap, seed=0)
    P0 = rs.uniform(-0.5, 0.5, (m, d))
    deltas = [2 * rs.rand(m, d) - 1 for _ in range(N)]
    rewards = rs.uniform(-3, 8, 2 * N).tolist()
    order = [5, 0, 3]
    agent.policy = P0.copy()
    agent.update_policy(deltas, rewards, order)
    out["upd_policy0"], out["upd_deltas"], out["upd_rewards"] = P0, np.array(deltas), np.array(rewards)
    out["upd_order"] = np.array(order, dtype=np.int64)
    out["upd_alpha_b"] = np.array([ap.alpha, ap.b])
    out["upd_policy1"] = np.array(agent.policy)
    out["upd_sorted"] = np.array(agent.sort_directions(deltas, rewards), dtype=np.int64)

    # warm start from a saved policy file (the caller writes W0 to an .npy first)
    import tempfile
    W0 = 0.05 * (2 * rs.rand(m, d) - 1)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "w0.npy")
        np.save(path, W0)
        ep = EnvParam("LeonSwimmer-Golden", n=3, H=80, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
        ap = ARSParam("Golden", V1=False, n_iter=2, H=80, N=4, b=4, alpha=0.0075, nu=0.01,
                      safe=False, threshold=0, initial_w=path)
        agent = ARSAgent(ep, ap, seed=8)
        assert np.array_equal(agent.policy, W0)
        rewards = np.empty((2, 8))
        pols = np.empty((2, m, d))
        for it in range(2):
            with contextlib.redirect_stdout(io.StringIO()):
                rewards[it] = agent.runOneIteration()
            pols[it] = agent.policy
    out["warm_w0"], out["warm_rewards"], out["warm_policies"] = W0, rewards, pols
    out["warm_mean"], out["warm_cov"] = np.array(agent.mean), np.array(agent.covariance)
    np.savez_compressed(os.path.join(OUT, "mirrors.npz"), **out)
    print("mirrors.npz", len(out))


def gen_long():
    """A reference run that actually LEARNS, so the V2 statistics move away from the reset
    state (hard part 1): n = 3, "realworld" parameters, N = 8, H = 200,
    60 iterations; checkpoints of policy / mean / diag(cov) / returns every 10 iterations."""
    out = {}
    n, N, H, iters, every = 3, 8, 200, 60, 10
    l_i, m_i, k, h = PARAM_SETS["realworld"]
    alpha, nu, seed = 0.02, 0.03, 4
    ep = EnvParam("LeonSwimmer-Golden", n=n, H=H, l_i=l_i, m_i=m_i, h=h, k=k, epsilon=0)
    ap = ARSParam("Golden", V1=False, n_iter=iters, H=H, N=N, b=N, alpha=alpha, nu=nu,
                  safe=False, threshold=0, initial_w="Zero")
    agent = ARSAgent(ep, ap, seed=seed)
    d, m = 2 * n + 2, n - 1
    curve = np.empty(iters)
    keep = list(range(every - 1, iters, every))
    rewards = np.empty((len(keep), 2 * N))
    pols = np.empty((len(keep), m, d))
    means = np.empty((len(keep), d))
    dcov = np.empty((len(keep), d))
    for it in range(iters):
        with contextlib.redirect_stdout(io.StringIO()):
            r = agent.runOneIteration()
        curve[it] = np.mean(r)
        if it in keep:
            j = keep.index(it)
            rewards[j], pols[j] = r, agent.policy
            means[j], dcov[j] = agent.mean, np.diag(agent.covariance)
    out["long_cfg"] = np.array([n, N, H, iters, every, seed], dtype=np.int64)
    out["long_phys"] = np.array([l_i, m_i, k, h, alpha, nu])
    out["long_curve"], out["long_rewards"], out["long_policies"] = curve, rewards, pols
    out["long_means"], out["long_diag_covs"] = means, dcov
    out["long_full_cov_last"] = np.array(agent.covariance)
    np.savez_compressed(os.path.join(OUT, "long.npz"), **out)
    print("long.npz: mean return first / last", curve[0], curve[-1], "max", curve.max())


def _load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_next_rows():
    """Round 3: the two "next" rows that rested on hand restatements.

    (a) safe_ars/ars.py Basic_ARS.train (:67-98): true top-b truncation (`order[:b]`, :96), sigma_R
        over the used returns (:57-60) and the divisor len(order) (:64).  `train` keeps no
        per-iteration policy, so it is run for n_iter = 1, 2, 3, ... from the same seed (the delta
        stream of a shorter run is a prefix of the longer one); the per-rollout returns of every
        iteration come from a replay through the reference's own rollout / sort_directions /
        update_policy methods on the same stream (checked here against train's all_returns).
    (b) ars/estimator.py Estimator.I (:36-62) and J (:64-87) on a Database of reference rollouts
        (ars/database.py:31-34), with the `subset` the constructor drew (:33)."""
    out = {}
    arsmod = _load_by_path("ref_safe_ars", "safe_ars/ars.py")
    cases = (
        # tag, n, pset, N, b, H, alpha, nu, seed, iters
        ("basic_n3_N8_b3", 3, "default", 8, 3, 80, 0.01, 0.02, 4, 3),
        ("basic_n3_N4_b6", 3, "realworld", 4, 6, 120, 0.0075, 0.01, 5, 2),   # b > N: len(order) = N
        ("basic_n6_N4_b2", 6, "default", 4, 2, 100, 0.01, 0.02, 6, 2),
    )
    for tag, n, pset, N, b, H, alpha, nu, seed, iters in cases:
        d, m = 2 * n + 2, n - 1
        pols = np.empty((iters, m, d))
        curves = []
        for k in range(1, iters + 1):
            agent = arsmod.Basic_ARS()
            np.random.seed(seed)
            with contextlib.redirect_stdout(io.StringIO()):
                curve, states = agent.train(k, make_env(n, pset), N, b, alpha, nu, H)
            pols[k - 1] = agent.policy
            curves.append(curve)
        assert all(np.array_equal(curves[-1][:len(c)], c) for c in curves)
        # replay with the reference's own methods for the per-rollout returns and the orders
        agent = arsmod.Basic_ARS()
        env = make_env(n, pset)
        np.random.seed(seed)
        agent.policy = np.zeros((m, d))
        rets = np.empty((iters, 2 * N))
        orders = np.full((iters, N), -1, dtype=np.int64)
        for it in range(iters):
            deltas = [2 * np.random.rand(m, d) - 1 for _ in range(N)]
            r = []
            for i in range(N):
                r.append(agent.rollout(env, agent.policy + nu * deltas[i], H)[0])
                r.append(agent.rollout(env, agent.policy - nu * deltas[i], H)[0])
            order = agent.sort_directions(deltas, r)[:b]
            agent.update_policy(deltas, r, order, alpha)
            rets[it] = r
            orders[it, :len(order)] = order
            assert np.array_equal(agent.policy, pols[it]) and np.mean(r) == curves[-1][it]
        out[tag + "_cfg"] = np.array([n, N, b, H, seed, iters], dtype=np.int64)
        l_i, m_i, k_, h = PARAM_SETS[pset]
        out[tag + "_phys"] = np.array([l_i, m_i, k_, h, alpha, nu])
        out[tag + "_policies"] = pols
        out[tag + "_curve"] = np.array(curves[-1])
        out[tag + "_returns"] = rets
        out[tag + "_orders"] = orders
        out[tag + "_last_states_shape"] = np.array(np.shape(states), dtype=np.int64)
        out[tag + "_last_state"] = np.array(states[-1][-1])

    # (b) estimator objectives
    from ars.database import Database
    from ars.estimator import Estimator
    H = 60
    l_i, m_i, k_, h = PARAM_SETS["realworld"]
    real = EnvParam("real world", n=3, H=H, l_i=l_i, m_i=m_i, h=h, k=k_, epsilon=0)
    renv = Environment(real)
    db = Database()
    rs = np.random.RandomState(31)
    for _ in range(3):
        P = 0.2 * (2 * rs.rand(2, 8) - 1)
        _, states = renv.rollout(P)          # V1 interaction (estimator.py:77)
        db.add_trajectory(states, P)
    guess = EnvParam("Simulator with estimation", n=3, H=H, m_i=1.01, l_i=1.01, h=h, k=10.01,
                     epsilon=0.01)
    np.random.seed(12)
    est = Estimator(db, guess, capacity=4)
    xs = np.array([[m_i, l_i, k_], [1.01, 1.01, 10.01], [0.9, 1.2, 9.0]])   # unknowns (m_i, l_i, k)
    out["est_policies"] = np.array(db.policies)
    out["est_trajectories"] = np.array(db.trajectories)
    out["est_subset"] = np.array(est.subset, dtype=np.int64)
    out["est_guess"] = np.array([1.01, 1.01, 10.01, h])          # m_i, l_i, k, h
    out["est_x"] = xs
    out["est_I"] = np.array([est.I(x) for x in xs])
    out["est_J"] = np.array([est.J(x) for x in xs])
    assert out["est_I"][0] == 0.0            # the reference's own check (estimator.py:138)
    np.savez_compressed(os.path.join(OUT, "next_rows.npz"), **out)
    print("next_rows.npz", len(out), "I", out["est_I"], "J", out["est_J"])


def gen_safe_ars():
    """Round 4: the third batched one-step consumer SURVEY 8f-1 cites -- safe_ars/ars.py Safe_ARS (:101-153):
    `isSafe` = sim_env.set_state(obs) + sim_env.step(action) + cost(sim obs) <= sim_thresh gates EVERY real
    step (:111-122, :141); a refused step leaves the real env where it is (:150-151), so the same action is
    proposed -- and refused -- for the rest of the horizon.  Cost = |thetadot_1| (obs[3]); the simulator is the
    default-parameter swimmer, the real env the "realworld" one.  (a) single rollouts of fixed policies, some
    refused at different steps, some never; (b) Safe_ARS.train (inherited from Basic_ARS, :67-98, with the gated
    rollout) for n_iter = 1, 2, 3 from one seed."""
    out = {}
    arsmod = _load_by_path("ref_safe_ars", "safe_ars/ars.py")

    def cost(obs):
        return abs(obs[3])

    n, H, sim_thresh, real_thresh = 3, 80, 0.30, 0.33
    out["cfg"] = np.array([n, H], dtype=np.int64)
    out["thresholds"] = np.array([sim_thresh, real_thresh])
    out["real_phys"] = np.array(PARAM_SETS["realworld"])
    out["sim_phys"] = np.array(PARAM_SETS["default"])
    rs = np.random.RandomState(77)
    pols = np.stack([sc * (2 * rs.rand(n - 1, 2 * n + 2) - 1) for sc in (0.05, 0.3, 0.6, 1.0, 1.5, 2.5, 0.8, 1.2)])
    agent = arsmod.Safe_ARS(cost, real_thresh, sim_thresh, make_env(n, "default"))
    R = np.empty(len(pols))
    states = np.empty((len(pols), H, 2 * n + 2))
    with contextlib.redirect_stdout(io.StringIO()):
        for i, P in enumerate(pols):
            R[i], st = agent.rollout(make_env(n, "realworld"), P, H)
            states[i] = np.array(st)
    out["rollout_policies"] = pols
    out["rollout_returns"] = R
    out["rollout_states"] = states
    # first refused step of each rollout (H = never): from then on the state repeats
    same = np.all(states[:, 1:] == states[:, :-1], axis=2)
    first = np.array([int(np.argmax(s)) + 1 if s.any() else H for s in same])
    out["rollout_first_refused"] = first
    print("safe_ars: first refused step per policy", first.tolist(), "returns", R)

    # (a') the same policies under the reference experiment's own cost, "maximum speed angle"
    #      (safe_ars/experiment.py:43-45: np.max of |thetadot_i|), n = 3 and a 6-segment chain
    def cost_max(x):
        return np.max([abs(x[3 + 2 * i]) for i in range((len(x) - 2) // 2)])

    for tag, nn, thr in (("max3", 3, (0.45, 0.5)), ("max6", 6, (0.8, 0.9))):
        rs2 = np.random.RandomState(78 + nn)
        pols2 = np.stack([sc * (2 * rs2.rand(nn - 1, 2 * nn + 2) - 1) for sc in (0.05, 0.4, 0.8, 1.2, 2.0, 3.0)])
        agent = arsmod.Safe_ARS(cost_max, thr[1], thr[0], make_env(nn, "default"))
        R2 = np.empty(len(pols2))
        st2 = np.empty((len(pols2), H, 2 * nn + 2))
        with contextlib.redirect_stdout(io.StringIO()):
            for i, P in enumerate(pols2):
                R2[i], st = agent.rollout(make_env(nn, "realworld"), P, H)
                st2[i] = np.array(st)
        same = np.all(st2[:, 1:] == st2[:, :-1], axis=2)
        first2 = np.array([int(np.argmax(q)) + 1 if q.any() else H for q in same])
        out[tag + "_cfg"] = np.array([nn, H], dtype=np.int64)
        out[tag + "_thresholds"] = np.array(thr)
        out[tag + "_policies"] = pols2
        out[tag + "_returns"] = R2
        out[tag + "_states"] = st2
        out[tag + "_first_refused"] = first2
        print("safe_ars", tag, "first refused step per policy", first2.tolist())

    N, b, Ht, alpha, nu, seed, iters = 6, 3, 60, 0.02, 0.9, 11, 3
    tp = np.empty((iters, n - 1, 2 * n + 2))
    curves = []
    for k in range(1, iters + 1):
        agent = arsmod.Safe_ARS(cost, real_thresh, sim_thresh, make_env(n, "default"))
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            curve, st = agent.train(k, make_env(n, "realworld"), N, b, alpha, nu, Ht)
        tp[k - 1] = agent.policy
        curves.append(curve)
    assert all(np.array_equal(curves[-1][:len(c)], c) for c in curves)
    out["train_cfg"] = np.array([N, b, Ht, seed, iters], dtype=np.int64)
    out["train_hyper"] = np.array([alpha, nu])
    out["train_policies"] = tp
    out["train_curve"] = np.array(curves[-1])
    out["train_last_states"] = np.array(st[-2 * N:])       # the last iteration's 2N rollouts [2N, Ht, d]
    np.savez_compressed(os.path.join(OUT, "safe_ars.npz"), **out)
    print("safe_ars.npz", len(out), "curve", out["train_curve"])


if __name__ == "__main__":
    if len(sys.argv) > 1:          # e.g. `make_golden.py mirrors long`: only these files
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_safe_ars()
    gen_next_rows()
    gen_mirrors()
    gen_long()
    gen_more()
    gen_kat()
    gen_steps()
    gen_trajectories()
    gen_ars()
