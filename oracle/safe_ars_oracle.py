"""CPU restatement of the reference's safe-exploration rollout (safe_ars/ars.py:101-153) and the training loop it
inherits (:67-98), NumPy on the host with the physics delegated to the C oracle (swimmer_oracle.c), statement by
statement and one rollout after the other like the reference.

TEST INFRASTRUCTURE ONLY (see oracle/swimmer_oracle.c).  Parity status: pinned by tests/test_oracle_golden.py
against tests/golden/safe_ars.npz (outputs of the reference's Safe_ARS.rollout / Safe_ARS.train, run in the build
container by tests/golden/make_golden.py::gen_safe_ars).
"""
import numpy as np

from . import swimmer_oracle as so


def safe_rollout(p_real, p_sim, cost, sim_thresh, policy, H):
    """safe_ars/ars.py:124-153.  Returns (R, states [H, d])."""
    obs = so.reset(p_real)                                  # :133
    R = 0.0
    states = []
    for _ in range(H):
        ac = policy @ obs                                   # :139
        sim_obs, _ = so.step(p_sim, obs, ac)                # isSafe: set_state(obs) + step(ac), :120-121
        if cost(sim_obs) <= sim_thresh:                     # :122, :141
            new_obs, rew = so.step(p_real, obs, ac)         # :142
            R += rew
            obs = new_obs
            states.append(obs)
        else:
            states.append(states[-1] if len(states) > 0 else obs)   # :151
    return R, np.array(states)


def safe_train(p_real, p_sim, cost, sim_thresh, n_iter, N, b, alpha, nu, H, seed):
    """safe_ars/ars.py:67-98 with Safe_ARS.rollout.  Returns (policy after each iteration [n_iter, m, d], curve)."""
    n = p_real.n
    m, d = n - 1, 2 * n + 2
    np.random.seed(seed)
    policy = np.zeros((m, d))
    pols, curve = [], []
    for _ in range(n_iter):
        deltas = [2 * np.random.rand(m, d) - 1 for _ in range(N)]                 # :84
        returns = []
        for i in range(N):
            returns.append(safe_rollout(p_real, p_sim, cost, sim_thresh, policy + nu * deltas[i], H)[0])
            returns.append(safe_rollout(p_real, p_sim, cost, sim_thresh, policy - nu * deltas[i], H)[0])
        max_rewards = [max(returns[2 * i], returns[2 * i + 1]) for i in range(N)]
        order = np.argsort(max_rewards).tolist()[::-1][:b]                        # :41-42, :96
        used = []
        for i in order:
            used += [returns[2 * i], returns[2 * i + 1]]
        sigma_r = np.std(used)                                                    # :60
        grad = np.zeros((m, d))
        for i in order:
            grad += (returns[2 * i] - returns[2 * i + 1]) * deltas[i]
        grad /= (len(order) * sigma_r)                                            # :64
        policy = policy + alpha * grad
        pols.append(policy.copy())
        curve.append(np.mean(returns))
    return np.array(pols), np.array(curve)
