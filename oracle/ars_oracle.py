"""CPU restatement of the reference's ARS iteration (ars/ars_agent.py:97-185, unsafe path),
NumPy on the host with the rollouts delegated to the C oracle (swimmer_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/swimmer_oracle.c).  Parity status: pinned by
tests/test_oracle_golden.py against tests/golden/ars.npz (outputs of the reference's
ARSAgent.runOneIteration / runTraining run in the build container).
"""
import numpy as np

from . import swimmer_oracle as so


class ArsOracle(object):
    """State of one reference ARSAgent: policy, V2 mean / covariance, every saved state."""

    def __init__(self, n, l_i, m_i, k, h, H, N, b, alpha, nu, V1, seed, policy0=None, top_b=0):
        self.top_b = top_b   # > 0: safe_ars/ars.py:95-96 semantics (only the best top_b directions)
        self.p = so.OracleParams.make(n, l_i, m_i, k, h)   # direction stays (1, 0)
        self.m, self.d = n - 1, 2 * n + 2
        self.H, self.N, self.b, self.alpha, self.nu, self.V1 = H, N, b, alpha, nu, V1
        self.policy = np.zeros((self.m, self.d)) if policy0 is None else np.array(policy0)
        self.mean = None if V1 else np.zeros(self.d)               # ars_agent.py:87-90
        self.covariance = None if V1 else np.identity(self.d)
        self.saved = []                                            # :91, never cleared
        self.rng = np.random.RandomState(seed)                     # same stream as np.random.seed

    def iteration(self):
        # :137-138  N draws of rand(m, d) in sequence
        deltas = [2 * self.rng.rand(self.m, self.d) - 1 for _ in range(self.N)]
        rewards = []
        for dl in deltas:                                          # :140-172
            for pol in (self.policy + self.nu * dl, self.policy - self.nu * dl):
                ret, traj = so.rollout(self.p, self.H, pol, self.mean, self.covariance)
                rewards.append(ret)
                if not self.V1:
                    self.saved.append(traj)
        r = np.array(rewards).reshape(self.N, 2)
        order = np.argsort(r.max(axis=1))[::-1]                    # :105-108
        if self.top_b > 0:
            order = order[:self.top_b]                              # safe_ars/ars.py:96
        used = r[order].reshape(-1)
        sigma = np.std(used)                                       # :123 (ddof = 0)
        grad = np.zeros_like(self.policy)
        for i in order:                                            # :126-127
            grad += (r[i, 0] - r[i, 1]) * deltas[i]
        # :128 divides by b whatever was used; safe_ars/ars.py:64 by len(order)
        grad /= (len(order) if self.top_b > 0 else self.b) * sigma
        self.policy = self.policy + self.alpha * grad              # :130
        if not self.V1:                                            # :179-182
            states = np.concatenate(self.saved, axis=0)
            self.mean = np.mean(states, axis=0)
            self.covariance = np.cov(states.T)
        return rewards

    def training(self, n_iter):
        curve = [np.mean(self.iteration())]                        # :195
        for _ in range(n_iter):
            curve.append(np.mean(self.iteration()))
        return np.array(curve)
