"""CPU restatement of the reference's estimator objectives (ars/estimator.py:36-87), NumPy on
the host with the physics delegated to the C oracle (swimmer_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/swimmer_oracle.c).  Parity status: pinned by
tests/test_oracle_golden.py against tests/golden/next_rows.npz (outputs of the reference's
Estimator.I / Estimator.J on a Database of reference rollouts, run in the build container).
"""
import numpy as np

from . import swimmer_oracle as so


def _params(n, guess, unknowns, x):
    """estimator.py:112-121: the guess with the unknowns replaced by x (EnvParam fields)."""
    d = dict(guess)
    for name, v in zip(unknowns, x):
        d[name] = v
    return so.OracleParams.make(n, d["l_i"], d["m_i"], d["k"], d["h"])


def objective_I(n, guess, x, policies, trajectories, subset, unknowns=("m_i", "l_i", "k")):
    """estimator.py:36-62.  For every selected rollout: re-simulate each stored transition from
    its stored state with the V1 action policy @ s (:52, ars/environment.py:29) and sum the
    Euclidean distances to the stored next states (:60); sum over the rollouts (:62)."""
    p = _params(n, guess, unknowns, x)
    distances = []
    for k in subset:                                               # :43 (repeats count twice)
        policy = np.asarray(policies[k])
        trajectory = np.asarray(trajectories[k])
        sim_states = []
        for s in trajectory[:-1]:                                  # :51-55
            nxt, _ = so.step(p, s, policy @ s)
            sim_states.append(nxt)
        real_states = trajectory[1:]                               # :58
        distances.append(np.sum(np.linalg.norm(np.array(sim_states) - real_states, ord=2, axis=1)))
    return np.sum(distances)


def objective_J(n, guess, x, policies, trajectories, subset, unknowns=("m_i", "l_i", "k")):
    """estimator.py:64-87.  Whole rollouts from reset under the stored policy (:76-78, V1
    interaction); per rollout the 2-norm over the steps of the per-step distances divided by the
    number of steps (:81-83); mean over the rollouts (:86)."""
    p = _params(n, guess, unknowns, x)
    distances = []
    for k in subset:
        real_states = np.asarray(trajectories[k])
        _, sim_states = so.rollout(p, len(real_states), np.asarray(policies[k]))
        per_step = np.linalg.norm(sim_states - real_states, ord=2, axis=1)
        distances.append(np.linalg.norm(per_step, ord=2) / len(real_states))
    return np.mean(distances)
