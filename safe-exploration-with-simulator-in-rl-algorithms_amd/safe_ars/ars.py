"""safe_ars/ars.py of the reference, batched: `Basic_ARS` (:9-98) and `Safe_ARS` (:101-153).

The reference runs its 2N rollouts one after the other, and inside a `Safe_ARS` rollout every real step is
gated by a ONE-STEP look-ahead in a simulator: `isSafe` = `sim_env.set_state(obs)` + `sim_env.step(action)` +
`cost(sim obs) <= sim_thresh` (:111-122, called at :141) -- exactly the batched set_state + step surface of the step
kernel (SURVEY 8f-1's third consumer).  Here a batch of rollouts is

  * `Basic_ARS`: ONE launch of the rollout kernels (`sw_rollout_f64`; the action `policy @ obs` of :26 is ARS V1's);
  * `Safe_ARS` with a NATIVE cost (`AbsObs(j)`: |obs[j]|; `MaxAbsThetaDot()`: max_i |thetadot_i|, the reference
    experiment's own cost, safe_ars/experiment.py:45): ONE launch of the fused gate kernel (`sw_safe_rollouts_f64`:
    action, simulator look-ahead, cost, gate, real step -- all in registers, one rollout per lane);
  * `Safe_ARS` with any other cost callable: all rollouts in LOCK STEP, per step two launches of `sw_step_f64` over the
    whole batch (simulator parameters, real parameters, the same states and actions), the cost evaluated once on the
    batch as a [d, B] tensor -- `abs(obs[3])` works unchanged -- with a per-swimmer fallback for callables that cannot
    take tensors, and the gate as a mask.

A refused swimmer keeps its state (:150-151), which makes it propose the same action again, so it stays refused for
the rest of the horizon, as in the reference.  Same class names, constructor arguments, method names and return types
as the reference; `rollouts` (plural) is the batched form the other methods are built on.  `real_env` / `sim_env` are
`SwimmerEnv` objects (only their physical parameters are used: every rollout starts from `reset()`, :125).  The native
cost objects are ordinary callables too (lists, arrays, [d, B] tensors), so they can be handed to the reference's own
classes unchanged.  No CPU fallback for the physics: every step goes through the C ABI.
"""
import numpy as np
import torch

from .. import kernels
from .._lib import (COST_ABS_OBS, COST_MAX_ABS_THETADOT, SwParams, kernel_flags, numpy_global_uniform_pm1,
                    require_gpu)


class NativeCost(object):
    """A state cost the fused kernel evaluates itself (kind / index of include/swimmer_hip.h SW_COST_*)."""
    kind, index = None, 0


class AbsObs(NativeCost):
    """cost(obs) = |obs[index]|, obs = [Gdot_x, Gdot_y, theta_1, thetadot_1, ...]."""
    kind = COST_ABS_OBS

    def __init__(self, index):
        self.index = int(index)

    def __call__(self, obs):
        return abs(obs[self.index])


class MaxAbsThetaDot(NativeCost):
    """cost(obs) = max_i |thetadot_i| -- "maximum speed angle", safe_ars/experiment.py:43-45."""
    kind = COST_MAX_ABS_THETADOT

    def __call__(self, obs):
        n = (len(obs) - 2) // 2
        vals = [abs(obs[3 + 2 * i]) for i in range(n)]
        if isinstance(vals[0], torch.Tensor):
            return torch.stack(vals).max(dim=0).values
        return np.max(vals)


def _params(env):
    return SwParams.make(env.n, env.l_i, env.m_i, env.k, env.h, env.direction)


class Basic_ARS(object):
    """safe_ars/ars.py:9-98 -- ARS with true top-b truncation (`order[:b]`, :96) and the divisor len(order) (:64)."""

    def _gate(self, state, action):      # Basic_ARS: every step is taken
        return None

    def rollouts(self, real_env, policies, H):
        """B rollouts of H steps.  policies: [B, m, d].  Returns (returns [B], states [B, H, d]) as NumPy arrays;
        states[b, t] is the observation after step t.  ONE launch of the rollout kernels (`sw_rollout_f64`, the V1
        action policy @ obs of :26): Basic_ARS.rollout is Environment.rollout without the whitening."""
        require_gpu()
        dev = getattr(real_env, "device", torch.device("cuda:0"))
        p_real = _params(real_env)
        P = torch.as_tensor(np.ascontiguousarray(policies, dtype=np.float64), device=dev)
        B, m, d = P.shape
        assert (m, d) == (p_real.m, p_real.d), f"policies must be [B, {p_real.m}, {p_real.d}]"
        traj = torch.empty((H, d, B), dtype=torch.float64, device=dev)
        R = kernels.rollout(p_real, H, P, traj=traj)
        return R.cpu().numpy(), traj.permute(2, 0, 1).cpu().numpy()

    def _lock_step_rollouts(self, real_env, policies, H):
        """The generic form: all rollouts advance one step at a time, `_gate` decides per swimmer whether the real
        step is taken (states[b, t] is the unchanged observation where step t was refused)."""
        require_gpu()
        dev = getattr(real_env, "device", torch.device("cuda:0"))
        p_real = _params(real_env)
        P = torch.as_tensor(np.ascontiguousarray(policies, dtype=np.float64), device=dev)
        B, m, d = P.shape
        assert (m, d) == (p_real.m, p_real.d), f"policies must be [B, {p_real.m}, {p_real.d}]"
        state = kernels.reset(p_real, B, dev)
        nxt = torch.empty_like(state)
        rew = torch.empty(B, dtype=torch.float64, device=dev)
        total = torch.zeros(B, dtype=torch.float64, device=dev)
        states = torch.empty((H, d, B), dtype=torch.float64, device=dev)
        self._prepare(B, d, dev)
        for t in range(H):
            action = torch.einsum("bmd,db->mb", P, state).contiguous()     # ac = policy @ obs (:139)
            safe = self._gate(state, action)
            kernels.step(p_real, state, action, out=nxt, reward=rew)
            if safe is None:
                state, nxt = nxt, state
                total += rew
            else:
                self._after_real_step(nxt, safe)
                state = torch.where(safe, nxt, state)                       # refused: stay (:150-151)
                total += torch.where(safe, rew, torch.zeros_like(rew))
            states[t] = state
        self._finish()
        return total.cpu().numpy(), states.permute(2, 0, 1).cpu().numpy()

    def _prepare(self, B, d, dev):
        pass

    def _after_real_step(self, nxt, safe):
        pass

    def _finish(self):
        pass

    def rollout(self, real_env, policy, H, render=False):
        """One rollout, with the reference's return types: (float, list of H observation lists)."""
        R, st = self.rollouts(real_env, np.asarray(policy, dtype=np.float64)[None], H)
        return float(R[0]), st[0].tolist()

    def sort_directions(self, deltas, rewards):
        """Directions by max(r+, r-), best first (:33-44): the same ascending argsort, reversed, as the reference."""
        pairs = np.asarray(rewards, dtype=np.float64).reshape(-1, 2)[:len(deltas)]
        return np.argsort(pairs.max(axis=1)).tolist()[::-1]

    def update_policy(self, deltas, returns, order, alpha):
        """:46-65 -- sigma_R (ddof 0) over the returns of the directions in `order`, step = sum (r+ - r-) delta
        over them, divided by len(order) * sigma_R."""
        idx = np.asarray(order, dtype=np.int64)
        pairs = np.asarray(returns, dtype=np.float64).reshape(-1, 2)[idx]      # rows [r_i+, r_i-] in `order`
        step = np.tensordot(pairs[:, 0] - pairs[:, 1], np.asarray(deltas, dtype=np.float64)[idx], axes=1)
        self.policy += alpha * step / (len(idx) * np.std(pairs))

    def train(self, n_iter, real_env, N, b, alpha, nu, H):
        """safe_ars/ars.py:67-98 with the 2N rollouts of an iteration as ONE lock-step batch.  The noise comes
        from NumPy's global generator in the reference's order (N draws of 2 * rand(m, d) - 1, :84)."""
        n_obs = real_env.observation_space.shape[0]
        n_ac = real_env.action_space.shape[0]
        self.policy = np.zeros((n_ac, n_obs))
        all_returns, states = [], []
        deltas = np.empty((N, n_ac, n_obs))
        for it in range(n_iter):
            numpy_global_uniform_pm1(deltas)
            pols = np.empty((2 * N, n_ac, n_obs))
            pols[0::2] = self.policy + nu * deltas
            pols[1::2] = self.policy - nu * deltas
            returns, st = self.rollouts(real_env, pols, H)
            returns = returns.tolist()
            states.extend(st)
            order = self.sort_directions(deltas, returns)
            self.update_policy(deltas, returns, order[:b], alpha)
            all_returns.append(np.mean(returns))
            if it % 10 == 0:
                print(f"Iteration {it}/{n_iter}: return = {all_returns[-1]}")
        return np.array(all_returns), np.array(states)


class Safe_ARS(Basic_ARS):
    """safe_ars/ars.py:101-153 -- every real step gated by a one-step look-ahead in `sim_env`."""

    def __init__(self, cost, real_threshold, sim_threshold, sim_env, rollout_kernel="auto"):
        """`rollout_kernel` (not in the reference): "auto" | "lane" | "quad" -- which form of the fused gate kernel a
        native cost runs on (auto: the mirror-quad form for n = 3 up to 8192 rollouts, one rollout per lane otherwise)."""
        self.cost = cost
        self._flags = kernel_flags(rollout_kernel)
        self.real_thresh = real_threshold
        self.sim_thresh = sim_threshold
        self.sim_env = sim_env
        self.real_violations = 0      # real steps whose cost exceeded real_thresh (the reference prints each, :143-144)

    def rollouts(self, real_env, policies, H, fused=None):
        """B gated rollouts.  fused=None: ONE launch of the fused kernel when `cost` is a NativeCost, the lock-step
        path otherwise; fused=False forces the lock-step path (same results: tests/test_hip_parity.py)."""
        native = isinstance(self.cost, NativeCost)
        if fused is None:
            fused = native
        if not fused:
            return self._lock_step_rollouts(real_env, policies, H)
        if not native:
            raise TypeError("the fused gate needs a NativeCost (AbsObs / MaxAbsThetaDot); other callables take the "
                            "lock-step path (fused=False)")
        require_gpu()
        dev = getattr(real_env, "device", torch.device("cuda:0"))
        p_real, p_sim = _params(real_env), _params(self.sim_env)
        p_real.flags = self._flags
        P = torch.as_tensor(np.ascontiguousarray(policies, dtype=np.float64), device=dev)
        B, d = P.shape[0], p_real.d
        traj = torch.empty((H, d, B), dtype=torch.float64, device=dev)
        over = torch.zeros(B, dtype=torch.int32, device=dev)
        self.first_refused = torch.empty(B, dtype=torch.int32, device=dev)
        R = kernels.safe_rollouts(p_real, p_sim, H, P, self.cost.kind, self.cost.index, self.sim_thresh,
                                  self.real_thresh, traj=traj, first_refused=self.first_refused, violations=over)
        self.real_violations += int(over.sum().item())
        return R.cpu().numpy(), traj.permute(2, 0, 1).cpu().numpy()

    def _prepare(self, B, d, dev):
        self._p_sim = _params(self.sim_env)
        self._sim_next = torch.empty((d, B), dtype=torch.float64, device=dev)
        self._sim_rew = torch.empty(B, dtype=torch.float64, device=dev)
        self._over = torch.zeros((), dtype=torch.int64, device=dev)

    def _cost_batch(self, obs):
        """cost over a [d, B] batch -> [B] tensor (one tensor call; per-swimmer calls if the callable refuses)."""
        B = obs.shape[1]
        try:
            c = self.cost(obs)
            c = torch.as_tensor(c, dtype=torch.float64, device=obs.device)
            if c.shape == (B,):
                return c
        except Exception:   # noqa: BLE001 -- a cost written for Python lists only
            pass
        rows = obs.T.cpu().numpy().tolist()
        return torch.as_tensor([float(self.cost(r)) for r in rows], dtype=torch.float64, device=obs.device)

    def isSafe(self, cost, thresh, env, state, action):
        """The reference's single-swimmer form (:111-122), through the drop-in env."""
        env.set_state(state)
        obs, _, _, _ = env.step(action)
        return cost(obs) <= thresh

    def _gate(self, state, action):
        kernels.step(self._p_sim, state, action, out=self._sim_next, reward=self._sim_rew)   # set_state(obs) + step(ac), all B
        return self._cost_batch(self._sim_next) <= self.sim_thresh

    def _after_real_step(self, nxt, safe):
        # counted on the device; read back once per batch (_finish), not once per step
        self._over += ((self._cost_batch(nxt) > self.real_thresh) & safe).sum()

    def _finish(self):
        self.real_violations += int(self._over.item())
