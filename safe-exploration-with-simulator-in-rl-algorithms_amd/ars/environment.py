"""Rollout driver: mirror of the reference `Environment` (ars/environment.py:11-60).

`rollout(policy, covariance, mean)` keeps the reference signature and return value
(total_reward, list of H post-step observations) but the whole H-step loop -- action
selection, physics, return accumulation, trajectory capture -- runs inside ONE launch of
the fused rollout kernel (sw_rollout_f64).  `rollout_batch` is the device-resident form for
many policies at once.
"""
import numpy as np
import torch

from .. import kernels
from ..envs.swimmer import SwimmerEnv
from .._lib import SwimmerHipError, STATUS_SINGULAR, kernel_flags, require_gpu


def inv_std_from_covariance(covariance, device):
    """diag(covariance) ** (-1/2) as the reference computes it (ars/environment.py:32)."""
    cov = np.asarray(covariance, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        dc = np.diag(cov) ** (-1 / 2)
    return torch.as_tensor(np.ascontiguousarray(dc), device=device)


class Environment(object):

    def __init__(self, env_param, device="cuda:0", rollout_kernel="auto"):
        self.env_param = env_param
        self.kernel_flags = kernel_flags(rollout_kernel)
        self.device = torch.device(device)
        # direction and max_u keep their defaults, as in the reference (:15-17)
        self.env = SwimmerEnv(envName=env_param.name, n=env_param.n, l_i=env_param.l_i,
                              m_i=env_param.m_i, h=env_param.h, k=env_param.k,
                              device=device)

    def select_action(self, policy, observation, covariance=None, mean=None):
        """Linear policy (V1: P s; V2: P diag(cov)^-1/2 (s - mean)), evaluated on the GPU."""
        require_gpu()
        P = torch.as_tensor(np.asarray(policy, dtype=np.float64), device=self.device)
        obs = torch.as_tensor(np.asarray(observation, dtype=np.float64), device=self.device)
        if covariance is None or mean is None:
            return (P @ obs).cpu().numpy()
        dc = inv_std_from_covariance(covariance, self.device)
        mu = torch.as_tensor(np.asarray(mean, dtype=np.float64), device=self.device)
        return ((P * dc[None, :]) @ (obs - mu)).cpu().numpy()

    def rollout_batch(self, policies, covariance=None, mean=None, H=None, want_traj=False,
                      state0=None):
        """policies [n_roll, m, d] -> (returns [n_roll] tensor, traj [H, d, n_roll] or None)."""
        require_gpu()
        p = self.env._params()
        p.flags = self.kernel_flags
        H = self.env_param.H if H is None else H
        pol = kernels._lib.dev_f64(policies, self.device)
        n_roll = pol.shape[0]
        mu = sc = None
        if covariance is not None and mean is not None:
            sc = inv_std_from_covariance(covariance, self.device)
            mu = kernels._lib.dev_f64(mean, self.device)
        traj = (torch.empty((H, p.d, n_roll), dtype=torch.float64, device=self.device)
                if want_traj else None)
        status = torch.zeros(n_roll, dtype=torch.int32, device=self.device)
        s0 = None if state0 is None else kernels._lib.dev_f64(state0, self.device)
        rets = kernels.rollout(p, H, pol, mean=mu, inv_std=sc, state0=s0, traj=traj,
                               status=status)
        if bool((status & STATUS_SINGULAR).any().item()):
            raise np.linalg.LinAlgError("Singular matrix")
        return rets, traj

    def rollout(self, policy, covariance=None, mean=None):
        """H steps following `policy`; returns (total_reward, saved_states) like the
        reference (:37-57): saved_states is a list of H observation lists (post-step)."""
        policy = np.asarray(policy, dtype=np.float64)
        if policy.shape != (self.env_param.n - 1, 2 * self.env_param.n + 2):
            raise SwimmerHipError(f"policy shape {policy.shape} does not match the env")
        rets, traj = self.rollout_batch(policy[None], covariance, mean, want_traj=True)
        total_reward = float(rets.item())
        saved_states = traj[:, :, 0].cpu().numpy().tolist()
        # leave the wrapped env in the final state, as stepping it H times would
        if saved_states:
            self.env.set_state(saved_states[-1])
        else:
            self.env.reset()
        return total_reward, saved_states

    def close(self):
        self.env.close()
