"""Simulator-parameter estimator: mirror of the reference `Estimator` objective functions
(ars/estimator.py:17-87, 112-121).

The reference evaluates its objective I(x) by looping over every stored transition:
`set_state(s)`, `step(select_action(policy, s))`, compare with the stored next state
(estimator.py:50-55) -- an embarrassingly parallel batch of single physics steps.  Here all
transitions of all selected trajectories go through ONE launch that steps every transition AND
compares it with its stored next state (sw_step_residual_f64: the simulated states never reach
memory); J(x) is one launch of the rollout kernel.  The CMA-ES search around them
(estimator.py:89-110) is orchestration and needs the `cma` package, which is optional.
"""
import dataclasses

import numpy as np
import torch

from .. import kernels
from .._lib import SwParams, require_gpu
from .parameters import EnvParam


class Estimator(object):

    def __init__(self, database, guess_param, capacity, unknowns=('m_i', 'l_i', 'k'),
                 device="cuda:0"):
        assert database.size > 0, "Database is empty"
        if database._device_batches and not database._trajectories:   # still on the GPU: keep it there
            assert database._device_batches[0][0].shape[0] == guess_param.H, "Rollouts are not the same"
        else:
            assert len(database.trajectories[0]) == guess_param.H, "Rollouts are not the same"
        self.guess_param = guess_param
        self.unknowns = unknowns
        self.database = database
        self.subset = np.random.randint(0, self.database.size, capacity)   # estimator.py:33
        self.iter = 0
        self.device = torch.device(device)
        self._cache = None
        self._partial = None
        self._jcache = None

    def convert_to_env_param(self, x):
        d = dataclasses.asdict(self.guess_param)
        for i in range(len(self.unknowns)):
            d[self.unknowns[i]] = x[i]
        return EnvParam(**d)

    def _params(self, x):
        ep = self.convert_to_env_param(x)
        return SwParams.make(ep.n, ep.l_i, ep.m_i, ep.k, ep.h, (1.0, 0.0))

    def _batch_from_device_store(self):
        """The same batch straight from a store that is still on the GPU (the rollout kernels'
        [H, d, R] tensors, Database.add_device_batch): no host round trip, no per-rollout loop.
        Transitions are grouped by iteration batch instead of by subset order (I(x) is a sum)."""
        db = self.database
        sizes = np.array([t.shape[2] for t, _ in db._device_batches])
        first = np.concatenate(([0], np.cumsum(sizes)))
        which = np.searchsorted(first, self.subset, side="right") - 1
        S, Nx, A, lens = [], [], [], []
        for b, (traj, pols) in enumerate(db._device_batches):
            cols = self.subset[which == b] - first[b]
            if cols.size == 0:
                continue
            idx = torch.as_tensor(cols, device=self.device)
            tr = traj.to(self.device).index_select(2, idx)                # [H, d, c]
            s = tr[:-1].permute(1, 2, 0).reshape(tr.shape[1], -1)           # [d, c * (H-1)], rollout-major
            nx = tr[1:].permute(1, 2, 0).reshape(tr.shape[1], -1)
            P = torch.as_tensor(np.ascontiguousarray(np.asarray(pols, dtype=np.float64)[cols]),
                                device=self.device)                          # [c, m, d]
            a = torch.einsum("cmd,tdc->mct", P, tr[:-1]).reshape(P.shape[1], -1)
            S.append(s)
            Nx.append(nx)
            A.append(a)
            lens += [tr.shape[0] - 1] * cols.size
        return (torch.cat(S, 1).contiguous(), torch.cat(Nx, 1).contiguous(),
                torch.cat(A, 1).contiguous(), lens)

    def _batch(self):
        """Device copies of the selected trajectories: states [d, T], next states [d, T],
        actions [m, T] (V1 action a = P s, estimator.py:52), segment lengths."""
        if self._cache is None:
            require_gpu()
            db = self.database
            if db._device_batches and not db._trajectories:
                self._cache = self._batch_from_device_store()
                return self._cache
            S, Nx, A, lens = [], [], [], []
            for k in self.subset:
                P = torch.as_tensor(np.asarray(self.database.policies[k], dtype=np.float64),
                                    device=self.device)
                tr = torch.as_tensor(np.asarray(self.database.trajectories[k], dtype=np.float64),
                                     device=self.device)
                s, nx = tr[:-1].T.contiguous(), tr[1:].T.contiguous()
                S.append(s)
                Nx.append(nx)
                A.append(P @ s)
                lens.append(s.shape[1])
            self._cache = (torch.cat(S, 1).contiguous(), torch.cat(Nx, 1).contiguous(),
                           torch.cat(A, 1).contiguous(), lens)
        return self._cache

    def I(self, x):
        """Sum over stored transitions of || sim_step(s_t, a_t) - s_{t+1} ||_2."""
        states, nexts, actions, _ = self._batch()
        # one pass: step + distance to the stored next state + per-workgroup sums (sw_step_residual_f64); the
        # simulated states are never written, only T / 256 partial sums are
        if self._partial is None or self._partial.shape[0] != kernels.step_residual_blocks(states.shape[1]):
            self._partial = torch.empty(kernels.step_residual_blocks(states.shape[1]), dtype=torch.float64,
                                        device=self.device)
        kernels.step_residual(self._params(x), states, actions, nexts, partial=self._partial)
        return float(self._partial.sum().item())

    def J(self, x):
        """Deprecated objective of the reference (estimator.py:64-87): whole-rollout distance.  ONE launch of the
        rollout kernels over every selected rollout (the reference re-runs them one after the other), then the
        per-step and per-rollout norms on the device."""
        require_gpu()
        p = self._params(x)
        if self._jcache is None:
            P = np.stack([np.asarray(self.database.policies[k], dtype=np.float64) for k in self.subset])
            real = np.stack([np.asarray(self.database.trajectories[k], dtype=np.float64) for k in self.subset])
            self._jcache = (torch.as_tensor(P, device=self.device).contiguous(),
                            torch.as_tensor(real, device=self.device).permute(1, 2, 0).contiguous())   # [H, d, K]
        P, real = self._jcache
        H, K = real.shape[0], real.shape[2]
        traj = torch.empty((H, p.d, K), dtype=torch.float64, device=self.device)
        kernels.rollout(p, H, P, traj=traj)
        per_step = torch.linalg.vector_norm(traj - real, ord=2, dim=1)             # [H, K]
        dists = torch.linalg.vector_norm(per_step, ord=2, dim=0) / H               # [K]
        return float(dists.mean().item())

    def estimate_real_env_param(self):
        """CMA-ES over I(x) (estimator.py:89-110); needs the optional `cma` package."""
        import cma
        d = dataclasses.asdict(self.guess_param)
        x0 = np.array([d[u] for u in self.unknowns], dtype=np.float64)
        es = cma.CMAEvolutionStrategy(x0, 1).optimize(self.I)
        est_x, _, _ = es.best.get()
        return self.convert_to_env_param(est_x)
