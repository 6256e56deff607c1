"""Gym-surface mirror of the reference swimmer (envs/gym_swimmer/swimmer/remy_swimmer_env.py)
on top of the HIP kernels.

`SwimmerEnv` keeps the reference's constructor, attributes, method names, argument meaning
and return types (lists of Python floats, reward float, done False, info {}), so code
written against the reference env runs unchanged; every physics evaluation goes through
the C ABI (sw_step_f64 / sw_accel_f64) on the GPU.  `VecSwimmerEnv` is the batched,
device-resident form (SoA state [d, n_env]) that the MI355X is actually fed with.
"""
import math

import numpy as np
import torch

from .. import kernels
from .._lib import SwParams, STATUS_SINGULAR, require_gpu


class Box(object):
    """Declarative bounds, as gym.spaces.Box is used by the reference (:36-39): never
    enforced (actions are not clipped, remy_swimmer_env.py never reads max_u after :38)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"


def register_kwargs():
    """The kwargs the reference registers `LeonSwimmer-v0` with
    (envs/gym_swimmer/register.py:5-11; note n = 5 there, 1000-step episodes)."""
    return dict(id="LeonSwimmer-v0", max_episode_steps=1000,
                kwargs={"direction": [1.0, 0.0], "n": 5, "max_u": 5.0, "l_i": 1.0, "k": 10.0,
                        "m_i": 1.0, "h": 0.001})


def _raise_if_singular(status):
    if status is not None and bool((status & STATUS_SINGULAR).any().item()):
        raise np.linalg.LinAlgError("Singular matrix")  # numpy.linalg.solve's error (:212)


class VecSwimmerEnv(object):
    """n_env independent swimmers on one GPU, state resident in HBM as SoA [d, n_env]."""

    def __init__(self, n_env, direction=(1.0, 0.0), n=3, max_u=5.0, l_i=1.0, k=10.0, m_i=1.0,
                 h=0.001, device="cuda:0", check_singular=False):
        require_gpu()
        self.n_env = int(n_env)
        self.n, self.max_u, self.l_i, self.k, self.m_i, self.h = n, max_u, l_i, k, m_i, h
        self.direction = np.array(direction, dtype=np.float64)
        self.params = SwParams.make(n, l_i, m_i, k, h, direction)
        self.device = torch.device(device)
        self.observation_space = Box(-1000, 1000, (2 * n + 2,))
        self.action_space = Box(-max_u, max_u, (n - 1,))
        self.check_singular = check_singular
        self.state = None
        self._next = None
        self._reward = None
        self._plans = {}
        self._status = (torch.zeros(self.n_env, dtype=torch.int32, device=self.device)
                        if check_singular else None)

    def reset(self):
        self.state = kernels.reset(self.params, self.n_env, self.device, out=self.state)
        return self.state

    def set_state(self, state):
        """state: [d, n_env] (SoA) tensor/array."""
        s = torch.as_tensor(state, dtype=torch.float64, device=self.device)
        assert tuple(s.shape) == (2 * self.n + 2, self.n_env), \
            f"State has not the right dimension: {tuple(s.shape)}"
        self.state = s.contiguous().clone()
        self._plans.clear()

    def get_state(self):
        return self.state

    def step(self, action):
        """action: [m, n_env] device tensor.  Returns (state, reward, done=False, info={});
        the returned tensors are views of internal double buffers valid until the next step.
        Passing the same action tensor object every step (refilled in place) reuses a
        pre-bound launch (kernels.StepPlan)."""
        if self.state is None:
            self.reset()
        a = action if isinstance(action, torch.Tensor) else torch.as_tensor(
            np.ascontiguousarray(action, dtype=np.float64), device=self.device)
        if self._next is None:
            self._next = torch.empty_like(self.state)
            self._reward = torch.empty(self.n_env, dtype=torch.float64, device=self.device)
        # a plan pins the stream that was current when it was made: the stream is part of the key
        key = (self.state.data_ptr(), a.data_ptr(), self._next.data_ptr(),
               torch.cuda.current_stream(self.device).cuda_stream)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 8:
                self._plans.clear()
            plan = kernels.StepPlan(self.params, self.state, a, self._next, self._reward,
                                    self._status)
            self._plans[key] = plan
        plan.launch()
        self.state, self._next = self._next, self.state
        if self.check_singular:
            _raise_if_singular(self._status)
        return self.state, self._reward, False, {}

    def compute_accelerations(self, action, state=None):
        st = self.state if state is None else state
        return kernels.accelerations(self.params, st, action)


class SwimmerEnv(object):
    """Drop-in for the reference `SwimmerEnv` (same constructor and methods, one swimmer)."""
    metadata = {'render.modes': ['human']}

    def __init__(self, envName="LeonSwimmer-v0", direction=[1., 0.], n=3,
                 max_u=5., l_i=1., k=10., m_i=1., h=0.001, device="cuda:0"):
        self.direction = np.array(direction)
        self.n = n
        self.max_u = max_u
        self.l_i = l_i
        self.k = k
        self.m_i = m_i
        self.h = h
        self.envName = envName
        self.observation_space = Box(-1000, 1000, (2 * n + 2,))
        self.action_space = Box(-max_u, max_u, (n - 1,))
        self.device = torch.device(device)
        self._env1 = None   # kernels.SingleEnv, created on first use

    # parameters are plain attributes in the reference and may be reassigned between
    # calls (ars/estimator.py builds envs per candidate), so the struct is rebuilt lazily
    def _params(self):
        d = self.direction
        key = (self.n, self.l_i, self.m_i, self.k, self.h, float(d[0]), float(d[1]))
        if key != getattr(self, "_params_key", None):   # rebuilt only when an attribute was reassigned
            self._params_key = key
            self._params_struct = SwParams.make(self.n, self.l_i, self.m_i, self.k, self.h, d)
        return self._params_struct

    def _handover(self, torque, G_dot, theta, theta_dot):
        """State and action into the handle's host-mapped I/O block (kernels.SingleEnv)."""
        require_gpu()
        if self._env1 is None:
            # the handle binds to the device that is current at its creation and launches there ever after
            # (sw_env1 keeps the device id; the caller's current device is never changed by a step)
            with torch.cuda.device(self.device):
                self._env1 = kernels.SingleEnv()
        a = np.asarray(torque, dtype=np.float64).reshape(-1)
        assert a.shape[0] == self.n - 1, f"Action {torque} has not the right dimension"
        io, d = self._env1.io, 2 * self.n + 2
        io[0:2] = G_dot
        io[2:d:2] = theta
        io[3:d:2] = theta_dot
        io[kernels.SingleEnv.ACTION:kernels.SingleEnv.ACTION + self.n - 1] = a
        return io, d

    def reset(self):
        self.G_dot = np.full(2, 0.)
        self.theta = np.full(self.n, math.pi / 2)
        self.theta_dot = np.full(self.n, 0.)
        return self.get_state()

    def step(self, action):
        self.G_dot, self.theta, self.theta_dot = self.next_observation(
            action, self.G_dot, self.theta, self.theta_dot)
        ob = self.get_state()
        reward = self.get_reward()
        done = self.check_terminal()
        info = {}
        return ob, reward, done, info

    def next_observation(self, torque, G_dot, theta, theta_dot):
        """remy_swimmer_env.py:69-93 for one swimmer: ONE kernel launch and one host wait per call
        (sw_env1_step: state and action go over in a pinned, device-mapped block the kernel reads
        and writes directly -- no tensors, no copies, no stream synchronisation)."""
        io, d = self._handover(torque, G_dot, theta, theta_dot)
        status = self._env1.step(self._params())
        if status & STATUS_SINGULAR:
            raise np.linalg.LinAlgError("Singular matrix")
        s = io[kernels.SingleEnv.NEXT:kernels.SingleEnv.NEXT + d]
        return s[0:2].copy(), s[2::2].copy(), s[3::2].copy()

    def compute_accelerations(self, torque, G_dot, theta, theta_dot):
        io, _ = self._handover(torque, G_dot, theta, theta_dot)
        self._env1.accelerations(self._params())
        K = kernels.SingleEnv
        return io[K.GDD:K.GDD + 2].copy(), io[K.TDD:K.TDD + self.n].copy()

    def get_state(self):
        ob = self.G_dot.tolist()
        for i in range(self.n):
            ob += [float(self.theta[i]), float(self.theta_dot[i])]
        return ob

    def set_state(self, s):
        assert len(s) == 2 + 2 * self.n, f"State {s} has not the right dimension"
        self.reset()
        self.G_dot = np.array(s[:2], dtype=np.float64)
        for i in range(self.n):
            self.theta[i] = s[2 + 2 * i]
            self.theta_dot[i] = s[3 + 2 * i]

    def get_reward(self):
        return self.G_dot.dot(self.direction)

    def check_terminal(self):
        return False

    def render(self, mode='human'):
        return

    def close(self):
        if self._env1 is not None:
            self._env1.close()
            self._env1 = None
        return None
