"""ctypes binding of oracle/swimmer_oracle.c (the CPU restatement of the reference's Gym
swimmer step, remy_swimmer_env.py:41-251, and rollout loop, ars/environment.py:19-57).

TEST INFRASTRUCTURE ONLY -- see swimmer_oracle.c's header.  Parity status: pinned by
tests/test_oracle_golden.py against tests/golden/*.npz (reference outputs).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libswimmer_oracle.so")
_lib = None


class OracleParams(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("l_i", ctypes.c_double), ("m_i", ctypes.c_double),
                ("k", ctypes.c_double), ("h", ctypes.c_double), ("dir_x", ctypes.c_double),
                ("dir_y", ctypes.c_double)]

    @classmethod
    def make(cls, n=3, l_i=1.0, m_i=1.0, k=10.0, h=1e-3, direction=(1.0, 0.0)):
        return cls(int(n), float(l_i), float(m_i), float(k), float(h),
                   float(direction[0]), float(direction[1]))


def build(force=False):
    """Compile libswimmer_oracle.so with the Makefile next to this file."""
    srcs = [os.path.join(_HERE, f) for f in ("swimmer_oracle.c", "twin_oracle.c", "swimmer_oracle.h")]
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libswimmer_oracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        pp = ctypes.POINTER(OracleParams)
        lib.swo_accelerations.argtypes = [pp, dp, dp, dp, dp]
        lib.swo_step.argtypes = [pp, dp, dp, dp, dp]
        lib.swo_reset.argtypes = [pp, dp]
        lib.swo_reset.restype = None
        lib.swo_rollout.argtypes = [pp, ctypes.c_int, dp, dp, dp, dp, dp, dp]
        lib.swo_step_batch.argtypes = [pp, ctypes.c_long, dp, dp, dp, dp]
        lib.swo_rollout_batch.argtypes = [pp, ctypes.c_long, ctypes.c_int, dp, dp, dp, dp, dp]
        lib.swt_accelerations.argtypes = [pp, dp, dp, dp, dp]
        lib.swt_step.argtypes = [pp, dp, dp, dp, dp]
        lib.swt_system.argtypes = [pp, dp, dp, dp, dp, dp]
        lib.swt_step_batch.argtypes = [pp, ctypes.c_long, dp, dp, dp, dp]
        lib.swo_num_threads.restype = ctypes.c_int
        lib.swo_set_num_threads.argtypes = [ctypes.c_int]
        lib.swo_set_num_threads.restype = None
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _check(rc):
    if rc == 1:
        raise np.linalg.LinAlgError("Singular matrix")
    if rc != 0:
        raise ValueError(f"oracle: bad arguments (rc={rc})")


def cpu_share():
    """Host cores this process may actually use: the affinity mask capped by the cgroup
    CPU quota (a GPU box exposes 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0))
    if os.environ.get("SWIMMER_CPU_THREADS"):
        return max(1, int(os.environ["SWIMMER_CPU_THREADS"]))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:                                                                    # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def num_threads():
    """Threads the *_batch functions run on (asks the OpenMP runtime)."""
    return int(_load().swo_num_threads())


def set_num_threads(t):
    _load().swo_set_num_threads(int(t))


def accelerations(p, state, u):
    state, u = _c(state), _c(u)
    gdd = np.empty(2)
    tdd = np.empty(p.n)
    _check(_load().swo_accelerations(ctypes.byref(p), _p(state), _p(u), _p(gdd), _p(tdd)))
    return gdd, tdd


def step(p, state, u):
    state, u = _c(state), _c(u)
    nxt = np.empty(2 * p.n + 2)
    r = np.empty(1)
    _check(_load().swo_step(ctypes.byref(p), _p(state), _p(u), _p(nxt), _p(r)))
    return nxt, float(r[0])


def reset(p):
    s = np.empty(2 * p.n + 2)
    _load().swo_reset(ctypes.byref(p), _p(s))
    return s


def rollout(p, H, policy, mean=None, cov=None, state0=None, want_traj=True):
    """Environment.rollout: returns (total_reward, traj[H, d])."""
    policy, mean, state0 = _c(policy), _c(mean), _c(state0)
    cov_diag = None if cov is None else _c(np.diag(np.asarray(cov)) if np.ndim(cov) == 2 else cov)
    ret = np.empty(1)
    traj = np.empty((H, 2 * p.n + 2)) if want_traj else None
    _check(_load().swo_rollout(ctypes.byref(p), int(H), _p(policy), _p(mean), _p(cov_diag),
                               _p(state0), _p(ret), _p(traj)))
    return float(ret[0]), traj


def step_batch(p, states, actions):
    states, actions = _c(states), _c(actions)
    B = states.shape[0]
    nxt = np.empty_like(states)
    rew = np.empty(B)
    _check(_load().swo_step_batch(ctypes.byref(p), B, _p(states), _p(actions), _p(nxt), _p(rew)))
    return nxt, rew


def rollout_batch(p, H, policies, mean=None, cov=None, want_traj=False):
    policies, mean = _c(policies), _c(mean)
    cov_diag = None if cov is None else _c(np.diag(np.asarray(cov)) if np.ndim(cov) == 2 else cov)
    R = policies.shape[0]
    rets = np.empty(R)
    traj = np.empty((R, H, 2 * p.n + 2)) if want_traj else None
    _check(_load().swo_rollout_batch(ctypes.byref(p), R, int(H), _p(policies), _p(mean),
                                     _p(cov_diag), _p(rets), _p(traj)))
    return rets, traj


# ---- native twin (rlglue/environment/SwimmerEnvironment.cpp), twin_oracle.c ----
def twin_accelerations(p, state, u):
    state, u = _c(state), _c(u)
    gdd = np.empty(2)
    tdd = np.empty(p.n)
    _check(_load().swt_accelerations(ctypes.byref(p), _p(state), _p(u), _p(gdd), _p(tdd)))
    return gdd, tdd


def twin_system(p, state, u):
    """The native env's dense (5n+2) system as assembled, and its solution: (A, B, X)."""
    nn = 5 * p.n + 2
    state, u = _c(state), _c(u)
    A, B, X = np.empty((nn, nn)), np.empty(nn), np.empty(nn)
    _check(_load().swt_system(ctypes.byref(p), _p(state), _p(u), _p(A), _p(B), _p(X)))
    return A, B, X


def twin_step(p, state, u):
    state, u = _c(state), _c(u)
    nxt = np.empty(2 * p.n + 2)
    r = np.empty(1)
    _check(_load().swt_step(ctypes.byref(p), _p(state), _p(u), _p(nxt), _p(r)))
    return nxt, float(r[0])


def twin_step_batch(p, states, actions):
    states, actions = _c(states), _c(actions)
    B = states.shape[0]
    nxt = np.empty_like(states)
    rew = np.empty(B)
    _check(_load().swt_step_batch(ctypes.byref(p), B, _p(states), _p(actions), _p(nxt), _p(rew)))
    return nxt, rew
