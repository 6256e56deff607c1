"""Functional wrappers: torch device tensors in, one C-ABI call (include/swimmer_hip.h) each.

Shapes follow the ABI: states are SoA [d, n_env], actions [m, n_env], policies AoS
[n_roll, m, d], trajectories [H, d, n_roll].  Everything is float64 and stays on the GPU;
all calls are asynchronous on torch's current stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import SwParams, check, load, ptr, require_gpu, stream_ptr


def _f64(shape, device):
    return torch.empty(shape, dtype=torch.float64, device=device)


def _want(t, name, shape):
    if t.dtype != torch.float64 or tuple(t.shape) != tuple(shape):
        raise _lib.SwimmerHipError(f"{name}: expected float64 tensor of shape {tuple(shape)}, "
                                   f"got {t.dtype} {tuple(t.shape)}")
    return t


def reset(p: SwParams, n_env: int, device="cuda:0", out=None):
    """SwimmerEnv.reset for n_env swimmers -> state [d, n_env]."""
    require_gpu()
    state = _f64((p.d, n_env), device) if out is None else _want(out, "out", (p.d, n_env))
    check(load().sw_reset_f64(ctypes.byref(p), n_env, ptr(state), stream_ptr()), "sw_reset_f64")
    return state


def step(p: SwParams, state, action, out=None, reward=None, status=None):
    """One physics step.  Returns (next_state [d, n_env], reward [n_env])."""
    require_gpu()
    n_env = state.shape[1]
    _want(state, "state", (p.d, n_env))
    _want(action, "action", (p.m, n_env))
    out = _f64((p.d, n_env), state.device) if out is None else _want(out, "out", (p.d, n_env))
    reward = _f64((n_env,), state.device) if reward is None else _want(reward, "reward", (n_env,))
    check(load().sw_step_f64(ctypes.byref(p), n_env, ptr(state), ptr(action), ptr(out),
                             ptr(reward), ptr(status), stream_ptr()), "sw_step_f64")
    return out, reward


def step_residual_blocks(n_transitions: int) -> int:
    return int(load().sw_step_residual_blocks(int(n_transitions)))


def step_residual(p: SwParams, state, action, next_ref, partial=None):
    """Estimator.I's inner sum (ars/estimator.py:36-62) in one pass: per-workgroup sums of
    || step(state, action) - next_ref ||_2 over the transitions (SoA [d, T], [m, T], [d, T]); their sum is I(x).
    The simulated next states are never written (sw_step_residual_f64)."""
    require_gpu()
    T = state.shape[1]
    _want(state, "state", (p.d, T))
    _want(action, "action", (p.m, T))
    _want(next_ref, "next_ref", (p.d, T))
    nb = int(load().sw_step_residual_blocks(T))
    partial = _f64((nb,), state.device) if partial is None else _want(partial, "partial", (nb,))
    check(load().sw_step_residual_f64(ctypes.byref(p), T, ptr(state), ptr(action), ptr(next_ref), ptr(partial),
                                      stream_ptr()), "sw_step_residual_f64")
    return partial


class StepPlan(object):
    """A pre-bound sw_step_f64 launch: all argument conversion is done once, `launch()` is a
    single foreign call (the per-launch Python overhead of `step()` is larger than the 8192-env
    kernel itself).  Launches on the stream that was current when the plan was made."""

    def __init__(self, p: SwParams, state, action, out, reward=None, status=None):
        require_gpu()
        n_env = state.shape[1]
        _want(state, "state", (p.d, n_env))
        _want(action, "action", (p.m, n_env))
        _want(out, "out", (p.d, n_env))
        if reward is not None:
            _want(reward, "reward", (n_env,))
        self._keep = (p, state, action, out, reward, status)
        self._fn = load().sw_step_f64
        self._args = (ctypes.byref(p), n_env, ptr(state), ptr(action), ptr(out), ptr(reward),
                      ptr(status), stream_ptr())

    def launch(self):
        rc = self._fn(*self._args)
        if rc:
            check(rc, "sw_step_f64")


class DirectComm(object):
    """sw_comm: the per-iteration all-gather issued straight into RCCL from native code on the
    current stream (no ProcessGroupNCCL in between).  COLLECTIVE constructor: rank 0 draws the
    unique id, `broadcast_id(id_bytes_or_None)` must hand every rank rank 0's 128 bytes, and
    `agree(ok)` (a collective AND over the ranks; identity with one rank) makes a failure on ANY
    rank -- RCCL not resolvable, the id not drawn, the communicator not created -- an exception
    on EVERY rank instead of a hang of the others in the next collective.  The communicator binds
    to the CURRENT device: construct it under `torch.cuda.device(...)`.  EXPERIMENTAL until it has
    met more than one rank."""

    def __init__(self, world, rank, broadcast_id, agree=None):
        require_gpu()
        agree = agree or (lambda ok: ok)
        lib = load()
        why = None
        buf = (ctypes.c_uint8 * 128)()
        if not lib.sw_comm_available():
            why = "RCCL (librccl.so.1) could not be resolved at run time"
        elif rank == 0:
            rc = lib.sw_comm_unique_id(buf)
            if rc:
                why = f"sw_comm_unique_id: {_lib.ERR_NAMES.get(rc, rc)}"
        if not agree(why is None):
            raise _lib.SwimmerHipError("direct RCCL set-up failed on " +
                                       (f"this rank: {why}" if why else "another rank"))
        ident = broadcast_id(bytes(buf) if rank == 0 else None)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(ident)
        h = ctypes.c_void_p()
        rc = lib.sw_comm_create(ctypes.byref(h), buf, world, rank)
        if not agree(rc == 0):
            if rc == 0:
                lib.sw_comm_destroy(h)
            raise _lib.SwimmerHipError("sw_comm_create failed on " +
                                       (f"this rank: {_lib.ERR_NAMES.get(rc, rc)}" if rc else "another rank"))
        self._h, self._lib, self.world, self.rank = h, lib, world, rank
        self._fn = lib.sw_comm_all_gather_f64

    def all_gather(self, send, gathered):
        """gathered[r * L : (r + 1) * L] <- rank r's send (L = send.numel()), on the current stream."""
        if gathered.numel() != self.world * send.numel():
            raise _lib.SwimmerHipError("all_gather: gathered must hold world x send doubles")
        rc = self._fn(self._h, ptr(send), ptr(gathered), send.numel(), stream_ptr())
        if rc:
            raise _lib.SwimmerHipError("sw_comm_all_gather_f64: " + self._lib.sw_comm_last_error(self._h).decode())
        return gathered

    def close(self):
        if self._h is not None:
            self._lib.sw_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


class SingleEnv(object):
    """sw_env1: ONE swimmer handed over in host memory -- the batch-1 surface under
    `SwimmerEnv.step` (remy_swimmer_env.py:41-56).  The handle owns a pinned, device-mapped I/O
    block; `io` is a NumPy view of it (offsets SW_ENV1_* of include/swimmer_hip.h).  A step is
    one kernel launch and one host wait: no tensor is created, nothing is copied by a call."""
    STATE, ACTION, NEXT, REWARD, GDD, TDD, DOUBLES = 0, 18, 32, 50, 52, 54, 64

    def __init__(self):
        require_gpu()
        import numpy as np
        lib = load()
        h = ctypes.c_void_p()
        check(lib.sw_env1_create(ctypes.byref(h)), "sw_env1_create")
        self._h, self._lib = h, lib
        self.io = np.ctypeslib.as_array(lib.sw_env1_io(h), shape=(self.DOUBLES,))
        self._status = ctypes.c_int32(0)
        self._step, self._accel = lib.sw_env1_step, lib.sw_env1_accel

    def step(self, p: SwParams):
        """State and action are in `io`; returns the status bits, next state / reward in `io`."""
        rc = self._step(self._h, ctypes.byref(p), ctypes.byref(self._status))
        if rc:
            check(rc, "sw_env1_step")
        return self._status.value

    def accelerations(self, p: SwParams):
        rc = self._accel(self._h, ctypes.byref(p))
        if rc:
            check(rc, "sw_env1_accel")

    def close(self):
        if self._h is not None:
            self.io = None
            self._lib.sw_env1_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


def accelerations(p: SwParams, state, action):
    """SwimmerEnv.compute_accelerations -> (Gdd [2, n_env], thdd [n, n_env])."""
    require_gpu()
    n_env = state.shape[1]
    _want(state, "state", (p.d, n_env))
    _want(action, "action", (p.m, n_env))
    gdd = _f64((2, n_env), state.device)
    tdd = _f64((p.n, n_env), state.device)
    check(load().sw_accel_f64(ctypes.byref(p), n_env, ptr(state), ptr(action), ptr(gdd),
                              ptr(tdd), stream_ptr()), "sw_accel_f64")
    return gdd, tdd


def moments_blocks(n_roll: int) -> int:
    return int(load().sw_moments_blocks(n_roll))


def rollout(p: SwParams, H: int, policies, mean=None, inv_std=None, state0=None,
            traj=None, final_state=None, moments=None, status=None, returns=None):
    """n_roll H-step rollouts, one linear policy each.  Optional output buffers are filled
    when given (see include/swimmer_hip.h for their shapes).  Returns `returns` [n_roll]."""
    require_gpu()
    n_roll = policies.shape[0]
    _want(policies, "policies", (n_roll, p.m, p.d))
    dev = policies.device
    if (mean is None) != (inv_std is None):
        raise _lib.SwimmerHipError("mean and inv_std must be given together")
    if mean is not None:
        _want(mean, "mean", (p.d,))
        _want(inv_std, "inv_std", (p.d,))
    if state0 is not None:
        _want(state0, "state0", (p.d, n_roll))
    if traj is not None:
        _want(traj, "traj", (H, p.d, n_roll))
    if final_state is not None:
        _want(final_state, "final_state", (p.d, n_roll))
    if moments is not None:
        _want(moments, "moments", (moments_blocks(n_roll), 2 * p.d))
    returns = _f64((n_roll,), dev) if returns is None else _want(returns, "returns", (n_roll,))
    check(load().sw_rollout_f64(ctypes.byref(p), n_roll, H, ptr(policies), ptr(mean),
                                ptr(inv_std), ptr(state0), ptr(returns), ptr(traj),
                                ptr(final_state), ptr(moments), ptr(status), stream_ptr()),
          "sw_rollout_f64")
    return returns


def safe_rollouts(p_real: SwParams, p_sim: SwParams, H: int, policies, cost_kind: int, cost_index: int,
                  sim_thresh: float, real_thresh: float, traj=None, first_refused=None, violations=None,
                  status=None, returns=None):
    """safe_ars/ars.py Safe_ARS.rollout for a whole batch in ONE launch (sw_safe_rollouts_f64): n_roll rollouts of H
    steps from the reset state, every real step gated by the one-step simulator look-ahead (`isSafe`, :111-122).
    policies [n_roll, m, d]; optional outputs: traj [H, d, n_roll], first_refused / violations / status [n_roll] int32."""
    require_gpu()
    n_roll = policies.shape[0]
    _want(policies, "policies", (n_roll, p_real.m, p_real.d))
    dev = policies.device
    if traj is not None:
        _want(traj, "traj", (H, p_real.d, n_roll))
    for name, t in (("first_refused", first_refused), ("violations", violations), ("status", status)):
        if t is not None and (t.dtype != torch.int32 or tuple(t.shape) != (n_roll,)):
            raise _lib.SwimmerHipError(f"{name}: expected int32 tensor of shape ({n_roll},)")
    returns = _f64((n_roll,), dev) if returns is None else _want(returns, "returns", (n_roll,))
    check(load().sw_safe_rollouts_f64(ctypes.byref(p_real), ctypes.byref(p_sim), n_roll, H, ptr(policies),
                                      int(cost_kind), int(cost_index), float(sim_thresh), float(real_thresh),
                                      ptr(returns), ptr(traj), ptr(first_refused), ptr(violations), ptr(status),
                                      stream_ptr()), "sw_safe_rollouts_f64")
    return returns


def ars_rollouts(p: SwParams, H: int, policy, deltas, nu: float, dir_begin: int, n_dir: int,
                 mean=None, inv_std=None, returns=None, traj=None, moments=None, status=None):
    """The 2*n_dir exploration rollouts P +- nu*delta_i, i in [dir_begin, dir_begin+n_dir)."""
    require_gpu()
    _want(policy, "policy", (p.m, p.d))
    if deltas.dim() != 3 or deltas.shape[0] < dir_begin + n_dir:
        raise _lib.SwimmerHipError("deltas: need [>= dir_begin + n_dir, m, d]")
    _want(deltas, "deltas", (deltas.shape[0], p.m, p.d))
    dev = policy.device
    n_roll = 2 * n_dir
    if (mean is None) != (inv_std is None):
        raise _lib.SwimmerHipError("mean and inv_std must be given together")
    if traj is not None:
        _want(traj, "traj", (H, p.d, n_roll))
    if moments is not None:
        _want(moments, "moments", (moments_blocks(n_roll), 2 * p.d))
    returns = _f64((n_roll,), dev) if returns is None else _want(returns, "returns", (n_roll,))
    check(load().sw_ars_rollouts_f64(ctypes.byref(p), dir_begin, n_dir, H, ptr(policy),
                                     ptr(deltas), float(nu), ptr(mean), ptr(inv_std),
                                     ptr(returns), ptr(traj), ptr(moments), ptr(status),
                                     stream_ptr()), "sw_ars_rollouts_f64")
    return returns


def ars_update(p: SwParams, returns, deltas, policy, alpha: float, b: float, top_b: int = 0,
               moments=None, running=None, n_new_states: int = 0, mean=None, inv_std=None,
               sigma_out=None):
    """In-place ARS update of `policy` (+ V2 statistics when `running` is given)."""
    require_gpu()
    n_dir = returns.shape[0] // 2
    _want(returns, "returns", (2 * n_dir,))
    _want(policy, "policy", (p.m, p.d))
    _want(deltas, "deltas", (deltas.shape[0], p.m, p.d))
    if deltas.shape[0] < n_dir:
        raise _lib.SwimmerHipError("deltas: fewer directions than returns")
    n_rows = 0
    if running is not None:
        _want(running, "running", (1 + 2 * p.d,))
        _want(mean, "mean", (p.d,))
        _want(inv_std, "inv_std", (p.d,))
        n_rows = moments.shape[0]
        _want(moments, "moments", (n_rows, 2 * p.d))
    check(load().sw_ars_update_f64(ctypes.byref(p), n_dir, ptr(returns), ptr(deltas),
                                   ptr(policy), float(alpha), float(b), int(top_b),
                                   ptr(moments), n_rows, ptr(running), int(n_new_states),
                                   ptr(mean), ptr(inv_std), ptr(sigma_out), stream_ptr()),
          "sw_ars_update_f64")
    return policy


def ars_update_gathered(p: SwParams, n_dir: int, gathered, world: int, chunk: int, rows_chunk: int,
                        deltas, policy, alpha: float, b: float, top_b: int = 0, running=None,
                        n_new_states: int = 0, mean=None, inv_std=None, sigma_out=None):
    """The ARS update reading the all-gathered result segments in place: `gathered` holds
    `world` segments [2*chunk returns | rows_chunk moment rows of 2d] (sw_ars_update_gathered_f64)."""
    require_gpu()
    seg = 2 * chunk + rows_chunk * 2 * p.d
    _want(gathered, "gathered", (world * seg,))
    _want(policy, "policy", (p.m, p.d))
    _want(deltas, "deltas", (deltas.shape[0], p.m, p.d))
    if deltas.shape[0] < n_dir:
        raise _lib.SwimmerHipError("deltas: fewer directions than n_dir")
    if running is not None:
        _want(running, "running", (1 + 2 * p.d,))
        _want(mean, "mean", (p.d,))
        _want(inv_std, "inv_std", (p.d,))
    check(load().sw_ars_update_gathered_f64(
        ctypes.byref(p), int(n_dir), ptr(gathered), int(world), int(chunk), int(rows_chunk),
        ptr(deltas), ptr(policy), float(alpha), float(b), int(top_b), ptr(running),
        int(n_new_states), ptr(mean), ptr(inv_std), ptr(sigma_out), stream_ptr()),
        "sw_ars_update_gathered_f64")
    return policy


def issue_interval_ns(mode: int, device="cuda:0", trips: int = 8192):
    """Lone-wave issue interval of one instruction class on this device (mode 0: independent
    v_fma_f64, 1: v_mov_b32): HIP events around sw_issue_probe, nanoseconds per instruction."""
    require_gpu()
    scratch = torch.empty(64, dtype=torch.float64, device=device)
    fn = load().sw_issue_probe
    best = None
    for _ in range(3):
        check(fn(mode, 16, ptr(scratch), stream_ptr()), "sw_issue_probe")      # warm
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        check(fn(mode, trips, ptr(scratch), stream_ptr()), "sw_issue_probe")
        e1.record()
        check(fn(mode, 2 * trips, ptr(scratch), stream_ptr()), "sw_issue_probe")
        e2.record()
        torch.cuda.synchronize()
        # the difference of the two launches cancels the fixed launch cost
        ns = (e1.elapsed_time(e2) - e0.elapsed_time(e1)) * 1e6 / (trips * 64)
        best = ns if best is None else min(best, ns)
    return best


def issue_interval_full_chip_ns(mode: int, device="cuda:0", trips: int = 8192, workgroups: int = 256, waves: int = 4,
                                settle_ms: float = 60.0):
    """The same interval with a wave on EVERY SIMD (sw_issue_probe_grid, 256 workgroups x 4 waves): what an
    instruction costs a wave once the whole chip issues -- f64 on every SIMD lowers the clock the chip
    sustains, so the probe first loads the chip for `settle_ms` and then reports the MEDIAN of five
    difference measurements (the lone-wave probe reports the best of three)."""
    require_gpu()
    scratch = torch.empty(64, dtype=torch.float64, device=device)
    fn = load().sw_issue_probe_grid

    def go(t):
        check(fn(mode, t, workgroups, waves, ptr(scratch), stream_ptr()), "sw_issue_probe_grid")
    per_launch_ms = trips * 64 * 2.3e-6
    for _ in range(max(1, int(settle_ms / per_launch_ms))):
        go(trips)
    samples = []
    for _ in range(5):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        go(trips)
        e1.record()
        go(2 * trips)
        e2.record()
        torch.cuda.synchronize()
        samples.append((e1.elapsed_time(e2) - e0.elapsed_time(e1)) * 1e6 / (trips * 64))
    return sorted(samples)[2]


def cov_acc_doubles(p: SwParams, n_roll: int, H: int) -> int:
    """Doubles in the accumulator of a covariance pass over (n_roll, H): the 1 + d + d*d sums
    followed by the pass's scratch (include/swimmer_hip.h)."""
    n = int(load().sw_cov_acc_doubles(ctypes.byref(p), int(n_roll), int(H)))
    if n < 0:
        raise _lib.SwimmerHipError("sw_cov_acc_doubles: bad arguments")
    return n


def new_cov_acc(p: SwParams, n_roll: int, H: int, device):
    return torch.zeros(cov_acc_doubles(p, n_roll, H), dtype=torch.float64, device=device)


def traj_moments(p: SwParams, traj, acc=None):
    """acc[:1 + d + d*d] += {count, sum(s-c), sum((s-c)(s-c)^T)} over traj [H, d, n_roll];
    acc = new_cov_acc(p, n_roll, H, device) (sums + scratch), reusable for further passes of
    the same shape.  Deterministic (no floating-point atomics)."""
    require_gpu()
    H, d, n_roll = traj.shape
    _want(traj, "traj", (H, p.d, n_roll))
    if acc is None:
        acc = new_cov_acc(p, n_roll, H, traj.device)
    if acc.dtype != torch.float64 or acc.dim() != 1 or acc.numel() < cov_acc_doubles(p, n_roll, H):
        raise _lib.SwimmerHipError("acc: need a float64 vector of cov_acc_doubles(p, n_roll, H)")
    check(load().sw_traj_moments_f64(ctypes.byref(p), n_roll, H, ptr(traj), ptr(acc),
                                     stream_ptr()), "sw_traj_moments_f64")
    return acc


class ArsPipeline(object):
    """Handle of a native sw_ars_pipeline (include/swimmer_hip.h): the ring-buffered
    copy stream + progress flag + ride-along covariance schedule of one ARS iteration,
    enqueued from C."""

    def __init__(self):
        require_gpu()
        h = ctypes.c_void_p()
        check(load().sw_ars_pipeline_create(ctypes.byref(h)), "sw_ars_pipeline_create")
        self._h = h
        self.slots = int(load().sw_ars_pipeline_slots())

    def close(self):
        if getattr(self, "_h", None) is not None:
            load().sw_ars_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def next_slot(self):
        """Ring slot of the next rollouts() call (the pipeline's own call count mod slots)."""
        return int(load().sw_ars_pipeline_next_slot(self._h))

    def host_slot_wait(self, slot):
        check(load().sw_ars_pipeline_host_slot_wait(self._h, slot), "sw_ars_pipeline_host_slot_wait")

    def sync_cov(self):
        check(load().sw_ars_pipeline_sync_cov(self._h), "sw_ars_pipeline_sync_cov")

    def timing(self, every):
        """every = k > 0: HIP events around every k-th rollout launch; 0 / False: off."""
        check(load().sw_ars_pipeline_timing(self._h, int(every)), "sw_ars_pipeline_timing")

    def rollout_ms(self):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(load().sw_ars_pipeline_rollout_ms(self._h, ctypes.byref(ms), ctypes.byref(n)),
              "sw_ars_pipeline_rollout_ms")
        return ms.value, n.value

    def rollouts(self, slot, p, n_dir_total, dir_begin, n_dir, H, deltas_host, deltas_dev, policy,
                 nu, mean, inv_std, returns, traj, moments, cov_acc, status):
        """deltas_host: pinned CPU tensor [n_dir_total, m, d]; everything else on the GPU."""
        if not deltas_host.is_pinned() or not deltas_host.is_contiguous():
            raise _lib.SwimmerHipError("deltas_host must be a contiguous pinned CPU tensor")
        check(load().sw_ars_iteration_rollouts_f64(
            self._h, slot, ctypes.byref(p), n_dir_total, dir_begin, n_dir, H,
            ctypes.c_void_p(deltas_host.data_ptr()), ptr(deltas_dev), ptr(policy), float(nu),
            ptr(mean), ptr(inv_std), ptr(returns), ptr(traj), ptr(moments), ptr(cov_acc),
            ptr(status), stream_ptr()), "sw_ars_iteration_rollouts_f64")

    def update(self, slot, p, n_dir, gathered, world, chunk, rows_chunk, deltas_dev, policy,
               alpha, b, top_b, running, n_new_states, mean, inv_std, sigma_out):
        """gathered: [world * (2*chunk + rows_chunk*2d)] all-gathered segments (see
        sw_ars_update_gathered_f64); world = 1: this rank's own segment."""
        check(load().sw_ars_iteration_update_f64(
            self._h, slot, ctypes.byref(p), n_dir, ptr(gathered), int(world), int(chunk),
            int(rows_chunk), ptr(deltas_dev), ptr(policy), float(alpha), float(b), int(top_b),
            ptr(running), int(n_new_states), ptr(mean), ptr(inv_std), ptr(sigma_out),
            stream_ptr()), "sw_ars_iteration_update_f64")
