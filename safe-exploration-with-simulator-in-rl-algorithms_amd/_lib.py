"""ctypes binding of the C ABI declared in include/swimmer_hip.h.

There is no CPU fallback: if csrc/libswimmer_hip.so is missing or a GPU is not present,
every compute entry point raises.  Tensors are torch CUDA(ROCm) float64 tensors; torch is
used for device memory and streams only.
"""
import ctypes
import os

import numpy as np
import torch  # imported before the library so that both share one libamdhip64

from . import _build

ABI_VERSION = 3
MAX_SEGMENTS = 8

ERR_NAMES = {0: "SW_OK", 1: "SW_ERR_NULL", 2: "SW_ERR_SEGMENTS", 3: "SW_ERR_SIZE",
             4: "SW_ERR_PARAM", 5: "SW_ERR_LAUNCH"}
STATUS_SINGULAR = 1
STATUS_NONFINITE = 2
STATUS_RANGE = 4
FLAG_ROLLOUT_LANE = 1   # sw_params.flags: force the lane-per-rollout kernel
FLAG_ROLLOUT_QUAD = 2   # force the segment-per-lane kernels (quad: n = 3, row: n = 4..8)
FLAG_MODEL_TWIN = 4     # integrate the native RL-Glue model (SwimmerEnvironment.cpp)
COST_ABS_OBS = 0            # sw_safe_rollouts_f64: cost = |obs[index]|
COST_MAX_ABS_THETADOT = 1   # cost = max_i |thetadot_i|


class SwParams(ctypes.Structure):
    """struct sw_params (include/swimmer_hip.h)."""
    _fields_ = [("n", ctypes.c_int32), ("flags", ctypes.c_int32), ("l_i", ctypes.c_double),
                ("m_i", ctypes.c_double), ("k", ctypes.c_double), ("h", ctypes.c_double),
                ("dir_x", ctypes.c_double), ("dir_y", ctypes.c_double)]

    @classmethod
    def make(cls, n=3, l_i=1.0, m_i=1.0, k=10.0, h=1e-3, direction=(1.0, 0.0), flags=0):
        return cls(int(n), int(flags), float(l_i), float(m_i), float(k), float(h),
                   float(direction[0]), float(direction[1]))

    @property
    def d(self):
        return 2 * self.n + 2

    @property
    def m(self):
        return self.n - 1


class SwimmerHipError(RuntimeError):
    pass


def kernel_flags(name):
    """'auto' | 'lane' | 'quad' -> sw_params.flags value."""
    try:
        return {"auto": 0, "lane": FLAG_ROLLOUT_LANE, "quad": FLAG_ROLLOUT_QUAD}[name]
    except KeyError:
        raise SwimmerHipError(f"rollout kernel must be 'auto', 'lane' or 'quad', not {name!r}")


_lib = None

_PROTOTYPES = {
    # name: (restype, argtypes)
    "sw_abi_version": (ctypes.c_int, []),
    "sw_max_segments": (ctypes.c_int, []),
    "sw_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "sw_moments_blocks": (ctypes.c_int64, [ctypes.c_int64]),
    "sw_reset_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p]),
    "sw_step_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 6),
    "sw_step_residual_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5),
    "sw_step_residual_blocks": (ctypes.c_int64, [ctypes.c_int64]),
    "sw_accel_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5),
    "sw_rollout_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
                       + [ctypes.c_void_p] * 10),
    "sw_safe_rollouts_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                            ctypes.c_double] + [ctypes.c_void_p] * 6),
    "sw_ars_rollouts_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_double] + [ctypes.c_void_p] * 7),
    "sw_ars_update_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "sw_traj_moments_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sw_cov_acc_doubles": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]),
    "sw_env1_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "sw_env1_destroy": (None, [ctypes.c_void_p]),
    "sw_env1_io": (ctypes.POINTER(ctypes.c_double), [ctypes.c_void_p]),
    "sw_env1_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "sw_env1_accel": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "sw_comm_available": (ctypes.c_int, []),
    "sw_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "sw_comm_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32,
                                      ctypes.c_int32]),
    "sw_comm_destroy": (None, [ctypes.c_void_p]),
    "sw_comm_all_gather_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_void_p]),
    "sw_comm_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "sw_issue_probe": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "sw_issue_probe_grid": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "sw_mt19937_uniform_pm1": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                                              ctypes.c_int64, ctypes.c_void_p]),
    "sw_mt19937_force_isa": (ctypes.c_int, [ctypes.c_int]),
    "sw_ars_pipeline_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "sw_ars_pipeline_destroy": (None, [ctypes.c_void_p]),
    "sw_ars_pipeline_slots": (ctypes.c_int, []),
    "sw_ars_pipeline_next_slot": (ctypes.c_int, [ctypes.c_void_p]),
    "sw_ars_pipeline_host_slot_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "sw_ars_pipeline_sync_cov": (ctypes.c_int, [ctypes.c_void_p]),
    "sw_ars_pipeline_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "sw_ars_pipeline_rollout_ms": (ctypes.c_int, [ctypes.c_void_p,
                                                  ctypes.POINTER(ctypes.c_double),
                                                  ctypes.POINTER(ctypes.c_int64)]),
    "sw_ars_iteration_rollouts_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int,
                                                     ctypes.c_void_p, ctypes.c_int64,
                                                     ctypes.c_int64, ctypes.c_int64,
                                                     ctypes.c_int32, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_double] + [ctypes.c_void_p] * 8),
    "sw_ars_iteration_update_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.c_int64,
                                                   ctypes.c_void_p, ctypes.c_int32,
                                                   ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_double, ctypes.c_double,
                                                   ctypes.c_int64, ctypes.c_void_p,
                                                   ctypes.c_int64, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p]),
    "sw_ars_update_gathered_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_int32,
                                                  ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_int64, ctypes.c_void_p,
                                                  ctypes.c_int64, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)


def library_path():
    return _build.LIB_PATH


def load():
    """Load csrc/libswimmer_hip.so (never builds implicitly, never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise SwimmerHipError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sw_abi_version() != ABI_VERSION:
        raise SwimmerHipError(f"ABI version mismatch: library {lib.sw_abi_version()}, "
                              f"binding {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def require_gpu():
    if not torch.cuda.is_available():
        raise SwimmerHipError("no ROCm GPU visible: the swimmer kernels run on MI355X only "
                              "(there is no CPU fallback)")


def check(rc, what):
    if rc != 0:
        msg = load().sw_strerror(rc).decode()
        raise SwimmerHipError(f"{what}: {ERR_NAMES.get(rc, rc)} ({msg})")


def ptr(t):
    """Device pointer of a contiguous float64 / int32 CUDA tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise SwimmerHipError("expected a contiguous tensor on the GPU")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f64(x, device):
    """Host data / tensor -> contiguous float64 tensor on `device`."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=device)


def numpy_global_uniform_pm1(out):
    """Fill the float64 array `out` with 2*u-1, u drawn from NumPy's GLOBAL legacy generator
    (the stream np.random.seed / np.random.rand use), through the native MT19937 code
    (csrc/host_rng.cpp).  Bit-identical to `out[...] = 2*np.random.rand(*out.shape)-1`
    and leaves NumPy's generator advanced by exactly that many draws."""
    flat = out.reshape(-1)
    if flat.dtype != np.float64 or not flat.flags.c_contiguous or not np.shares_memory(flat, out):
        raise SwimmerHipError("numpy_global_uniform_pm1 needs a C-contiguous float64 array")
    fn = load().sw_mt19937_uniform_pm1
    dst = flat.ctypes.data_as(ctypes.c_void_p)
    try:
        # operate on NumPy's own state in place: struct { uint32_t key[624]; int pos; }
        addr = np.random.mtrand._rand._bit_generator.ctypes.state_address
        rc = fn(ctypes.c_void_p(addr),
                ctypes.cast(ctypes.c_void_p(addr + 624 * 4), ctypes.POINTER(ctypes.c_int32)),
                flat.size, dst)
    except AttributeError:   # NumPy without that attribute: copy the state out and back
        kind, key, pos, has_gauss, cached = np.random.get_state()
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        p = ctypes.c_int32(pos)
        rc = fn(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), flat.size, dst)
        np.random.set_state((kind, key, p.value, has_gauss, cached))
    check(rc, "sw_mt19937_uniform_pm1")
    return out
