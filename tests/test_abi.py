"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/swimmer_hip.h declares, validates its arguments before touching the
device, and the product never routes through the oracle."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "swimmer_hip.h")
PKG = os.path.join(ROOT, "safe-exploration-with-simulator-in-rl-algorithms_amd")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def sw():
    import swimmer_amd
    return swimmer_amd


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("sw_step_f64", "sw_rollout_f64", "sw_ars_rollouts_f64", "sw_ars_update_f64",
                 "sw_accel_f64", "sw_reset_f64", "sw_traj_moments_f64", "sw_strerror",
                 "sw_abi_version", "sw_ars_update_gathered_f64", "sw_ars_pipeline_create",
                 "sw_ars_iteration_rollouts_f64", "sw_ars_iteration_update_f64",
                 "sw_mt19937_uniform_pm1", "sw_env1_create", "sw_env1_step", "sw_env1_accel",
                 "sw_env1_io", "sw_env1_destroy"):
        assert must in names


def test_library_exports_every_declared_symbol(sw):
    lib = ctypes.CDLL(sw._lib.library_path())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/swimmer_hip.h but not exported"
    assert set(sw._lib.EXPORTED_SYMBOLS) == set(declared_functions())
    assert sw._lib.load().sw_abi_version() == 3
    assert sw._lib.load().sw_max_segments() == 8


def test_struct_layout_matches_header(sw):
    # int32 n, int32 flags, 6 doubles
    assert ctypes.sizeof(sw.SwParams) == 8 + 6 * 8
    p = sw.SwParams.make(6, 0.8, 1.2, 10.2, 1e-3, (0.0, 1.0))
    assert (p.n, p.d, p.m) == (6, 14, 5) and p.dir_y == 1.0


def test_argument_validation_needs_no_gpu(sw):
    lib = sw._lib.load()
    bad_n = sw.SwParams.make(9)
    assert lib.sw_reset_f64(ctypes.byref(bad_n), 4, ctypes.c_void_p(8), None) == 2
    bad_l = sw.SwParams.make(3, l_i=-1.0)
    assert lib.sw_reset_f64(ctypes.byref(bad_l), 4, ctypes.c_void_p(8), None) == 4
    nan_k = sw.SwParams.make(3, k=float("nan"))
    assert lib.sw_step_f64(ctypes.byref(nan_k), 4, None, None, None, None, None, None) == 4
    ok = sw.SwParams.make(3)
    assert lib.sw_step_f64(ctypes.byref(ok), 4, None, None, None, None, None, None) == 1
    assert lib.sw_step_f64(ctypes.byref(ok), -1, None, None, None, None, None, None) == 3
    assert lib.sw_step_f64(ctypes.byref(ok), 0, None, None, None, None, None, None) == 0
    assert lib.sw_rollout_f64(None, 1, 1, *([None] * 10)) == 1
    assert lib.sw_moments_blocks(0) == 0 and lib.sw_moments_blocks(65) == 5   # one row per 16 rollouts
    assert lib.sw_strerror(2).decode().startswith("number of segments")
    # the batch-1 surface validates before it touches a handle or the device
    assert lib.sw_env1_step(None, ctypes.byref(bad_n), None) == 2
    assert lib.sw_env1_step(None, ctypes.byref(ok), None) == 1
    assert lib.sw_env1_create(None) == 1


def test_compute_fails_loudly_without_gpu(sw):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sw.SwimmerHipError):
        sw.kernels.reset(sw.SwParams.make(3), 4)
    env = sw.SwimmerEnv()
    env.reset()                      # host-side state only
    with pytest.raises(sw.SwimmerHipError):
        env.step([0.0, 0.0])


def test_missing_library_is_an_error_not_a_fallback(sw, monkeypatch):
    monkeypatch.setattr(sw._lib, "_lib", None)
    monkeypatch.setattr(sw._build, "LIB_PATH", "/nonexistent/libswimmer_hip.so")
    with pytest.raises(sw.SwimmerHipError, match="no CPU fallback"):
        sw._lib.load()


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    offenders = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\bimport\s+oracle\b|\bfrom\s+oracle\b|swimmer_oracle|libswimmer_oracle",
                             text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"import oracle", bench)]
    body = bench[bench.index("def cpu_baseline"):bench.index("def aux_step_only")]
    assert len(uses) == 1 and "import oracle" in body
