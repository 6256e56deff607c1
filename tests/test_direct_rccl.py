"""The exchange step issued straight into RCCL from native code (sw_comm_*, include/swimmer_hip.h).

One GPU cannot host two RCCL ranks, so what can be checked here is: the entry points validate
their arguments without a GPU; and, on the MI355X, the ARS loop with a ONE-rank communicator and
the all-gather forced gives bit-identical policies through torch.distributed and through the
native path (the same child process bench.py's aux.collective_one_rank legs run)."""
import ctypes
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_comm_entry_points_validate_without_a_gpu():
    import swimmer_amd as sw
    lib = sw._lib.load()
    assert lib.sw_comm_available() in (0, 1)
    ident = (ctypes.c_uint8 * 128)()
    h = ctypes.c_void_p()
    assert lib.sw_comm_unique_id(None) == 1                       # SW_ERR_NULL
    assert lib.sw_comm_create(None, ident, 1, 0) == 1
    assert lib.sw_comm_create(ctypes.byref(h), None, 1, 0) == 1
    assert lib.sw_comm_create(ctypes.byref(h), ident, 0, 0) == 3  # SW_ERR_SIZE
    assert lib.sw_comm_create(ctypes.byref(h), ident, 2, 2) == 3
    assert lib.sw_comm_all_gather_f64(None, None, None, 4, None) == 1
    lib.sw_comm_destroy(None)                                     # a no-op, like free(NULL)


def _child(extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--collective-one-rank-child",
                          "--segments", "3", "--horizon", "200", "--directions", "64"] + extra,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")][-1])


@pytest.mark.gpu
def test_native_all_gather_gives_the_torch_paths_bits():
    a, b = _child([]), _child(["--direct-rccl"])
    assert a["path"] == "torch.distributed" and b["path"].startswith("sw_comm_all_gather_f64")
    assert a["policy_sha256"] == b["policy_sha256"]      # 39 iterations, same policy to the last bit
    assert a["collective_us"] > 0 and b["collective_us"] > 0
