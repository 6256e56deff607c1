"""The O(n) block-tridiagonal formulation of the accelerations (scripts/chain_formulation.py: derived and counted
before any kernel was written) against the reference's own `compute_accelerations` outputs."""
import importlib.util
import os

import numpy as np

from conftest import PARAM_SETS, ROOT


def test_block_tridiagonal_formulation_reproduces_the_reference(golden):
    spec = importlib.util.spec_from_file_location("chain_formulation", os.path.join(ROOT, "scripts", "chain_formulation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = golden.steps
    for n in (2, 3, 4, 5, 6, 8):
        for ps, (l, m, k, h) in PARAM_SETS.items():
            key = f"n{n}_{ps}"
            for st, u, gdd, tdd in zip(g[key + "_state"], g[key + "_action"], g[key + "_gdd"], g[key + "_tdd"]):
                a, b = mod.accelerations(n, l, m, k, st, u)
                assert np.abs(a - gdd).max() <= 1e-12
                assert np.abs(b - tdd).max() <= 1e-12 * max(1.0, np.abs(tdd).max())
    # the count that decided against building it: the n = 6 lane kernel would still issue > 600 instructions per step
    assert sum(mod.operation_counts(6).values()) > 600
