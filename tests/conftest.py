import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# physical parameter sets used by the golden fixtures (tests/golden/make_golden.py)
PARAM_SETS = {
    "default": (1.0, 1.0, 10.0, 1e-3),
    "realworld": (0.8, 1.2, 10.2, 1e-3),
    "odd": (1.3, 0.7, 4.5, 2.5e-3),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "perf_lint: code-layout checks tied to one compiler build (tests/test_loop_placement.py); "
                                       "a failure means 're-run the placement sweep', not 'the results are wrong'")
    # The suites load the in-tree libraries; build them when a fresh checkout has none
    # (hipcc cross-compiles gfx950 without a GPU; __graft_entry__.build() does the same).
    import swimmer_amd
    import oracle
    try:
        swimmer_amd._build.build_library()
    except Exception as exc:   # noqa: BLE001 -- surfaced by the tests that need the library
        print(f"conftest: could not build libswimmer_hip.so: {exc}")
    oracle.build()


def observed(name, figures):
    """Keep what a parity test OBSERVED (not only that it passed): printed, and written to
    $SWIMMER_REPORT_DIR or gpurun_out/parity_observed/<name>.json -- gpurun merges that directory back, so the
    figures of a run on the GPU box can be read (and copied into profiles/) afterwards."""
    import json
    print(f"observed {name}: {figures}")
    out = os.environ.get("SWIMMER_REPORT_DIR") or os.path.join(ROOT, "gpurun_out", "parity_observed")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(figures, f, indent=1)
    except OSError:
        pass       # a read-only tree: the printout is all there is


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G(object):
        def __getattr__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return G()
