"""The driver's entry points: smoke() on the GPU, and the bench's JSON contract on a tiny run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_smoke_runs():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
def test_bench_emits_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--directions", "64", "--horizon", "50", "--cpu-seconds", "0.5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["value"] > 0 and "workload" in d["config"]
