"""bench.py's multi-rank code path rehearsed on the one-GPU box: every rank on cuda:0, the exchange staged through
the host over gloo (SWIMMER_BENCH_BACKEND=gloo SWIMMER_BENCH_SINGLE_DEVICE=1 -- knobs the driver never sets).
Functional, not a timing: the launcher spawns the ranks, the directions are sharded, every rank must end with
rank 0's policy bit for bit, rank 0 prints the one line.  The box allows six GPU processes: four ranks."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _rehearse(*flags):
    env = dict(os.environ, SWIMMER_BENCH_BACKEND="gloo", SWIMMER_BENCH_SINGLE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                          "--horizon", "100", "--no-cpu-baseline", "--no-aux", *flags],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_strong_scaling_of_the_six_segment_config_on_four_ranks():
    """configs[4] as `bench.py --gpus N --scaling strong --segments 6` runs it: 2048 directions in all."""
    rec = _rehearse("--scaling", "strong", "--segments", "6")
    assert rec["n_gpus"] == 4 and rec["scaling"] == "strong" and rec["config"]["ranks_seen"] == 4
    assert rec["config"]["segments"] == 6 and rec["config"]["directions_total"] == 2048
    assert rec["value"] > 0 and rec["steps"] == 3


def test_weak_scaling_on_four_ranks():
    rec = _rehearse()
    assert rec["n_gpus"] == 4 and rec["scaling"] == "weak" and rec["config"]["ranks_seen"] == 4
    assert rec["config"]["directions_total"] == 4 * 512
