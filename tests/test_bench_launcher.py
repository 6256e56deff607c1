"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run):
rendezvous variables, relay of rank 0's line, propagation of a failing rank.  CPU only: the
child processes here are small stand-in scripts that rendezvous over gloo."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_environment_wiring():
    env = bench.rank_environment(3, 8, 29555, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3"
    assert env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29555"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"          # the caller's choice is kept
    assert bench.rank_environment(0, 2, 1, base={})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_flags_of_both_scaling_modes():
    a = bench.parse(["--gpus", "8", "--steps", "5", "--warmup", "1"])
    assert (a.scaling, a.directions, a.total_directions, a.segments) == ("weak", 512, 2048, 3)
    a = bench.parse(["--gpus", "8", "--scaling", "strong", "--segments", "6"])
    assert (a.scaling, a.total_directions, a.segments) == ("strong", 2048, 6)


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")          # MASTER_ADDR / MASTER_PORT from the launcher
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"ranks_seen": dist.get_world_size(), "sum": t.item(),
                          "argv": sys.argv[1:], "local_rank": os.environ["LOCAL_RANK"]}), flush=True)
    else:
        print("noise from rank", rank, flush=True)   # must not reach the launcher's stdout
    dist.barrier()
    dist.destroy_process_group()
""")

FAIL_SCRIPT = textwrap.dedent("""
    import os, sys, time
    if os.environ["RANK"] == "1":
        sys.exit(3)
    time.sleep(120)      # a rank waiting in a collective for the one that died
""")


def _run_launcher(tmp_path, script_text, world, argv):
    script = tmp_path / "rank.py"
    script.write_text(script_text)
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.exit(bench.launch_ranks({world}, {argv!r}, script={str(script)!r}, timeout=100))
    """)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)


def test_launcher_spawns_ranks_and_relays_rank0(tmp_path):
    out = _run_launcher(tmp_path, RANK_SCRIPT, 3, ["--gpus", "3", "--steps", "2"])
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                       # exactly rank 0's line
    rec = json.loads(lines[0])
    assert rec == {"ranks_seen": 3, "sum": 6.0, "argv": ["--gpus", "3", "--steps", "2"], "local_rank": "0"}
    assert "noise from rank" in out.stderr                   # other ranks' stdout goes to stderr


def test_launcher_propagates_a_failing_rank_and_stops_the_others(tmp_path):
    t0 = time.monotonic()
    out = _run_launcher(tmp_path, FAIL_SCRIPT, 3, [])
    assert out.returncode == 3
    assert time.monotonic() - t0 < 60                        # did not wait for the sleepers
    assert "a rank failed" in out.stderr


def test_bench_without_gpus_fails_loudly_through_the_launcher():
    """The real script, two ranks, no GPU in this container: both ranks must fail, the
    launcher must return non-zero and print no result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu rehearsal test")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == ""


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
