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


STRONG_N6_RANK_SCRIPT = textwrap.dedent("""
    # Stand-in for one rank of `bench.py --gpus 8 --scaling strong --segments 6` on CPU (no GPU here): the real flag
    # parser, the real shard arithmetic and packed-segment layout of ars/sharding.py, the exchange over gloo -- only
    # the kernels are replaced by rank-tagged numbers.
    import json, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    from swimmer_amd.ars.sharding import all_gather_segments, returns_from_segments, segment_len, shard_bounds
    args = bench.parse(sys.argv[1:])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == args.gpus
    dist.init_process_group("gloo")
    n, N = args.segments, (args.total_directions if args.scaling == "strong" else args.directions * world)
    d = 2 * n + 2
    lo, hi, chunk = shard_bounds(N, rank, world)
    rows_chunk = -(-2 * chunk // 16)
    L = segment_len(chunk, rows_chunk, 2 * d)
    send = torch.zeros(L, dtype=torch.float64)
    send[:2 * (hi - lo)] = torch.arange(2 * lo, 2 * hi, dtype=torch.float64)      # "returns": their own global index
    send[2 * chunk:] = float(rank + 1)                                             # "moment rows"
    gathered = torch.zeros(world * L, dtype=torch.float64)
    all_gather_segments(send, gathered, world)
    rets = returns_from_segments(gathered, N, world, chunk)
    ok = torch.equal(rets, torch.arange(2 * N, dtype=torch.float64))
    moments_ok = all(bool((gathered[r * L + 2 * chunk:(r + 1) * L] == r + 1.0).all()) for r in range(world))
    flag = torch.tensor([1.0 if (ok and moments_ok) else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"ranks_seen": world, "scaling": args.scaling, "segments": n, "directions_total": N,
                          "directions_per_rank": chunk, "segment_doubles": L, "all_ranks_ok": bool(flag.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_eight_rank_strong_scaling_of_the_six_segment_config_rehearsed_on_cpu(tmp_path):
    """`bench.py --gpus 8 --scaling strong --segments 6` (configs[4] as stated) rehearsed WITHOUT GPUs: bench's own
    launcher spawns eight ranks that parse the real flags, shard 2048 directions with the real helpers, exchange their
    packed [returns | moment rows] segments over gloo and check the gathered layout on every rank (the GPU rehearsal
    with the real kernels is tests/test_bench_rehearsal.py, four ranks on one card)."""
    out = _run_launcher(tmp_path, STRONG_N6_RANK_SCRIPT, 8, ["--gpus", "8", "--scaling", "strong", "--segments", "6"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"ranks_seen": 8, "scaling": "strong", "segments": 6, "directions_total": 2048,
                   "directions_per_rank": 256, "segment_doubles": 2 * 256 + 32 * 28, "all_ranks_ok": True}
