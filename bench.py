#!/usr/bin/env python3
"""bench.py -- headline benchmark of the swimmer + ARS hot path on MI355X.

One "step" = one ARS V2 iteration on the 3-segment swimmer (BASELINE.json configs[2]):
sample N perturbations on the host (NumPy legacy RNG, like the reference), run the 2N
H-step rollouts in the fused HIP rollout kernel (trajectory capture + V2 moments on), gather
returns (RCCL all-gather when --gpus > 1), update the policy and the running state
statistics, and reduce the full state covariance from the recorded trajectories.

  --scaling weak   (default) every GPU gets --directions (512) directions: N = 512 * n_gpus
                   (4 GPUs = the 2048-direction config); with more than one rank the line also
                   carries aux.strong_2048_directions, the fixed-size problem on the same ranks
  --scaling strong BASELINE configs[3] / [4] as stated: --total-directions (2048) directions in
                   all, sharded over the ranks (add --segments 6 for configs[4])

metric = env-steps/s over the whole job = 2 * N * H * steps / wall time of the timed region.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      dominant kernel (rollout): algorithmic HBM bytes per launch / HIP-event
                duration of that launch, against the 8 TB/s HBM3E peak
  cpu_baseline  the C restatement of the reference (oracle/, OpenMP over the host cores) on
                a bounded sample of the same workload
and an "aux" object: the physics-step-only kernel (configs[1]) at 8192 envs and at a
bandwidth-bound batch, the per-GPU shards of configs[3] / [4], the saturated rollout regime.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns the N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (also fine)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MEASURED_COPY_PEAK_GBPS = 5690.0   # scripts/ubench/stream_copy on the box (profiles/r01_g_stream_copy.log)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
# Instructions one wave issues per env-step in the segment-per-lane rollout kernels: ISA count of
# the hot loop incl. trajectory stores + moments (scripts/isa_loop_stats.py, the out-of-line
# re-normalisation blocks not counted): (f64 VALU, everything else).  Round 1: 145 / 256 in all.
ROLLOUT_INSTR_PER_STEP = {3: (81, 32), 6: (201, 38)}   # n = 3: the mirror-quad kernel (round 2's quad kernel: 98 + 26)
# Measured issue interval of a lone wave (profiles/r01_ubench_issue_cost.log): an independent f64
# FMA / multiply 2.12 ns, a 32-bit move / DPP move / SALU 1.92-2.03 ns: what actually bounds the
# latency-bound rollout (boxes differ by ~2 % in clock, so the fraction can come out just above 1).
LONE_WAVE_NS_F64 = 2.12
LONE_WAVE_NS_OTHER = 1.98
TIME_EVERY = 8                  # HIP events around every 8th rollout launch of the timed region (a timed
                                # launch costs ~10 us of pipeline bubbles: 3 samples at --steps 20)
POSTPASS_LAUNCHES = 16          # + every launch of an untimed post-pass


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--directions", type=int, default=512, help="weak: ARS directions per GPU")
    ap.add_argument("--total-directions", type=int, default=2048,
                    help="strong: ARS directions of the whole job (BASELINE configs[3]/[4])")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--segments", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--collective-one-rank-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--direct-rccl", action="store_true",
                    help="issue the all-gather straight into RCCL from native code (sw_comm) "
                         "instead of through torch.distributed")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------
# Launcher: `python bench.py --gpus N` without a torch.distributed.run wrapper.  The parent has
# made no GPU call (importing torch initialises nothing); it only spawns one child per rank with
# the usual rendezvous variables, relays rank 0's line and propagates failures.
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environment(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    return env


def launch_ranks(world, argv, script=None, timeout=None):
    """Spawn `world` rank processes of this script, wait, print rank 0's stdout.  Returns the
    exit code: 0 only if every rank exited 0.  When one rank fails the others (who would wait
    for it in a collective forever) are terminated by PID."""
    script = script or os.path.abspath(__file__)
    timeout = float(os.environ.get("SWIMMER_BENCH_LAUNCH_TIMEOUT", "1500")) if timeout is None else timeout
    port = _free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(
            [sys.executable, script] + list(argv), env=rank_environment(r, world, port),
            stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0] if failed[0] > 0 else 1
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            print(f"bench launcher: ranks still running after {timeout:.0f} s", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.05)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(5)
    for ln in (out0[0].splitlines() if out0 and out0[0] else []):
        # rank 0's result line goes to stdout; whatever the communication libraries printed to
        # its stdout ("[Gloo] Rank 0 is connected ...") goes to stderr
        print(ln, file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr, flush=True)
    if rc:
        print(f"bench launcher: a rank failed (exit code {rc}); codes = "
              f"{[p.poll() for p in procs]}", file=sys.stderr)
    return rc


class stdout_to_stderr(object):
    """File-descriptor-level redirect of stdout to stderr: RCCL prints a version banner to fd 1
    when its first communicator comes up, and stdout is reserved for the ONE result line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


# ---------------------------------------------------------------------------------------
_C_TOKEN = None


def _strip_c_comments(text):
    """C / C++ source with comments removed and runs of whitespace collapsed (string and character
    literals are kept verbatim): what the compiler sees of it, so that editing a comment does not
    make the measured traffic look stale."""
    global _C_TOKEN
    import re
    if _C_TOKEN is None:
        _C_TOKEN = re.compile(r'''("(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')|(/\*.*?\*/|//[^\n]*)''', re.S)
    out = _C_TOKEN.sub(lambda m: m.group(1) if m.group(1) else " ", text)
    return " ".join(out.split())


def kernel_source_hash():
    """sha256 over the sources libswimmer_hip.so is built from (csrc/*.hip, *.h, *.cpp and
    include/*.h, by sorted name; comments and whitespace stripped).  profiles/rNN_pmc_traffic.json
    stores the hash of the tree its PMC passes ran on (scripts/pmc_traffic_json.py), so a kernel change
    without a new pass shows as `traffic_stale` instead of silently printing last round's bytes."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "safe-exploration-with-simulator-in-rl-algorithms_amd", "csrc")
    files = sorted(glob.glob(os.path.join(pkg, "*.hip")) + glob.glob(os.path.join(pkg, "*.h")) +
                   glob.glob(os.path.join(pkg, "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(_strip_c_comments(open(f, errors="replace").read()).encode())
    return h.hexdigest()


def pmc_traffic(kernel, n, directions, H):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None when
    the measured workload is not the one being benchmarked.  `stale` = the passes ran on other
    kernel sources than the ones this run was built from."""
    if (n, directions, H) != (3, 512, 1000):
        return None
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_j_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            doc = json.load(open(path))
            t = doc.get(kernel)
            if t:
                return dict(t, source="profiles/" + name,
                            stale=doc.get("source_sha256") != kernel_source_hash())
    return None


ISSUE_NS = {"f64": LONE_WAVE_NS_F64, "other": LONE_WAVE_NS_OTHER, "source": "profiles/r01_ubench_issue_cost.log"}


def calibrate_issue_intervals(sw, device):
    """Measure the two lone-wave issue intervals on THIS device (sw_issue_probe, ~10 ms)."""
    try:
        f64, other = sw.kernels.issue_interval_ns(0, device), sw.kernels.issue_interval_ns(1, device)
        # a lone wave issues one instruction per 1.7 .. 2.6 ns on every MI355X seen so far; anything
        # else means the probe did not have its SIMD to itself (several processes sharing the GPU in
        # a rehearsal): keep the recorded intervals rather than print a nonsensical bound
        if not (1.0 < f64 < 4.0 and 1.0 < other < 4.0):
            raise ValueError(f"implausible intervals {f64:.3f} / {other:.3f} ns")
        ISSUE_NS.update(f64=f64, other=other, source="measured on this device (sw_issue_probe)")
    except Exception as exc:   # noqa: BLE001 -- keep the recorded intervals
        ISSUE_NS["source"] += f" (live calibration not used: {exc})"


FULL_CHIP_NS = {}     # waves per workgroup of the probe grid -> {"f64": ns, "other": ns}


def full_chip_intervals(sw, device, waves):
    """The two issue intervals with `256 x waves` probe waves on the chip (4: one on every SIMD, what the n = 6 /
    2048-direction launch does; 2: every other SIMD, the n = 3 / 2048-direction launch), measured after the
    chip has settled under that load (kernels.issue_interval_full_chip_ns)."""
    if waves not in FULL_CHIP_NS:
        try:
            f64 = sw.kernels.issue_interval_full_chip_ns(0, device, waves=waves)
            other = sw.kernels.issue_interval_full_chip_ns(1, device, waves=waves)
            FULL_CHIP_NS[waves] = {"f64": f64, "other": other} if (1.0 < f64 < 8.0 and 1.0 < other < 8.0) else None
        except Exception:   # noqa: BLE001 -- the line then carries the idle-chip pricing only
            FULL_CHIP_NS[waves] = None
    return FULL_CHIP_NS[waves]


def issue_bound(n, H, kern_ms, full_chip=None):
    """The rollout kernel's real ceiling: every wave runs alone on its SIMD and issues ONE
    instruction per ~2 ns, so a rollout batch cannot finish faster than H x instructions per
    step x the shortest issue interval there is (floor_ms), whatever the batch size.  priced_ms
    prices the two instruction classes with their own intervals (independent 3-operand FMAs for
    the f64 class: the kernel's mix of multiplies, adds and 2-operand FMACs issues a little
    faster than that, so the measured launch can come in under priced_ms)."""
    if n not in ROLLOUT_INSTR_PER_STEP or not kern_ms:
        return None
    f64, other = ROLLOUT_INSTR_PER_STEP[n]
    fastest = min(ISSUE_NS["f64"], ISSUE_NS["other"])
    floor_ms = H * (f64 + other) * fastest * 1e-6
    priced_ms = H * (f64 * ISSUE_NS["f64"] + other * ISSUE_NS["other"]) * 1e-6
    out = {"instructions_per_step": f64 + other, "f64_instructions_per_step": f64,
           "lone_wave_ns_per_f64_instruction": ISSUE_NS["f64"],
           "lone_wave_ns_per_other_instruction": ISSUE_NS["other"],
           "intervals": ISSUE_NS["source"],
           "floor_ms": floor_ms, "frac": floor_ms / kern_ms,
           "priced_ms": priced_ms, "measured_over_priced": kern_ms / priced_ms}
    if full_chip:
        # the same mix priced with the intervals a wave sees while the WHOLE chip issues (lower sustained
        # clock): what is left above 1 here is contention inside the launch (the covariance waves sharing
        # SIMDs with the rollout waves), not the chip's power budget
        fc_ms = H * (f64 * full_chip["f64"] + other * full_chip["other"]) * 1e-6
        out.update({"full_chip_ns_per_f64_instruction": full_chip["f64"],
                    "full_chip_ns_per_other_instruction": full_chip["other"],
                    "priced_ms_full_chip": fc_ms, "measured_over_priced_full_chip": kern_ms / fc_ms})
    return out


def rollout_kernel_name(n):
    return ("rollout_oct3_kernel<true,true,true>" if n == 3
            else f"rollout_row_kernel<{n},true,true,true>" if n >= 4
            else f"rollout_kernel<{n},true,false>")


def rollout_algorithmic_bytes(n, n_dir_local, H):
    """Algorithmic HBM bytes of one rollout launch: every post-step state is materialised
    (8 d bytes per env-step, as the reference does, ars/environment.py:53) + per rollout its
    delta row (8 m d), return (8) and status (4)."""
    d = 2 * n + 2
    return 2 * n_dir_local * H * 8 * d + 2 * n_dir_local * (8 * (n - 1) * d + 12)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n, H, directions, seconds):
    """Time the oracle (C port of the reference step/rollout, OpenMP) on whole rollout
    batches of the benchmark's shape until `seconds` have elapsed."""
    import numpy as np
    import oracle
    oracle.build()
    oracle.set_num_threads(oracle.cpu_share())
    p = oracle.OracleParams.make(n)
    d, m = 2 * n + 2, n - 1
    rng = np.random.RandomState(0)
    deltas = 2 * rng.rand(directions, m, d) - 1
    pol = np.empty((2 * directions, m, d))
    pol[0::2] = 0.01 * deltas
    pol[1::2] = -0.01 * deltas
    mean = np.zeros(d)
    var = np.ones(d)
    oracle.rollout_batch(p, 10, pol[:64], mean, var)  # warm the thread pool
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        oracle.rollout_batch(p, H, pol, mean, var)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done * 2 * directions * H / dt, "unit": "env-steps/s",
            "cores": oracle.num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{done} batches of {2 * directions} rollouts x H={H} (n={n}), "
                      f"oracle/swimmer_oracle.c with OpenMP, {dt:.1f} s"}


def aux_step_only(sw, torch, n, device):
    """Physics-step-only kernel (configs[1]): 8192 envs, and a bandwidth-bound batch."""
    import numpy as np
    p = sw.SwParams.make(n)
    d, m = 2 * n + 2, n - 1
    out = {}
    rng = np.random.default_rng(0)
    for tag, B, reps in (("envs_8192", 8192, 200), ("envs_4194304", 1 << 22, 20)):
        st = torch.as_tensor(rng.uniform(-1, 1, (d, B)), device=device)
        ac = torch.as_tensor(rng.uniform(-1, 1, (m, B)), device=device)
        nxt = torch.empty_like(st)
        rew = torch.empty(B, dtype=torch.float64, device=device)
        plan = sw.kernels.StepPlan(p, st, ac, nxt, rew)   # pre-bound launch, one foreign call
        for _ in range(5):
            plan.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.launch()
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) * 1e-3 / reps
        byts = (2 * d + m + 1) * 8 * B
        out[tag] = {"us_per_launch": per * 1e6, "env_steps_per_s": B / per,
                    "algorithmic_GBps": byts / per / 1e9,
                    "hbm_frac": byts / per / 1e9 / HBM_PEAK_GBPS,
                    "frac_of_measured_copy_peak": byts / per / 1e9 / MEASURED_COPY_PEAK_GBPS}
        if B == 8192:
            # the launch-bound loop as ONE hipGraph launch: 100 steps ping-ponging two state buffers, captured
            # from the same pre-bound launches (the library launches on the capturing stream like any other)
            try:
                back = sw.kernels.StepPlan(p, nxt, ac, st, rew)
                side = torch.cuda.Stream(device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fwd_s = sw.kernels.StepPlan(p, st, ac, nxt, rew)     # plans bind the stream current at creation
                    back_s = sw.kernels.StepPlan(p, nxt, ac, st, rew)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        for _ in range(50):
                            fwd_s.launch()
                            back_s.launch()
                    for _ in range(3):
                        graph.replay()
                    side.synchronize()
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record()
                    for _ in range(10):
                        graph.replay()
                    g1.record()
                    side.synchronize()
                gper = g0.elapsed_time(g1) * 1e-3 / 1000
                out[tag].update(graph_us_per_step=gper * 1e6, graph_env_steps_per_s=B / gper,
                                graph_note="100 step launches captured in one hipGraph, 10 replays")
                del graph, back, fwd_s, back_s
            except Exception as exc:   # noqa: BLE001 -- reported, the eager number stands
                out[tag]["graph_error"] = f"{type(exc).__name__}: {exc}"
    return out


def aux_single_env(sw, torch, device, steps=3000):
    """The batch-1 drop-in surfaces north_star keeps: us per SwimmerEnv.step (Gym surface,
    remy_swimmer_env.py:41-56), per bare sw_env1_step (what is under it), and per RL-Glue env_step
    (SwimmerEnvironment.cpp:53-68) -- each ONE kernel launch + one host wait, state handed over in
    a pinned, device-mapped block.  Beside them the reference's own CPU step, measured where the
    reference can run (it cannot travel to the GPU box)."""
    import ctypes
    import numpy as np
    out = {"reference_cpu_us_per_step": 111.0,
           "reference_cpu_source": "SURVEY 8(d): reference Environment.rollout, 8.97e3 env-steps/s on one core of the "
                                   "build container (scripts/time_reference_here.py); the reference cannot run on this box"}
    env = sw.SwimmerEnv(device=device)
    env.reset()
    act = np.array([0.3, -0.2])
    for _ in range(200):
        env.step(act)
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(act)
    out["gym_step_us"] = (time.perf_counter() - t0) / steps * 1e6
    h, p, K = env._env1, env._params(), sw.kernels.SingleEnv
    t0 = time.perf_counter()
    for _ in range(steps):
        h.step(p)
        h.io[K.STATE:K.STATE + 8] = h.io[K.NEXT:K.NEXT + 8]
    out["env1_step_us"] = (time.perf_counter() - t0) / steps * 1e6
    env.close()

    class Abs(ctypes.Structure):
        _fields_ = [("numInts", ctypes.c_uint), ("numDoubles", ctypes.c_uint), ("numChars", ctypes.c_uint),
                    ("intArray", ctypes.POINTER(ctypes.c_int)), ("doubleArray", ctypes.POINTER(ctypes.c_double)),
                    ("charArray", ctypes.c_char_p)]
    lib = ctypes.CDLL(os.path.join(os.path.dirname(sw._lib.library_path()), "librlglue_swimmer_hip.so"))
    lib.env_init.restype = ctypes.c_char_p
    lib.env_start.restype = ctypes.c_void_p
    lib.env_step.restype = ctypes.c_void_p
    lib.env_step.argtypes = [ctypes.POINTER(Abs)]
    lib.env_init()
    lib.env_start()
    torque = (ctypes.c_double * 2)(0.3, -0.2)
    a = Abs(0, 2, 0, None, torque, None)
    for _ in range(200):
        lib.env_step(ctypes.byref(a))
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.env_step(ctypes.byref(a))
    out["rlglue_env_step_us"] = (time.perf_counter() - t0) / steps * 1e6
    lib.env_cleanup()
    out["note"] = ("one launch + one host wait per step (sw_env1_step); gym_step_us includes the Python "
                   "list / ndarray conversions of the reference's surface")
    return out


def aux_next_rows(sw, torch, device, n=3, H=1000, directions=512):
    """SURVEY 8(f) rows either side of the hot path, measured: f-1, the estimator's objective
    I(x) over one iteration's device-resident rollouts (one step-kernel launch over every stored
    transition + the norm reduction); f-2, the native twin model's step kernel at a
    bandwidth-bound batch and its rollouts on the lane kernel."""
    import numpy as np
    from swimmer_amd.ars.estimator import Estimator
    out = {}
    state = np.random.get_state()
    ep = sw.EnvParam("LeonSwimmer-Bench", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("Bench", V1=True, n_iter=0, H=H, N=directions, b=directions, alpha=0.0075, nu=0.01,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=0, device=device, record_trajectories=True)
    agent.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    est = Estimator(agent.database, ep, capacity=2 * directions, device=device)
    est.I([1.0, 1.0, 10.0])
    torch.cuda.synchronize()
    samples = []
    for i in range(40):
        t0 = time.perf_counter()
        est.I([1.0 + 1e-3 * i, 1.0, 10.0])      # .item() inside: host-synchronous, like CMA-ES uses it
        samples.append(time.perf_counter() - t0)
    # median: one host hiccup (a 38 ms stall was seen once in 20 evaluations) would own the mean
    dt = sorted(samples)[len(samples) // 2]
    T = 2 * directions * (H - 1)
    out["estimator_objective"] = {"transitions": T, "us_per_evaluation": dt * 1e6,
                                  "us_mean": sum(samples) / len(samples) * 1e6, "us_max": max(samples) * 1e6,
                                  "transitions_per_s": T / dt,
                                  "note": "Estimator.I(x): ONE launch that steps every stored transition of a "
                                          "device-resident store and compares it with its stored next state "
                                          "(sw_step_residual_f64, 144 B per transition) + the sum of its per-workgroup "
                                          "partials + .item()",
                                  "algorithmic_GBps": T * (16 * (2 * n + 2) + 8 * (n - 1)) / dt / 1e9}
    del est, agent
    # f-3: ARS V1 and the true top-b truncation (safe_ars semantics) as options of the same loop;
    # f-4: checkpoint save / load of a running agent
    import tempfile
    for tag, kw, v1 in (("ars_v1_iteration", {}, True), ("ars_top_b_64_iteration", {"top_b": 64}, False)):
        apx = sw.ARSParam("Bench", V1=v1, n_iter=0, H=H, N=directions, b=directions, alpha=0.0075, nu=0.01,
                          safe=False, threshold=0, initial_w="Zero")
        ag = sw.ARSAgent(ep, apx, seed=0, device=device, **kw)
        for _ in range(4):
            ag.run_iteration_async(want_returns=False)
        torch.cuda.synchronize()
        rounds = []
        for _ in range(3):      # median of three rounds: the boxes show a ~38 ms stall every few seconds
            t0 = time.perf_counter()
            for _ in range(20):
                ag.run_iteration_async(want_returns=False)
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0) / 20 * 1e3)
        out[tag] = {"ms_per_iteration": sorted(rounds)[1], "ms_rounds": rounds}
        if not v1:
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, "ck.npz")
                t0 = time.perf_counter()
                ag.save_checkpoint(path)
                t1 = time.perf_counter()
                ag.load_checkpoint(path)
                t2 = time.perf_counter()
                out["checkpoint"] = {"save_ms": (t1 - t0) * 1e3, "load_ms": (t2 - t1) * 1e3,
                                     "bytes": os.path.getsize(path)}
        del ag
    np.random.set_state(state)
    # twin model (SW_FLAG_MODEL_TWIN): step kernel, 4 194 304 envs
    p = sw.SwParams.make(n, 1.0, 1.0, 10.0, 0.01, flags=sw._lib.FLAG_MODEL_TWIN)
    d, m, B = 2 * n + 2, n - 1, 1 << 22
    rng = np.random.default_rng(0)
    st = torch.as_tensor(rng.uniform(-1, 1, (d, B)), device=device)
    ac = torch.as_tensor(rng.uniform(-1, 1, (m, B)), device=device)
    nxt, rew = torch.empty_like(st), torch.empty(B, dtype=torch.float64, device=device)
    plan = sw.kernels.StepPlan(p, st, ac, nxt, rew)
    for _ in range(3):
        plan.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.launch()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e-3 / 10
    byts = (2 * d + m + 1) * 8 * B
    out["twin_step_envs_4194304"] = {"us_per_launch": per * 1e6, "env_steps_per_s": B / per,
                                     "algorithmic_GBps": byts / per / 1e9,
                                     "hbm_frac": byts / per / 1e9 / HBM_PEAK_GBPS}
    del st, ac, nxt, rew, plan
    # twin rollouts: 65 536 rollouts x H = 1000 on the lane kernel (no capture)
    R = 65536
    pol = torch.as_tensor(0.01 * (2 * np.random.RandomState(1).rand(4096, m, d) - 1), device=device).repeat(R // 4096, 1, 1)
    rets = torch.empty(R, dtype=torch.float64, device=device)
    sw.kernels.rollout(p, H, pol, returns=rets)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        sw.kernels.rollout(p, H, pol, returns=rets)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out["twin_rollouts_65536"] = {"ms": ms, "env_steps_per_s": R * H / (ms * 1e-3),
                                  "kernel": f"rollout_kernel<{n},false,true> (twin model, one rollout per lane)"}
    return out


def aux_safe_gate(sw, torch, device, n=3, n_roll=1024, H=1000, lock_step=True):
    """SURVEY 8(f-1), third consumer: the safe-exploration gate (safe_ars/ars.py:111-153) at the size of a BASELINE
    iteration -- 1024 rollouts x H = 1000, every real step preceded by a one-step simulator look-ahead and the cost
    max_i |thetadot_i| (safe_ars/experiment.py:45), thresholds out of reach so that no lane ever stops (the most work
    the gate can ask for: two physics steps per env-step).  Fused: ONE launch (sw_safe_rollouts_f64); lock step: the
    same gate composed from two step-kernel launches + the mask per step (any Python cost callable), 40 steps timed."""
    import numpy as np
    rs = np.random.RandomState(3)
    d, m = 2 * n + 2, n - 1
    pol = torch.as_tensor(0.01 * (2 * rs.rand(n_roll, m, d) - 1), device=device)
    p_real, p_sim = sw.SwParams.make(n, 0.8, 1.2, 10.2, 1e-3), sw.SwParams.make(n)
    traj = torch.empty((H, d, n_roll), dtype=torch.float64, device=device)
    first = torch.empty(n_roll, dtype=torch.int32, device=device)

    def timed(p):
        def go():
            sw.kernels.safe_rollouts(p, p_sim, H, pol, sw._lib.COST_MAX_ABS_THETADOT, 0, 1e9, 1e9, traj=traj,
                                     first_refused=first)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for a, b in ev:
            a.record()
            go()
            b.record()
        torch.cuda.synchronize()
        assert int((first != H).sum().item()) == 0
        return sorted(a.elapsed_time(b) for a, b in ev)[4]
    ms = timed(p_real)                                      # n = 3, <= 8192 rollouts: the mirror-quad form
    ms_lane = timed(sw.SwParams.make(n, 0.8, 1.2, 10.2, 1e-3, flags=sw._lib.FLAG_ROLLOUT_LANE))
    out = {"rollouts": n_roll, "horizon": H, "fused_ms_per_launch": ms, "env_steps_per_s": n_roll * H / (ms * 1e-3),
           "physics_steps_per_s": 2 * n_roll * H / (ms * 1e-3),
           "kernel": ("safe_rollout_oct3_kernel<true,false> (two mirror quads per rollout: one geometry, two dynamics "
                      "per env-step, 170 instructions)" if n == 3 else
                      f"safe_rollout_row_kernel<{n}> (one segment per lane: two row_steps per env-step)"),
           "lane_form_ms_per_launch": ms_lane, "trajectory_capture": True}
    if not lock_step:
        return out
    real = sw.SwimmerEnv(n=n, l_i=0.8, m_i=1.2, k=10.2, device=device)
    agent = sw.safe_ars.Safe_ARS(sw.safe_ars.MaxAbsThetaDot(), 1e9, 1e9, sw.SwimmerEnv(n=n, device=device))
    P = pol.cpu().numpy()
    agent.rollouts(real, P, 8, fused=False)
    t0 = time.perf_counter()
    agent.rollouts(real, P, 40, fused=False)
    out["lock_step_us_per_step"] = (time.perf_counter() - t0) / 40 * 1e6
    out["lock_step_note"] = "per step of the whole 1024-rollout batch: two step-kernel launches + the cost and the mask in torch"
    return out


def aux_rollout_saturated(sw, torch, device, n=3, n_roll=262144, H=1000, reps=15):
    """Where the ROLLOUT path is HBM-bound: the lane-per-rollout kernel on a batch that fills the
    chip (262 144 rollouts x H = 1000) with every post-step state captured, 16.8 GB of stores."""
    import numpy as np
    p = sw.SwParams.make(n, flags=sw._lib.FLAG_ROLLOUT_LANE)
    d, m = 2 * n + 2, n - 1
    rng = np.random.RandomState(1)
    pol = torch.as_tensor(0.01 * (2 * rng.rand(4096, m, d) - 1), device=device).repeat(n_roll // 4096, 1, 1)
    traj = torch.empty((H, d, n_roll), dtype=torch.float64, device=device)
    rets = torch.empty(n_roll, dtype=torch.float64, device=device)
    # Median of `reps` launches, each with its own events, after three warm-up launches.  Until round 3 this
    # leg timed THREE launches after ONE warm-up, wherever the legs before it had left the chip's clocks: on
    # one box the same binary then reads 0.51 .. 0.61 of the HBM peak at 65 536 rollouts (idle before / full
    # load before; profiles/r04_a_lane_ab.log), which is all the "slide" of this number over round 3 was --
    # the kernel's machine code is identical to round 2's.
    for _ in range(3):
        sw.kernels.rollout(p, H, pol, traj=traj, returns=rets)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        sw.kernels.rollout(p, H, pol, traj=traj, returns=rets)
        e1.record()
    torch.cuda.synchronize()
    samples = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    ms = samples[len(samples) // 2]
    byts = n_roll * H * 8 * d + n_roll * (8 * m * d + 8)
    del traj
    torch.cuda.empty_cache()
    return {"kernel": f"rollout_kernel<{n},false,false> (one rollout per lane)",
            "rollouts": n_roll, "horizon": H, "trajectory_capture": True, "ms": ms,
            "ms_min": samples[0], "ms_max": samples[-1], "launches": reps,
            "env_steps_per_s": n_roll * H / (ms * 1e-3), "algorithmic_bytes": byts,
            "achieved_GBps": byts / (ms * 1e-3) / 1e9,
            "hbm_frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}


class ArsLeg(object):
    """One ARS V2 workload on this rank: agent + the timed loop + the kernel-timing post-pass."""

    def __init__(self, sw, torch, n, H, N, device, direct_rccl=None):
        self.sw, self.torch, self.n, self.H, self.N = sw, torch, n, H, N
        ep = sw.EnvParam("LeonSwimmer-Bench", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
        ap = sw.ARSParam("Bench", V1=False, n_iter=0, H=H, N=N, b=N, alpha=0.0075, nu=0.01,
                         safe=False, threshold=0, initial_w="Zero")
        self.agent = sw.ARSAgent(ep, ap, seed=0, device=device, full_covariance=True, direct_rccl=direct_rccl)

    def run(self, warmup, steps, sync, time_every=TIME_EVERY, postpass=POSTPASS_LAUNCHES):
        agent, torch = self.agent, self.torch
        for _ in range(warmup):
            agent.run_iteration_async(want_returns=False)
        sync()
        # HIP events on the launch stream around every k-th rollout launch of the timed region:
        # a timed launch costs ~10 us of pipeline bubbles (measured), so timing all of them
        # would slow the very loop being measured by 3 %
        agent._pipe.timing(time_every)
        t0 = time.perf_counter()
        for _ in range(steps):
            agent.run_iteration_async(want_returns=False)
        sync()
        dt = time.perf_counter() - t0
        k_ms, k_n = agent._pipe.rollout_ms()
        # post-pass outside the timed region: every launch timed, and (world > 1) the collective
        res = {"seconds": dt, "kernel_ms_timed_region": k_ms, "kernel_samples_timed_region": k_n}
        if postpass > 0 and agent.n_local > 0:
            agent._pipe.timing(1)
            agent.collective_timing(True)
            for _ in range(postpass):
                agent.run_iteration_async(want_returns=False)
            sync()
            p_ms, p_n = agent._pipe.rollout_ms()
            res.update(kernel_ms_postpass=p_ms, kernel_samples_postpass=p_n,
                       collective_us=agent.collective_us())
            agent.collective_timing(False)
            tot = k_ms * k_n + p_ms * p_n
            res.update(kernel_ms=tot / (k_n + p_n), kernel_samples=k_n + p_n)
        else:
            res.update(kernel_ms=k_ms, kernel_samples=k_n)
        agent._pipe.timing(0)
        return res

    def check(self, rank):
        import numpy as np
        bad = int((self.agent._status != 0).sum().item())
        if bad or not np.isfinite(self.agent.policy).all():
            raise SystemExit(f"rank {rank}: bench produced {bad} bad rollouts / non-finite policy")


def leg_roofline(n, n_local, H, kern_ms, full_chip=None):
    alg = rollout_algorithmic_bytes(n, n_local, H)
    ach = alg / (kern_ms * 1e-3) / 1e9 if kern_ms else None
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS if ach else None, "algorithmic_bytes": alg,
            "kernel": rollout_kernel_name(n), "kernel_ms": kern_ms,
            "issue_bound": issue_bound(n, H, kern_ms, full_chip)}


def aux_collective_one_rank(n, H, directions, timeout=150, direct=False):
    """What the per-iteration all-gather costs BEFORE any wire time: the same ARS loop with a
    one-rank RCCL process group and the collective forced (ProcessGroupNCCL + the RCCL kernel on
    the critical stream).  The multi-GPU iteration is this plus the time on the xGMI links.
    Runs in a CHILD process with a timeout: a communication library that fails to come up on
    some box must not take the N = 1 line with it."""
    cmd = [sys.executable, os.path.abspath(__file__), "--collective-one-rank-child",
           "--segments", str(n), "--horizon", str(H), "--directions", str(directions)]
    if direct:      # the same loop with ncclAllGather called from native code (sw_comm_all_gather_f64)
        cmd.append("--direct-rccl")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"one-rank RCCL leg did not finish in {timeout} s"}
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": f"one-rank RCCL leg failed (exit code {out.returncode}): {out.stderr[-300:]}"}
    return json.loads(lines[-1])


def collective_one_rank_child(args):
    """Child of aux_collective_one_rank: prints one JSON object."""
    import torch
    import torch.distributed as dist
    import swimmer_amd as sw
    from swimmer_amd.ars import sharding
    device = "cuda:0"
    torch.cuda.set_device(0)
    sw._lib.load()
    with stdout_to_stderr():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0,
                                world_size=1, device_id=torch.device(device))
        dist.barrier()      # the communicator (and its banner) come up here at the latest
        torch.cuda.set_stream(torch.cuda.Stream(device))
        sharding._FORCE_COLLECTIVE = True
        iters = 20
        leg = ArsLeg(sw, torch, args.segments, args.horizon, args.directions, device,
                     direct_rccl=args.direct_rccl)
        r = leg.run(3, iters, torch.cuda.synchronize, time_every=4, postpass=16)
        leg.check(0)
        import hashlib
        res = {"directions": args.directions, "ms_per_iteration": r["seconds"] / iters * 1e3,
               "collective_us": r.get("collective_us"), "backend": dist.get_backend(),
               "path": "sw_comm_all_gather_f64 (native)" if args.direct_rccl else "torch.distributed",
               "policy_sha256": hashlib.sha256(leg.agent.policy.tobytes()).hexdigest()[:16],
               "note": "one rank, all-gather forced: framework + RCCL launch cost per iteration, no wire time"}
        dist.destroy_process_group()
    print(json.dumps(res), flush=True)


def aux_ars_shard(sw, torch, n, H, directions, device, iters=24, warm=10, probe_waves=0):
    """One GPU's shard of a sharded config (configs[3]: n = 3, configs[4]: n = 6; 2048
    directions over 8 GPUs = 256 per GPU) as a self-contained ARS iteration loop."""
    import numpy as np
    state = np.random.get_state()
    leg = ArsLeg(sw, torch, n, H, directions, device)
    # ten warm-up iterations: a full-chip f64 launch only settles at its sustained clock after a few
    # milliseconds of load (2048 directions, n = 6: 0.78 ms per iteration after 3 warm-ups, 0.67 after 10)
    r = leg.run(warm, iters, torch.cuda.synchronize, time_every=4, postpass=8)
    # median of three timed rounds (the main leg times exactly K steps once, as the contract says; these
    # legs can afford to shrug off a stall of the box or a busy host)
    rounds = [r["seconds"]] + [leg.run(0, iters, torch.cuda.synchronize, time_every=4, postpass=0)["seconds"]
                               for _ in range(2)]
    leg.check(0)
    np.random.set_state(state)
    dt = sorted(rounds)[1] / iters
    # probe_waves: price the launch with the intervals of a chip that is as full as this launch makes it
    # (measured right here, while the chip is still warm from the leg)
    fc = full_chip_intervals(sw, device, probe_waves) if probe_waves else None
    return {"segments": n, "directions": directions, "ms_per_iteration": dt * 1e3,
            "ms_rounds": [x / iters * 1e3 for x in rounds],
            "env_steps_per_s": 2 * directions * H / dt,
            "roofline": leg_roofline(n, directions, H, r["kernel_ms"], fc)}


def aux_host_cost(sw, torch, device, gpu_ms):
    """Host time per pipelined ARS iteration at the sizes an 8-rank job puts on EVERY rank's host (every rank
    draws every direction's noise: ars_agent.py:137-138 runs once per direction, serially, in the reference):
    with H = 10 the GPU needs a few microseconds per iteration, so the loop time IS the host time (native
    MT19937 stream + pinned H2D + two launches + Python).  `host_bound` = the host needs more than 0.8 x the
    GPU iteration of that rank's shard: the ranks would then wait for their own hosts, not for each other."""
    import time
    import numpy as np
    state = np.random.get_state()
    out = {}
    for tag, n, N, shard_key in (("n3_4096_directions", 3, 4096, "n3"), ("n6_2048_directions", 6, 2048, "n6"),
                                 ("n6_4096_directions", 6, 4096, "n6")):
        ep = sw.EnvParam("LeonSwimmer-Host", n=n, H=10, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
        ap = sw.ARSParam("Host", V1=False, n_iter=0, H=10, N=N, b=N, alpha=0.0075, nu=0.01,
                         safe=False, threshold=0, initial_w="Zero")
        ag = sw.ARSAgent(ep, ap, seed=0, device=device, full_covariance=True)
        for _ in range(150):
            ag.run_iteration_async(want_returns=False)
        torch.cuda.synchronize()
        rounds = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(150):
                ag.run_iteration_async(want_returns=False)
            rounds.append((time.perf_counter() - t0) / 150 * 1e6)
            torch.cuda.synchronize()
        host_us = sorted(rounds)[1]
        g = gpu_ms.get(shard_key)
        out[tag] = {"host_us_per_iteration": host_us, "us_rounds": rounds,
                    "noise_doubles_per_iteration": N * (n - 1) * (2 * n + 2),
                    "gpu_us_per_iteration_of_a_rank": None if g is None else g * 1e3,
                    "host_bound": None if g is None else bool(host_us > 0.8 * g * 1e3)}
        del ag
    np.random.set_state(state)
    return out


def summary(line, aux):
    """Flat digest of the line: one scalar per number a reader of the driver's record needs."""
    def dig(obj, *path):
        for k in path:
            if not isinstance(obj, dict) or k not in obj or obj[k] is None:
                return None
            obj = obj[k]
        return obj

    def r3(x):
        return None if x is None else float(f"{x:.4g}")

    roof = line["roofline"]
    m = f"n{dig(line, 'config', 'segments') or 3}"      # the main leg's keys carry its chain length (default 3)
    out = {m + "_ms": r3(line["ms_per_step"]), m + "_sps": r3(line["value"]), m + "_kernel_ms": r3(roof.get("kernel_ms")),
           m + "_hbm_frac": r3(roof.get("frac")), m + "_issue_frac": r3(roof.get("issue_bound_frac")),
           m + "_instr": roof.get("instructions_per_step"), "traffic_stale": roof.get("traffic_stale")}
    if dig(line, "n_gpus") not in (None, 1):
        out["n_gpus"] = line["n_gpus"]
        out["collective_us"] = r3(dig(aux, "collective_us"))
        out["strong2048_ms"] = r3(dig(aux, "strong_2048_directions", "ms_per_iteration"))
        out["strong2048_sps"] = r3(dig(aux, "strong_2048_directions", "env_steps_per_s"))
    for tag, key in (("n3_sh256", "shard_n3_256_directions"), ("n6_sh256", "shard_n6_256_directions"),
                     ("n3_2048_1gpu", "ars_2048_directions_one_gpu"), ("n6_2048_1gpu", "ars_2048_directions_one_gpu_n6")):
        out[tag + "_ms"] = r3(dig(aux, key, "ms_per_iteration"))
        out[tag + "_sps"] = r3(dig(aux, key, "env_steps_per_s"))
        out[tag + "_kernel_ms"] = r3(dig(aux, key, "roofline", "kernel_ms"))
        out[tag + "_hbm_frac"] = r3(dig(aux, key, "roofline", "frac"))
        out[tag + "_issue_frac"] = r3(dig(aux, key, "roofline", "issue_bound", "frac"))
        out[tag + "_over_priced"] = r3(dig(aux, key, "roofline", "issue_bound", "measured_over_priced"))
        if "2048" in tag:
            out[tag + "_over_priced_full_chip"] = r3(dig(aux, key, "roofline", "issue_bound",
                                                         "measured_over_priced_full_chip"))
    out.update({
        "step8192_us": r3(dig(aux, "step_only", "envs_8192", "us_per_launch")),
        "step8192_sps": r3(dig(aux, "step_only", "envs_8192", "env_steps_per_s")),
        "step8192_graph_us": r3(dig(aux, "step_only", "envs_8192", "graph_us_per_step")),
        "step4m_hbm_frac": r3(dig(aux, "step_only", "envs_4194304", "hbm_frac")),
        "sat262144_hbm_frac": r3(dig(aux, "rollout_saturated", "hbm_frac")),
        "sat65536_hbm_frac": r3(dig(aux, "rollout_saturated_65536", "hbm_frac")),
        "sat_n6_hbm_frac": r3(dig(aux, "rollout_saturated_n6", "hbm_frac")),
        "sat_n6_sps": r3(dig(aux, "rollout_saturated_n6", "env_steps_per_s")),
        "twin_step4m_hbm_frac": r3(dig(aux, "next_rows", "twin_step_envs_4194304", "hbm_frac")),
        "coll1_us": r3(dig(aux, "collective_one_rank", "collective_us")),
        "coll1_ms": r3(dig(aux, "collective_one_rank", "ms_per_iteration")),
        "coll1_direct_us": r3(dig(aux, "collective_one_rank_direct", "collective_us")),
        "coll1_direct_ms": r3(dig(aux, "collective_one_rank_direct", "ms_per_iteration")),
        "host_us_n3_4096": r3(dig(aux, "host_cost_at_8_ranks", "n3_4096_directions", "host_us_per_iteration")),
        "host_us_n6_2048": r3(dig(aux, "host_cost_at_8_ranks", "n6_2048_directions", "host_us_per_iteration")),
        "host_us_n6_4096": r3(dig(aux, "host_cost_at_8_ranks", "n6_4096_directions", "host_us_per_iteration")),
        "host_bound_n3_4096": dig(aux, "host_cost_at_8_ranks", "n3_4096_directions", "host_bound"),
        "host_bound_n6_2048": dig(aux, "host_cost_at_8_ranks", "n6_2048_directions", "host_bound"),
        "host_bound_n6_4096": dig(aux, "host_cost_at_8_ranks", "n6_4096_directions", "host_bound"),
        "gym_step_us": r3(dig(aux, "single_env", "gym_step_us")),
        "env1_step_us": r3(dig(aux, "single_env", "env1_step_us")),
        "rlglue_step_us": r3(dig(aux, "single_env", "rlglue_env_step_us")),
        "ref_cpu_step_us_build_container_const": r3(dig(aux, "single_env", "reference_cpu_us_per_step")),
        "estI_us": r3(dig(aux, "next_rows", "estimator_objective", "us_per_evaluation")),
        "gate_ms": r3(dig(aux, "next_rows", "safe_ars_gate", "fused_ms_per_launch")),
        "gate_sps": r3(dig(aux, "next_rows", "safe_ars_gate", "env_steps_per_s")),
        "gate_n6_ms": r3(dig(aux, "next_rows", "safe_ars_gate_n6", "fused_ms_per_launch")),
        "v1_ms": r3(dig(aux, "next_rows", "ars_v1_iteration", "ms_per_iteration")),
        "topb_ms": r3(dig(aux, "next_rows", "ars_top_b_64_iteration", "ms_per_iteration")),
        "cpu_sps": r3(dig(line, "cpu_baseline", "value")), "cpu_cores": dig(line, "cpu_baseline", "cores"),
    })
    return {k: v for k, v in out.items() if v is not None}


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal knobs (never set by the driver): run the multi-rank code path on a one-GPU box,
    # every rank on cuda:0 with gloo staged through the host instead of RCCL.
    backend = os.environ.get("SWIMMER_BENCH_BACKEND", "nccl")
    single_device = bool(os.environ.get("SWIMMER_BENCH_SINGLE_DEVICE"))
    if single_device:
        local = 0
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows "
                         f"{torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():    # RCCL / gloo banners must not land in front of the result line
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(device))
            else:
                dist.init_process_group(backend)
            dist.barrier()

    import swimmer_amd as sw
    sw._lib.load()

    # Everything runs on a stream of its own, not on the null stream: work on the null stream is
    # implicitly ordered against every other blocking stream of the process, and once a
    # ProcessGroupNCCL exists that costs the iteration 13 us of device-side waits (measured with
    # one rank: 0.3031 vs 0.2903 ms; scripts/collective_overhead.py).  The library itself
    # follows the caller's current stream.
    torch.cuda.set_stream(torch.cuda.Stream(device))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def same_policy_everywhere(agent, what):
        # every rank must hold the same policy (redundant deterministic update)
        pol = torch.as_tensor(agent.policy)
        ref = pol.clone() if backend != "nccl" else pol.to(device)
        mine = ref.clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, mine):
            raise SystemExit(f"rank {rank}: policy differs from rank 0 after the {what} loop")

    n, H = args.segments, args.horizon
    N = args.directions * world if args.scaling == "weak" else args.total_directions
    if rank == 0:
        calibrate_issue_intervals(sw, device)
    leg = ArsLeg(sw, torch, n, H, N, device, direct_rccl=(True if args.direct_rccl else None))
    agent = leg.agent
    res = leg.run(args.warmup, args.steps, sync)
    dt = max_over_ranks(res["seconds"])
    if world > 1:
        same_policy_everywhere(agent, "timed")
    leg.check(rank)

    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_aux:
        # the fixed-size problem of configs[3] / [4] on the same ranks (strong scaling), so one
        # driver run per N yields both curves
        sleg = ArsLeg(sw, torch, n, H, args.total_directions, device)
        sres = sleg.run(3, 12, sync, postpass=8)
        sdt = max_over_ranks(sres["seconds"]) / 12
        same_policy_everywhere(sleg.agent, "strong-scaling")
        sleg.check(rank)
        strong = {"directions_total": args.total_directions,
                  "directions_per_gpu": sleg.agent.chunk, "ms_per_iteration": sdt * 1e3,
                  "env_steps_per_s": 2 * args.total_directions * H / sdt,
                  "kernel_ms": sres["kernel_ms"], "collective_us": sres.get("collective_us")}
        del sleg

    if rank == 0:
        steps_per_iter = 2 * N * H
        value = steps_per_iter * args.steps / dt
        d = 2 * n + 2
        kern_ms = res["kernel_ms"]
        assert res["kernel_samples_timed_region"] == -(-args.steps // TIME_EVERY)
        local_steps = 2 * agent.n_local * H
        roof = leg_roofline(n, agent.n_local, H, kern_ms)
        flops_per_step = {3: 330.0, 6: 1100.0}.get(n, 40.0 * n * n)  # fp64 flop count, DESIGN.md
        traffic = pmc_traffic(rollout_kernel_name(n), n, agent.n_local, H)
        cov_alg = local_steps * 8 * d       # the ride-along covariance pass reads every state once
        stale = bool(traffic and traffic.get("stale"))
        ib = roof.get("issue_bound") or {}
        roof.update({
            "traffic": (traffic or {}).get("traffic_bytes"),
            "traffic_stale": stale,     # True: the PMC passes ran on OTHER kernel sources than this build
            "traffic_breakdown": (traffic or {}).get("breakdown"),
            "traffic_source": (traffic or {}).get("source"),
            # scalars of issue_bound once more at this level (a record that keeps scalar members only
            # still shows what bounds the kernel)
            "issue_bound_frac": ib.get("frac"), "issue_floor_ms": ib.get("floor_ms"),
            "instructions_per_step": ib.get("instructions_per_step"),
            "simd_occupancy": min(1.0, -(-2 * agent.n_local // (8 if n == 3 else 4)) / 1024.0),
            # the launch does TWO jobs: this iteration's rollouts (the bytes `achieved` counts)
            # and the covariance pass over the previous iteration's trajectories (extra
            # workgroups of the same grid).  Against the algorithmic bytes of both, the counters
            # show no wasted traffic:
            "algorithmic_bytes_with_covariance_pass": roof["algorithmic_bytes"] + cov_alg,
            "traffic_over_algorithmic": (traffic["traffic_bytes"] / (roof["algorithmic_bytes"] + cov_alg)
                                         if traffic and not stale else None),
            "kernel_ms_timed_region": res["kernel_ms_timed_region"],
            "kernel_samples_timed_region": res["kernel_samples_timed_region"],
            "kernel_ms_postpass": res.get("kernel_ms_postpass"),
            "kernel_samples": res["kernel_samples"],
            "note": "the fused rollout is fp64-VALU-issue bound by construction (state, policy and "
                    "sums stay in registers; see issue_bound); HBM is the contract's roofline. "
                    "The HBM-bound regimes of this path are aux.step_only (step kernel, >= 1e6 "
                    "envs) and aux.rollout_saturated (rollouts filling the chip).  kernel_ms = "
                    "mean over the sampled launches of the timed region and every launch of an "
                    "untimed post-pass",
            "fp64_tflops": local_steps * flops_per_step / (kern_ms * 1e-3) / 1e12,
            "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS})
        if args.scaling == "weak":
            workload = (f"ARS V2 iteration, {n}-segment swimmer, {args.directions} directions/GPU x 2 "
                        f"rollouts x H={H} (BASELINE configs[2] per GPU)")
        else:
            workload = (f"ARS V2 iteration, {n}-segment swimmer, {N} directions in all x 2 rollouts x "
                        f"H={H}, sharded over {world} GPU(s) (BASELINE configs[{3 if n == 3 else 4}])")
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "directions_total": N, "directions_per_gpu": agent.chunk, "horizon": H,
                       "segments": n, "trajectory_capture": True, "full_covariance": True,
                       "parallelism": f"directions sharded over {world} GPU(s), "
                                      "1 all-gather/iteration",
                       "ranks_seen": dist.get_world_size() if world > 1 else 1,
                       "backend": (dist.get_backend() if world > 1 else None)},
            "roofline": roof,
        }
        aux = {}
        if world > 1:
            aux["collective_us"] = res.get("collective_us")
            if strong:
                aux["strong_2048_directions"] = strong
        # the step-only sweep, the shard legs and the CPU baseline belong to the N = 1 line only;
        # none of them may take the headline numbers above with it
        def guarded(fn, *a, **kw):
            try:
                return fn(*a, **kw)
            except Exception as exc:   # noqa: BLE001 -- reported in the line
                return {"error": f"{type(exc).__name__}: {exc}"}

        if not args.no_aux and world == 1:
            aux["step_only"] = guarded(aux_step_only, sw, torch, n, device)
            # configs[3]'s whole problem on ONE GPU
            aux["ars_2048_directions_one_gpu"] = guarded(aux_ars_shard, sw, torch, n, H, 2048, device,
                                                         probe_waves=(2 if n == 3 else 4))
            if n != 6:   # configs[4]'s whole problem on ONE GPU (4096 rollouts = a wave on every SIMD)
                aux["ars_2048_directions_one_gpu_n6"] = guarded(aux_ars_shard, sw, torch, 6, H, 2048, device,
                                                                probe_waves=4)
            aux["shard_n3_256_directions"] = guarded(aux_ars_shard, sw, torch, 3, H, 256, device)
            aux["shard_n6_256_directions"] = guarded(aux_ars_shard, sw, torch, 6, H, 256, device)
            aux["collective_one_rank"] = guarded(aux_collective_one_rank, n, H, args.directions)
            aux["collective_one_rank_direct"] = guarded(aux_collective_one_rank, n, H, args.directions, direct=True)
            # what every rank's HOST does per iteration at the 8-rank sizes, against the GPU iteration of a rank
            # (n = 3: this line's main leg, 512 directions per GPU; n = 6: the 256-direction shard -- the launch
            # time is the same up to 512 directions of n = 6, a wave on every fourth SIMD)
            gpu_ms = {"n3": dt / args.steps * 1e3 if n == 3 else None,
                      "n6": (aux.get("shard_n6_256_directions") or {}).get("ms_per_iteration")}
            aux["host_cost_at_8_ranks"] = guarded(aux_host_cost, sw, torch, device, gpu_ms)
            aux["next_rows"] = guarded(aux_next_rows, sw, torch, device)
            if isinstance(aux["next_rows"], dict):
                aux["next_rows"]["safe_ars_gate"] = guarded(aux_safe_gate, sw, torch, device)
                aux["next_rows"]["safe_ars_gate_n6"] = guarded(aux_safe_gate, sw, torch, device, n=6, n_roll=512,
                                                               lock_step=False)
            aux["rollout_saturated"] = guarded(aux_rollout_saturated, sw, torch, device)
            # one wave per SIMD (65 536 rollouts, a 4.2 GB buffer): the same kernel streams faster
            # than with four (16.8 GB, a 2 MB stride between the rows a step writes)
            aux["rollout_saturated_65536"] = guarded(aux_rollout_saturated, sw, torch, device, n_roll=65536)
            # the six-segment chain in the same regime (configs[4]'s swimmer, one rollout per lane, 112 B per env-step;
            # H = 500 keeps the buffer at 3.7 GB): where the lane kernel's O(n^2 .. n^3) step stands against HBM
            aux["rollout_saturated_n6"] = guarded(aux_rollout_saturated, sw, torch, device, n=6, n_roll=65536,
                                                  H=500, reps=9)
        if not args.no_aux and world == 1:
            aux["single_env"] = guarded(aux_single_env, sw, torch, device)
        if aux:
            line["aux"] = aux
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = guarded(cpu_baseline, n, H, args.directions, args.cpu_seconds)
        # LAST key, flat and short (< 1900 characters): every config's number survives a record
        # that keeps only the tail of the line and only scalar members of its objects
        line["summary"] = summary(line, aux)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.collective_one_rank_child:
        return collective_one_rank_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
