#!/usr/bin/env python3
"""bench.py -- headline benchmark of the swimmer + ARS hot path on MI355X.

One "step" = one ARS V2 iteration on the 3-segment swimmer (BASELINE.json configs[2]):
sample N perturbations on the host (NumPy legacy RNG, like the reference), run the 2N
H-step rollouts in the fused HIP rollout kernel (trajectory capture + V2 moments on), gather
returns (RCCL all-gather when --gpus > 1), update the policy and the running state
statistics, and reduce the full state covariance from the recorded trajectories.
Weak scaling: every GPU gets --directions (default 512) directions, so the whole job runs
N = 512 * n_gpus directions (4 GPUs = the 2048-direction config).

metric = env-steps/s over the whole job = 2 * N * H * steps / wall time of the timed region.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      dominant kernel (rollout): algorithmic HBM bytes per launch / HIP-event
                duration of that launch, against the 8 TB/s HBM3E peak
  cpu_baseline  the C restatement of the reference (oracle/, OpenMP over the host cores) on
                a bounded sample of the same workload
and an "aux" object with the physics-step-only kernel (configs[1]) at 8192 envs and at a
bandwidth-bound batch.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MEASURED_COPY_PEAK_GBPS = 5690.0   # scripts/ubench/stream_copy on the box (profiles/r01_g_stream_copy.log)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
# Instructions one wave issues per env-step in the segment-per-lane rollout kernels (ISA count
# of the hot loop incl. trajectory stores + moments, scripts/isa_loop_stats.py) and the measured
# issue interval of a lone wave (profiles/r01_ubench_issue_cost.log: 1.92-2.13 ns per
# independent instruction of any kind): what actually bounds the latency-bound rollout.
ROLLOUT_INSTR_PER_STEP = {3: 145, 6: 256}
TIME_EVERY = 8
LONE_WAVE_NS_PER_INSTR = 1.90   # lower edge of the measured per-instruction intervals (v_mov_b64 1.92, f64 + SALU 1.98, f64 FMA 2.13; DPP-heavy mixes come in slightly under)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--directions", type=int, default=512, help="ARS directions per GPU")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--segments", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-aux", action="store_true")
    return ap.parse_args()


def pmc_traffic(kernel, n, directions, H):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None when
    the measured workload is not the one being benchmarked."""
    path = os.path.join(ROOT, "profiles", "r01_j_pmc_traffic.json")
    if not os.path.exists(path) or (n, directions, H) != (3, 512, 1000):
        return None
    return json.load(open(path)).get(kernel, {})


def issue_bound(n, H, kern_ms):
    """The rollout kernel's real ceiling: every wave runs alone on its SIMD and can issue one
    instruction per ~2.1 ns, so a rollout batch cannot finish faster than
    H x instructions-per-step x that interval, whatever the batch size."""
    if n not in ROLLOUT_INSTR_PER_STEP:
        return None
    floor_ms = H * ROLLOUT_INSTR_PER_STEP[n] * LONE_WAVE_NS_PER_INSTR * 1e-6
    return {"instructions_per_step": ROLLOUT_INSTR_PER_STEP[n],
            "lone_wave_ns_per_instruction": LONE_WAVE_NS_PER_INSTR,
            "floor_ms": floor_ms, "frac": floor_ms / kern_ms}


def cpu_baseline(n, H, directions, seconds):
    """Time the oracle (C port of the reference step/rollout, OpenMP) on whole rollout
    batches of the benchmark's shape until `seconds` have elapsed."""
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
            "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{done} batches of {2 * directions} rollouts x H={H} (n={n}), "
                      f"oracle/swimmer_oracle.c with OpenMP, {dt:.1f} s"}


def aux_step_only(sw, n, device):
    """Physics-step-only kernel (configs[1]): 8192 envs, and a bandwidth-bound batch."""
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
    return out


def aux_more_directions(sw, n, H, device, directions=2048, iters=12):
    """The same ARS iteration at configs[3]'s problem size (2048 directions = 4096 rollouts) on
    ONE GPU: the rollout kernel is latency-bound, so the larger batch rides along almost free."""
    ep = sw.EnvParam("LeonSwimmer-Bench", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("Bench", V1=False, n_iter=iters, H=H, N=directions, b=directions, alpha=0.0075,
                     nu=0.01, safe=False, threshold=0, initial_w="Zero")
    state = np.random.get_state()
    agent = sw.ARSAgent(ep, ap, seed=0, device=device, full_covariance=True)
    for _ in range(3):
        agent.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        agent.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    np.random.set_state(state)
    del agent
    return {"directions": directions, "ms_per_iteration": dt * 1e3,
            "env_steps_per_s": 2 * directions * H / dt}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one process per GPU)")
    # Rehearsal knobs (never set by the driver): run the multi-rank code path on a one-GPU box,
    # every rank on cuda:0 with gloo staged through the host instead of RCCL.
    backend = os.environ.get("SWIMMER_BENCH_BACKEND", "nccl")
    if os.environ.get("SWIMMER_BENCH_SINGLE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    import swimmer_amd as sw
    sw._lib.load()

    # Everything runs on a stream of its own, not on the null stream: work on the null stream is
    # implicitly ordered against every other blocking stream of the process, and once a
    # ProcessGroupNCCL exists that costs the iteration 13 us of device-side waits (measured with
    # one rank: 0.3031 vs 0.2903 ms; scripts/collective_overhead.py).  The library itself
    # follows the caller's current stream.
    torch.cuda.set_stream(torch.cuda.Stream(device))

    n, H = args.segments, args.horizon
    N = args.directions * world
    ep = sw.EnvParam("LeonSwimmer-Bench", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("Bench", V1=False, n_iter=args.steps, H=H, N=N, b=N, alpha=0.0075, nu=0.01,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=0, device=device, full_covariance=True)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        agent.run_iteration_async(want_returns=False)
    sync()
    # HIP events on the launch stream around every 8th rollout launch of the timed region: a
    # timed launch costs ~10 us of pipeline bubbles (measured), so timing all of them would
    # slow the very loop being measured by 3 %
    agent._pipe.timing(TIME_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.run_iteration_async(want_returns=False)
    sync()
    dt = time.perf_counter() - t0
    kern_ms, kern_launches = agent._pipe.rollout_ms()
    agent._pipe.timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank must hold the same policy (redundant deterministic update)
        pol = torch.as_tensor(agent.policy)
        ref = pol.clone() if backend != "nccl" else pol.to(device)
        mine = ref.clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, mine):
            raise SystemExit(f"rank {rank}: policy differs from rank 0 after {args.steps} iterations")
    bad = int((agent._status != 0).sum().item())
    if bad or not np.isfinite(agent.policy).all():
        raise SystemExit(f"bench produced {bad} bad rollouts / non-finite policy")

    if rank == 0:
        steps_per_iter = 2 * N * H
        value = steps_per_iter * args.steps / dt
        d = 2 * n + 2
        assert kern_launches == -(-args.steps // TIME_EVERY)
        local_steps = 2 * agent.n_local * H
        # algorithmic HBM bytes of one rollout launch: every post-step state is
        # materialised (8 d bytes per env-step, as the reference does, ars/environment.py:53)
        # + per rollout its delta row (8 m d), return (8) and status (4)
        alg_bytes = local_steps * 8 * d + 2 * agent.n_local * (8 * (n - 1) * d + 12)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        flops_per_step = {3: 330.0, 6: 1100.0}.get(n, 40.0 * n * n)  # fp64 flop count, DESIGN.md
        traffic = pmc_traffic("rollout_quad3_kernel<true,true,true>", n, args.directions, H)
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ARS V2 iteration, {n}-segment swimmer, "
                                   f"{args.directions} directions/GPU x 2 rollouts x H={H} "
                                   f"(BASELINE configs[2] per GPU)",
                       "directions_total": N, "horizon": H, "segments": n,
                       "trajectory_capture": True, "full_covariance": True,
                       "parallelism": f"directions sharded over {world} GPU(s), "
                                      "1 all-gather/iteration"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": (traffic or {}).get("traffic_bytes"),
                         "traffic_breakdown": (traffic or {}).get("breakdown"),
                         "algorithmic_bytes": alg_bytes,
                         "kernel": ("rollout_quad3_kernel<true,true,true>" if n == 3
                                    else f"rollout_row_kernel<{n},true,true,true>" if n >= 4
                                    else f"rollout_kernel<{n},true,false>"),
                         "kernel_ms": kern_ms,
                         "note": "the fused rollout is fp64-VALU-latency bound by construction "
                                 "(state, policy and sums stay in registers); HBM is the "
                                 "contract's roofline, see DESIGN.md and aux.step_only; the launch "
                                 "also carries the covariance pass over the previous iteration's "
                                 "trajectories (traffic_breakdown), achieved counts the rollouts' "
                                 "bytes only",
                         "fp64_tflops": local_steps * flops_per_step / (kern_ms * 1e-3) / 1e12,
                         "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                         "issue_bound": issue_bound(n, H, kern_ms)},
        }
        # the step-only sweep and the CPU baseline belong to the N = 1 line only
        if not args.no_aux and world == 1:
            line["aux"] = {"step_only": aux_step_only(sw, n, device),
                           "ars_2048_directions_one_gpu": aux_more_directions(sw, n, H, device)}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n, H, args.directions, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
