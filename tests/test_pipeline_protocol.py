"""Protocol of the native ARS pipeline and of the agent around it (round-2 findings):
ring slot owned by the pipeline, ranks with an empty shard, checkpoints loaded into an agent
that has already run, trajectory recording on the asynchronous path, stale HIP errors."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sw():
    import swimmer_amd
    return swimmer_amd


def _agent(sw, N=6, H=120, seed=9, n=3, **kw):
    ep = sw.EnvParam("LeonSwimmer-Test", n=n, H=H, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
    ap = sw.ARSParam("Test", V1=False, n_iter=4, H=H, N=N, b=N, alpha=0.0075, nu=0.01,
                     safe=False, threshold=0, initial_w="Zero")
    return sw.ARSAgent(ep, ap, seed=seed, device="cuda:0", **kw)


def test_slot_is_the_pipelines_own_count(sw):
    a = _agent(sw)
    pipe = a._pipe
    assert pipe.next_slot() == 0
    for k in range(6):                       # more iterations than slots
        assert pipe.next_slot() == k % pipe.slots
        a.run_iteration_async(want_returns=False)
    a._it = 0                                # the agent's counter no longer matters
    assert pipe.next_slot() == 6 % pipe.slots
    # a call with any other slot is refused, nothing is launched, the count stays
    wrong = (pipe.next_slot() + 1) % pipe.slots
    ap = a.agent_param
    with pytest.raises(sw.SwimmerHipError, match="SW_ERR_SIZE"):
        pipe.rollouts(wrong, a.params, ap.N, a.lo, a.n_local, ap.H, a._deltas_host[wrong],
                      a._deltas2[wrong], a._policy, ap.nu, a._mean, a._inv_std, a._returns_local,
                      a._traj2[wrong], a._moments_local, a._cov_acc, a._status)
    assert pipe.next_slot() == 6 % pipe.slots
    a.runOneIteration()
    torch.cuda.synchronize()


def test_checkpoint_loaded_into_an_agent_that_has_run(sw, tmp_path):
    """load_checkpoint on a used agent: the ring phase is the pipeline's, every queued kernel is
    drained first, and the continuation is bit-identical to the uninterrupted run."""
    a = _agent(sw, seed=9)
    for _ in range(2):
        a.runOneIteration()
    a.save_checkpoint(str(tmp_path / "ck.npz"))
    tail_a = [a.runOneIteration() for _ in range(3)]
    b = _agent(sw, seed=77)
    for _ in range(3):                       # 3 iterations: its ring phase differs from a's
        b.run_iteration_async(want_returns=False)
    b.load_checkpoint(str(tmp_path / "ck.npz"))      # no synchronisation by the caller
    tail_b = [b.runOneIteration() for _ in range(3)]
    assert np.array_equal(np.array(tail_a), np.array(tail_b))
    assert np.array_equal(a.policy, b.policy) and np.array_equal(a.mean, b.mean)
    assert np.array_equal(a.covariance, b.covariance)
    # the same agent again, now with iterations in flight
    for _ in range(5):
        b.run_iteration_async(want_returns=False)
    b.load_checkpoint(str(tmp_path / "ck.npz"))
    tail_c = [b.runOneIteration() for _ in range(3)]
    assert np.array_equal(np.array(tail_a), np.array(tail_c))
    assert np.array_equal(a.covariance, b.covariance)


def test_round1_checkpoint_format_is_converted(sw, tmp_path):
    """Files written before the statistics became {n, mean - c, M2} hold raw sums."""
    a = _agent(sw, seed=5)
    for _ in range(2):
        a.runOneIteration()
    a.save_checkpoint(str(tmp_path / "new.npz"))
    z = dict(np.load(tmp_path / "new.npz"))
    d = a.d
    n, mr, m2 = z["running"][0], z["running"][1:1 + d], z["running"][1 + d:]
    old = dict(z)
    old.pop("format")
    old["running"] = np.concatenate(([n], mr * n, m2 + (mr * n) * mr))    # raw S1, S2 about c
    np.savez(tmp_path / "old.npz", **old)
    b = _agent(sw, seed=1)
    b.load_checkpoint(str(tmp_path / "old.npz"))
    got = b._running.cpu().numpy()
    assert got[0] == n and np.allclose(got[1:1 + d], mr, rtol=1e-14, atol=0)
    assert np.allclose(got[1 + d:], m2, rtol=1e-9, atol=0)


def test_async_iterations_record_matching_policies(sw):
    """run_iteration_async(record_trajectories=True) without runOneIteration: the stored
    policy of rollout 2i / 2i+1 is P_before_update +/- nu * delta_i, and replaying it
    reproduces the stored trajectory."""
    a = _agent(sw, N=4, H=60, seed=3, record_trajectories=True)
    nu = a.agent_param.nu
    pre, deltas = [], []
    rng = np.random.RandomState(3)            # the stream the agent draws from
    for _ in range(3):
        pre.append(a.policy.copy())
        deltas.append(2 * rng.rand(4, a.m, a.d) - 1)
        a.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    pols = np.array(a.database.policies)
    trajs = np.array(a.database.trajectories)
    assert pols.shape == (3 * 8, a.m, a.d) and trajs.shape == (3 * 8, 60, a.d)
    for it in range(3):
        for i in range(4):
            assert np.array_equal(pols[it * 8 + 2 * i], pre[it] + nu * deltas[it][i])
            assert np.array_equal(pols[it * 8 + 2 * i + 1], pre[it] - nu * deltas[it][i])
    assert not np.array_equal(pre[0], pre[2])            # the policy did move in between
    # first iteration ran with mean 0 / cov I: replay two of its stored policies
    env = sw.Environment(a.real_env_param)
    for r in (0, 5):
        _, states = env.rollout(pols[r])
        assert np.abs(np.array(states) - trajs[r]).max() <= 1e-12


def _loaded_hip_runtime():
    """The libamdhip64 this process already has mapped (torch's), not a second copy."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime mapped")


def test_a_stale_hip_error_is_not_blamed_on_our_launch(sw):
    p = sw.SwParams.make(3)
    sw.kernels.reset(p, 8)
    hip = _loaded_hip_runtime()
    assert hip.hipSetDevice(9999) != 0                 # a failed call of "somebody else"
    st = sw.kernels.reset(p, 16)                       # must not report SW_ERR_LAUNCH
    assert st.shape == (8, 16)
    torch.cuda.synchronize()


# ---- ranks with an empty shard (world > N) ------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, H, iters, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swimmer_amd as sw
        a = _agent(sw, N=N, H=H, seed=11)
        rets = [a.runOneIteration() for _ in range(iters)]
        cov = a.reduce_covariance()
        pol, mean = a.policy, a.mean
        try:
            a.run_iteration_async(want_returns=False)
            a.covariance
            raised = False
        except sw.SwimmerHipError:
            raised = True
        out.put((rank, np.array(rets), pol, mean, cov, a.n_local, raised))
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_directions(sw):
    """world = 3, N = 2: rank 2 owns no direction.  It must keep pace with the ring (its H2D of
    the deltas may not overtake the update that still reads the slot) for more iterations than
    the ring has slots, and end with the same policy as everybody else."""
    N, H, world, iters = 2, 100, 3, 7
    ref = _agent(sw, N=N, H=H, seed=11)
    ref_rets = np.array([ref.runOneIteration() for _ in range(iters)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, H, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert [r[5] for r in res] == [1, 1, 0]
    sd = np.sqrt(np.diag(ref.covariance))
    for rank, rets, pol, mean, cov, _, raised in res:
        assert np.array_equal(rets[0], ref_rets[0]), rank
        assert np.allclose(rets, ref_rets, rtol=1e-5, atol=1e-20), rank
        assert np.abs(pol - ref.policy).max() < 1e-8
        assert np.array_equal(pol, res[0][2])               # identical on every rank
        assert (np.abs(cov - ref.covariance) <= 1e-9 * np.outer(sd, sd)).all()
        assert raised      # reading `covariance` after a further iteration without the collective


def test_vec_env_plans_are_per_stream(sw):
    """A pre-bound step launch pins the stream it was made under; stepping the same env under
    another stream must not reuse it (the launch would run on the old stream, unordered with
    the caller's work)."""
    env = sw.VecSwimmerEnv(64, n=3)
    act = torch.full((2, 64), 0.5, dtype=torch.float64, device="cuda:0")
    env.reset()
    env.step(act)
    env.step(act)
    assert len(env._plans) == 2
    side = torch.cuda.Stream("cuda:0")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        env.step(act)
        env.step(act)
        side.synchronize()
    assert len(env._plans) == 4
    streams = {k[3] for k in env._plans}
    assert streams == {torch.cuda.current_stream().cuda_stream, side.cuda_stream}
    ref = sw.VecSwimmerEnv(64, n=3)
    ref.reset()
    for _ in range(4):
        want, _, _, _ = ref.step(act)
    assert torch.equal(env.get_state(), want)


def _store_worker(rank, world, port, path, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swimmer_amd as sw
        out.put((rank, _store_run(sw, path)))
    finally:
        dist.destroy_process_group()


def _store_run(sw, path):
    ep = sw.EnvParam("LeonSwimmer-Test", n=3, H=30, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
    ap = sw.ARSParam("Test", V1=True, n_iter=10, H=30, N=4, b=4, alpha=0.01, nu=0.02,
                     safe=False, threshold=0, initial_w="Zero")
    agent = sw.ARSAgent(ep, ap, seed=6, device="cuda:0", record_trajectories=True)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        curve = agent.runTraining(save_data_path=path)
    return curve


def test_sharded_runs_store_every_rollout(sw, tmp_path):
    """The reference appends EVERY rollout to its Database (ars_agent.py:172) and saves it every
    10 iterations (:213-214).  With two ranks each rank stores its shard in its own file;
    Database.load(path) reads them back, and together they are the single-process store."""
    single = str(tmp_path / "single.npz")
    curve1 = _store_run(sw, single)
    world, shared = 2, str(tmp_path / "sharded.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_store_worker, args=(r, world, port, shared, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for _, curve in res:
        assert np.array_equal(curve, curve1)
    files = sorted(os.listdir(tmp_path))
    assert "sharded.rank000of002.npz" in files and "sharded.rank001of002.npz" in files
    assert "sharded.npz" not in files
    a, b = sw.ars.database.Database(), sw.ars.database.Database()
    a.load(single)
    b.load(shared)
    iters, per_it, per_rank = 11, 8, 4          # runTraining: 1 warm-up + 10 iterations
    assert a.size == b.size == iters * per_it
    ta, tb = np.array(a.trajectories), np.array(b.trajectories)
    pa, pb = np.array(a.policies), np.array(b.policies)
    for it in range(iters):
        for r in range(per_it):
            rank, local = divmod(r, per_rank)
            j = rank * iters * per_rank + it * per_rank + local
            assert np.array_equal(ta[it * per_it + r], tb[j])
            assert np.array_equal(pa[it * per_it + r], pb[j])


def test_issue_probe_reports_plausible_intervals(sw):
    """sw_issue_probe (bench.py's live calibration of the lone-wave issue ceiling): an independent
    f64 FMA and a 32-bit move issue every ~2 ns for a lone wave on MI355X."""
    f64 = sw.kernels.issue_interval_ns(0)
    mov = sw.kernels.issue_interval_ns(1)
    print(f"lone-wave issue intervals: v_fma_f64 {f64:.3f} ns, v_mov_b32 {mov:.3f} ns")
    assert 1.0 < mov < 4.0 and 1.0 < f64 < 5.0 and mov <= f64 * 1.05
    with pytest.raises(sw.SwimmerHipError):
        sw.kernels.issue_interval_ns(7)


def test_estimator_reads_a_store_that_is_still_on_the_gpu(sw):
    """Estimator.I(x) (ars/estimator.py:36-62) over rollouts recorded by the ARS loop: the store
    holds the rollout kernels' device tensors, and the objective is built from them without a host
    round trip -- same value as through the reference-shaped host lists."""
    import copy
    from swimmer_amd.ars.estimator import Estimator
    a = _agent(sw, N=6, H=80, seed=2, record_trajectories=True)
    for _ in range(3):
        a.run_iteration_async(want_returns=False)
    torch.cuda.synchronize()
    db = a.database
    assert db._device_batches and not db._trajectories and db.size == 36
    guess = sw.EnvParam("guess", n=3, H=80, l_i=0.81, m_i=1.19, h=1e-3, k=10.1, epsilon=0.01)
    np.random.seed(1)
    fast = Estimator(db, guess, capacity=20)
    assert fast._batch()[0].shape == (8, 20 * 79)
    assert db._device_batches and not db._trajectories          # nothing was materialised
    host_db = copy.copy(db)
    host_db._device_batches = list(db._device_batches)
    host_db._trajectories, host_db._policies = [], []
    host_db.materialize()
    np.random.seed(1)
    slow = Estimator(host_db, guess, capacity=20)
    assert np.array_equal(fast.subset, slow.subset)
    for x in ([1.19, 0.81, 10.1], [1.2, 0.8, 10.2], [1.0, 1.0, 9.0]):     # unknowns = (m_i, l_i, k)
        assert fast.I(x) == pytest.approx(slow.I(x), rel=1e-12)
    # V2 rollouts were run with whitened actions; the estimator replays V1 actions P s (the
    # reference's estimator.py:52 does the same), so even the true parameters leave a residual
    assert fast.I([1.2, 0.8, 10.2]) >= 0.0
