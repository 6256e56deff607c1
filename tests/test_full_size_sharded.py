"""BASELINE configs[3] and configs[4] AS STATED: ARS V2 with 2048 directions x 2 rollouts x
H = 1000 sharded over 8 ranks (n = 3 and n = 6), at full size.

The GPU box allows at most 6 processes on its card, so the 8-rank form is exercised in two
complementary ways:
  * 8 LOGICAL ranks in one process through the C ABI: each rank's shard is one
    sw_ars_rollouts_f64 launch writing straight into its segment of the gathered buffer
    (exactly the bytes the all-gather would deliver), then sw_ars_update_gathered_f64 with
    world = 8 -- the kernels, the packed layout and the rank-major merge at 8 x 256 directions;
  * 4 PROCESSES sharing the GPU (gloo, staged through the host) with the real ARSAgent and
    torch.distributed collective at 4 x 512 directions.
The rank-major gather itself at world = 8 / N = 2048 is covered on CPU (tests/test_sharding_gloo.py).
Everything must be bit-identical to the single-process run (shards are aligned to the 16-rollout
moment rows) and within 1e-9 of the oracle's ARS loop."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_TOTAL, H, ITERS, SEED = 2048, 1000, 2, 0
ALPHA, NU = 0.0075, 0.01


def _agent(sw, n, **kw):
    ep = sw.EnvParam("LeonSwimmer-Test", n=n, H=H, l_i=1.0, m_i=1.0, h=1e-3, k=10.0, epsilon=0)
    ap = sw.ARSParam("Test", V1=False, n_iter=ITERS, H=H, N=N_TOTAL, b=N_TOTAL, alpha=ALPHA, nu=NU,
                     safe=False, threshold=0, initial_w="Zero")
    return sw.ARSAgent(ep, ap, seed=SEED, device="cuda:0", **kw)


@pytest.fixture(scope="module")
def single_process():
    """The reference points: one process, all 2048 directions on the one GPU, + the oracle."""
    import swimmer_amd as sw
    from oracle.ars_oracle import ArsOracle
    out = {}
    for n in (3, 6):
        agent = _agent(sw, n)
        o = ArsOracle(n, 1.0, 1.0, 10.0, 1e-3, H, N_TOTAL, N_TOTAL, ALPHA, NU, False, SEED)
        its = []
        for it in range(ITERS):
            r = np.array(agent.runOneIteration())
            ro = np.array(o.iteration())
            dR, dP = np.abs(r - ro).max(), np.abs(agent.policy - o.policy).max()
            print(f"configs[{3 if n == 3 else 4}] n={n} it{it}: max|dR| vs oracle {dR:.3e} "
                  f"(|R| up to {np.abs(ro).max():.2f}), max|dP| {dP:.3e}")
            assert dR <= 1e-9 * max(1.0, np.abs(ro).max())
            assert dP <= 1e-9 and np.abs(agent.mean - o.mean).max() <= 1e-9
            its.append((r, agent.policy, agent.mean, agent._inv_std.cpu().numpy()))
        out[n] = its
        del agent
    return out


@pytest.mark.parametrize("n", [3, 6])
def test_eight_logical_ranks_through_the_c_abi(single_process, n):
    import swimmer_amd as sw
    k = sw.kernels
    world, dev = 8, "cuda:0"
    chunk = N_TOTAL // world
    p = sw.SwParams.make(n)
    m, d = p.m, p.d
    rows_chunk = k.moments_blocks(2 * chunk)
    seg = 2 * chunk + rows_chunk * 2 * d
    f64 = dict(dtype=torch.float64, device=dev)
    policy, mean, inv_std = torch.zeros((m, d), **f64), torch.zeros(d, **f64), torch.ones(d, **f64)
    running = torch.zeros(1 + 2 * d, **f64)
    gathered = torch.zeros(world * seg, **f64)
    status = torch.zeros(2 * chunk, dtype=torch.int32, device=dev)
    rng = np.random.RandomState(SEED)
    for it in range(ITERS):
        deltas = torch.as_tensor(2 * rng.rand(N_TOTAL, m, d) - 1, device=dev)
        for r in range(world):          # rank r's launch; its segment is where the gather puts it
            s = gathered[r * seg:(r + 1) * seg]
            k.ars_rollouts(p, H, policy, deltas, NU, r * chunk, chunk, mean=mean, inv_std=inv_std,
                           returns=s[:2 * chunk], moments=s[2 * chunk:].view(rows_chunk, 2 * d),
                           status=status)
            assert int(status.abs().sum()) == 0
        k.ars_update_gathered(p, N_TOTAL, gathered, world, chunk, rows_chunk, deltas, policy, ALPHA,
                              float(N_TOTAL), 0, running=running, n_new_states=2 * N_TOTAL * H,
                              mean=mean, inv_std=inv_std)
        rets = gathered.view(world, seg)[:, :2 * chunk].reshape(-1).cpu().numpy()
        r1, p1, m1, s1 = single_process[n][it]
        assert np.array_equal(rets, r1)
        assert np.array_equal(policy.cpu().numpy(), p1)
        assert np.array_equal(mean.cpu().numpy(), m1)
        assert np.array_equal(inv_std.cpu().numpy(), s1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swimmer_amd as sw
        agent = _agent(sw, n, full_covariance=False)
        its = []
        for _ in range(ITERS):
            r = np.array(agent.runOneIteration())
            its.append((r, agent.policy, agent.mean, agent._inv_std.cpu().numpy()))
        out.put((rank, its, (agent.lo, agent.hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [3, 6])
def test_four_processes_share_the_gpu(single_process, n):
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    assert [r[2] for r in res] == [(i * 512, (i + 1) * 512) for i in range(world)]
    for rank, its, _ in res:
        for it in range(ITERS):
            for got, want in zip(its[it], single_process[n][it]):
                assert np.array_equal(got, want), (rank, it)
