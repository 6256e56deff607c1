"""End-to-end check of the sharded ARS path with TWO ranks sharing the one GPU of the test
box (gloo for the exchange, staged through the host; RCCL needs one GPU per rank).  Every
rank must hold the same policy / statistics as a single-process run on the same seed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_agent(sw, N, H, V1, seed, **kw):
    ep = sw.EnvParam("LeonSwimmer-Test", n=3, H=H, l_i=0.8, m_i=1.2, h=1e-3, k=10.2, epsilon=0)
    ap = sw.ARSParam("Test", V1=V1, n_iter=3, H=H, N=N, b=N, alpha=0.0075, nu=0.01, safe=False,
                     threshold=0, initial_w="Zero")
    return sw.ARSAgent(ep, ap, seed=seed, device="cuda:0", **kw)


def _worker(rank, world, port, N, H, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swimmer_amd as sw
        agent = _make_agent(sw, N, H, False, 11)
        rets = [agent.runOneIteration() for _ in range(3)]
        out.put((rank, np.array(rets), agent.policy, agent.mean, agent.reduce_covariance(),
                 (agent.lo, agent.hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world", [(16, 2), (12, 2), (32, 4)])   # 12: padded, unaligned shards
def test_ranks_match_single_process(N, world):
    import swimmer_amd as sw
    H = 120
    ref = _make_agent(sw, N, H, False, 11)
    ref_rets = np.array([ref.runOneIteration() for _ in range(3)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    shards = [r[5] for r in res]
    assert shards[0][0] == 0 and shards[-1][1] == N
    assert all(shards[i][1] == shards[i + 1][0] for i in range(world - 1))
    for rank, rets, pol, mean, cov, _ in res:
        if (N // world) % 8 == 0:
            # shards aligned to the 16-rollout moment rows: same kernels, same inputs, same
            # summation order -> bit-identical to the single-process run
            assert np.array_equal(rets, ref_rets), rank
            assert np.array_equal(pol, ref.policy) and np.array_equal(mean, ref.mean)
        else:
            # unaligned shards group the V2 sums differently (last-bit differences in the
            # statistics, amplified by the whitening of these short, nearly symmetric
            # rollouts whose returns are ~1e-10)
            assert np.array_equal(rets[0], ref_rets[0]), rank     # first iteration: mean 0, cov I
            assert np.allclose(rets, ref_rets, rtol=1e-5, atol=1e-20), rank
            assert np.abs(pol - ref.policy).max() < 1e-8
            assert np.abs(mean - ref.mean).max() < 1e-10
        sd = np.sqrt(np.diag(ref.covariance))
        assert (np.abs(cov - ref.covariance) <= 1e-9 * np.outer(sd, sd)).all()
    for other in res[1:]:
        assert np.array_equal(res[0][2], other[2])   # every rank holds the same policy


def _ckpt_worker(rank, world, port, N, H, path, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swimmer_amd as sw
        a = _make_agent(sw, N, H, False, 5)
        for _ in range(2):
            a.runOneIteration()
        a.save_checkpoint(path)                     # collective; rank 0 writes
        dist.barrier()
        tail_a = [a.runOneIteration() for _ in range(2)]
        cov_a = a.reduce_covariance()
        b = _make_agent(sw, N, H, False, 999)       # another seed: the checkpoint restores the stream
        b.load_checkpoint(path)
        tail_b = [b.runOneIteration() for _ in range(2)]
        cov_b = b.reduce_covariance()
        out.put((rank, np.array(tail_a), np.array(tail_b), a.policy, b.policy, cov_a, cov_b,
                 b._cov_acc[:b._cov_sums].cpu().numpy(), a._cov_acc[:a._cov_sums].cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_checkpoint_resume_with_two_ranks_is_bit_exact(tmp_path):
    """A checkpoint written by a two-rank run keeps EVERY rank's covariance sums (cov_acc_ranks): a
    resume at the same world size restores them rank by rank, so returns, policy AND the full covariance
    continue bit for bit (with the total on rank 0 only the covariance would differ by summation order)."""
    N, H, world = 16, 100, 2
    path = str(tmp_path / "ck2.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, N, H, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    z = np.load(path, allow_pickle=False)
    assert z["cov_acc_ranks"].shape[0] == world
    assert np.array_equal(z["cov_acc_ranks"].sum(0), z["cov_acc"])
    for rank, tail_a, tail_b, pol_a, pol_b, cov_a, cov_b, acc_b, acc_a in res:
        assert np.array_equal(tail_a, tail_b), rank
        assert np.array_equal(pol_a, pol_b), rank
        assert np.array_equal(acc_a, acc_b), rank          # this rank's own sums, bit for bit
        assert np.array_equal(cov_a, cov_b), rank
