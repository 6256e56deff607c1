"""Host-side logic of the mirror classes that needs no GPU."""
import numpy as np
import pytest

import swimmer_amd as sw
from swimmer_amd.ars.database import Database
from swimmer_amd.ars.sharding import shard_bounds, pack_local, exchange
from swimmer_amd.ars.parameters import Threshold


def test_param_dataclasses_keep_reference_field_names():
    ep = sw.EnvParam('LeonSwimmer-RealWorld', n=3, H=1000, l_i=.8, m_i=1.2, h=1e-3, k=10.2,
                     epsilon=0)                       # ars/plot_graph.py:14-16 call shape
    ap = sw.ARSParam('RLControl', V1=True, n_iter=100, H=1000, N=1, b=1, alpha=0.0075, nu=0.01,
                     safe=False, threshold=0, initial_w='Zero')
    assert (ep.n, ep.H, ep.l_i, ep.m_i, ep.k) == (3, 1000, .8, 1.2, 10.2)
    assert (ap.N, ap.b, ap.alpha, ap.nu, ap.initial_w) == (1, 1, 0.0075, 0.01, 'Zero')
    t = Threshold(1.0, 1.0, 0.5)
    assert t.compute_alpha(1) == pytest.approx(1.0)   # K A/(1-B) (1 - B) = 1


def test_spaces_and_registration():
    from swimmer_amd.envs import Box, register_kwargs
    kw = register_kwargs()
    assert kw["id"] == "LeonSwimmer-v0" and kw["max_episode_steps"] == 1000
    assert kw["kwargs"]["n"] == 5                    # envs/gym_swimmer/register.py:9
    b = Box(-5.0, 5.0, (2,))
    assert b.shape == (2,) and b.low == -5.0


def test_database_roundtrip(tmp_path):
    db = Database()
    rng = np.random.default_rng(0)
    for _ in range(3):
        db.add_trajectory(rng.standard_normal((7, 8)).tolist(), rng.standard_normal((2, 8)))
    assert db.size == 3
    db.save(str(tmp_path / "db.npz"))
    z = np.load(tmp_path / "db.npz")
    assert z["policies"].shape == (3, 2, 8) and z["trajectories"].shape == (3, 7, 8)
    db2 = Database()
    db2.load(str(tmp_path / "db.npz"))
    assert db2.size == 3
    assert np.array_equal(np.array(db2.trajectories), z["trajectories"])


def test_sort_directions_matches_reference_rule():
    # max(r+, r-) descending (ars_agent.py:105-108)
    rewards = [1.0, 5.0, 7.0, -1.0, 0.0, 6.0, 2.0, 2.5]
    order = sw.ARSAgent.sort_directions(None, [None] * 4, rewards)
    assert order == [1, 2, 0, 3]


def test_shard_bounds_cover_all_directions():
    for n_dir in (1, 7, 512, 2048, 2050):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi, chunk = shard_bounds(n_dir, r, world)
                assert 0 <= lo <= hi <= n_dir and hi - lo <= chunk
                seen += list(range(lo, hi))
            assert seen == list(range(n_dir))


def test_exchange_single_rank_is_identity():
    import torch
    r = torch.arange(8, dtype=torch.float64)
    m = torch.ones((1, 16), dtype=torch.float64)
    ra, ma = exchange(r, m, 4, 1)
    assert ra is r and ma is m
    buf = pack_local(r[:6], m, 4, 2)
    assert buf.numel() == 8 + 2 * 16 and float(buf[6]) == 0.0 and float(buf[8]) == 1.0


def test_database_refuses_mixed_or_incomplete_shard_sets(tmp_path):
    """Database.load(path) falls back to the per-rank shards of a distributed run; left-overs of
    a run with another world size, or a missing rank, must raise instead of merging silently."""
    stem = str(tmp_path / "store")
    P, T = np.zeros((2, 8)), np.zeros((5, 8))

    def shard(rank, world, n):
        db = Database()
        for i in range(n):
            db.add_trajectory((T + 10 * rank + i).tolist(), P + rank)
        db.save(stem + ".npz", rank=rank, world=world)

    shard(0, 2, 2)
    shard(1, 2, 3)
    db = Database()
    db.load(stem + ".npz")
    assert db.size == 5 and db.policies[2][0, 0] == 1.0          # rank order: 0, 0, 1, 1, 1
    shard(0, 4, 1)                                               # left-over of another run
    with pytest.raises(ValueError, match="several runs"):
        Database().load(stem + ".npz")
    db = Database()
    db.load(stem + ".npz", world=2)                              # explicit choice is honoured
    assert db.size == 5
    with pytest.raises(ValueError, match="incomplete"):
        Database().load(stem + ".npz", world=4)                  # ranks 1..3 of world 4 missing
