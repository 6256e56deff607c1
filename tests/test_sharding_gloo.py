"""world_size = 2 (and 3, uneven) exchange of returns + moment rows over gloo on CPU: the
N > 1 path of ARSAgent.run_iteration_async without the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from swimmer_amd.ars.sharding import exchange, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_return(i, sign):
    return 1000.0 * i + (0.25 if sign > 0 else -0.5)


def _worker(rank, world, port, n_dir, w, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi, chunk = shard_bounds(n_dir, rank, world)
        rows_chunk = -(-2 * chunk // 16)
        rl = torch.tensor([_fake_return(i, s) for i in range(lo, hi) for s in (1, -1)],
                          dtype=torch.float64)
        rows_local = -(-2 * (hi - lo) // 16) if hi > lo else 0
        ml = torch.full((rows_local, w), float(rank + 1), dtype=torch.float64)
        ra, ma = exchange(rl, ml, n_dir, world, None, rows_chunk)
        expect = torch.tensor([_fake_return(i, s) for i in range(n_dir) for s in (1, -1)],
                              dtype=torch.float64)
        ok = torch.equal(ra, expect) and ma.shape == (world * rows_chunk, w)
        # the update sums the rows: every rank must see the same total
        tot = ma.sum(0)
        allr = [-(-2 * (shard_bounds(n_dir, r, world)[1] - shard_bounds(n_dir, r, world)[0]) // 16)
                * (r + 1.0) for r in range(world)]
        ok = ok and bool(torch.all(tot == sum(allr)))
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_dir", [(2, 64), (2, 7), (3, 100)])
def test_exchange_over_gloo(world, n_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_dir, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def _seg_worker(rank, world, port, n_dir, width, out):
    from swimmer_amd.ars.sharding import (all_gather_segments, returns_from_segments,
                                          segment_len)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi, chunk = shard_bounds(n_dir, rank, world)
        rows_chunk = -(-2 * chunk // 16)
        L = segment_len(chunk, rows_chunk, width)
        send = torch.zeros(L, dtype=torch.float64)
        send[:2 * (hi - lo)] = torch.tensor([_fake_return(i, s) for i in range(lo, hi) for s in (1, -1)],
                                            dtype=torch.float64)
        rows_local = -(-2 * (hi - lo) // 16) if hi > lo else 0
        send[2 * chunk:2 * chunk + rows_local * width] = float(rank + 1)
        gathered = torch.zeros(world * L, dtype=torch.float64)
        g = all_gather_segments(send, gathered, world)
        ra = returns_from_segments(g, n_dir, world, chunk)
        expect = torch.tensor([_fake_return(i, s) for i in range(n_dir) for s in (1, -1)],
                              dtype=torch.float64)
        ok = torch.equal(ra, expect)
        # the layout the update kernel indexes: segment r at r*L, moments behind 2*chunk returns
        for r in range(world):
            rl, rh, _ = shard_bounds(n_dir, r, world)
            rows_r = -(-2 * (rh - rl) // 16) if rh > rl else 0
            seg = g[r * L:(r + 1) * L]
            ok = ok and bool((seg[2 * chunk:2 * chunk + rows_r * width] == r + 1.0).all())
            ok = ok and bool((seg[2 * chunk + rows_r * width:] == 0.0).all())
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_dir", [(2, 64), (2, 9), (3, 20), (8, 2048)])   # last: configs[3] / [4]
def test_packed_segments_over_gloo(world, n_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, world, port, n_dir, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]
