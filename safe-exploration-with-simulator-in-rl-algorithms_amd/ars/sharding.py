"""Sharding of the ARS direction batch over ranks (one process per GPU, RCCL over xGMI).

The 2N rollouts of an iteration are independent given (policy, mean, cov)
(ars/ars_agent.py:140-172), so rank r runs directions [r*chunk, (r+1)*chunk) and ONE
all-gather per iteration carries both the returns and the per-workgroup V2 moment rows
(a few KB: latency-bound, link bandwidth is irrelevant).  The gather is rank-major, which
preserves the reference's `rewards[2i], rewards[2i+1]` indexing (ars_agent.py:105,122).
Every rank then runs the same deterministic update, so no broadcast is needed.

These helpers are device-agnostic (they only touch torch tensors and torch.distributed) so
the world_size > 1 logic is exercised on CPU with the gloo backend in tests/.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_dir, rank, world):
    """Directions [lo, hi) owned by `rank`; `chunk` is the (padded) per-rank slot count."""
    chunk = -(-n_dir // world)
    lo = min(n_dir, rank * chunk)
    hi = min(n_dir, lo + chunk)
    return lo, hi, chunk


def pack_local(returns_local, moments_local, chunk, rows_chunk):
    """[2*chunk returns | rows_chunk * w moment values], zero padded."""
    w = 0 if moments_local is None else moments_local.shape[1]
    buf = torch.zeros(2 * chunk + rows_chunk * w, dtype=torch.float64,
                      device=returns_local.device)
    buf[:returns_local.numel()] = returns_local
    if moments_local is not None and moments_local.numel():
        buf[2 * chunk:2 * chunk + moments_local.numel()] = moments_local.reshape(-1)
    return buf


def exchange(returns_local, moments_local, n_dir, world, group=None, rows_chunk=0):
    """All-gather one iteration's results.

    returns_local : [2 * n_local]        moments_local : [rows_local, w] or None
    -> (returns_all [2 * n_dir], moments_all [world * rows_chunk, w] or None)
    """
    chunk = -(-n_dir // world)
    if world == 1:
        return returns_local, moments_local
    w = 0 if moments_local is None else moments_local.shape[1]
    buf = pack_local(returns_local, moments_local, chunk, rows_chunk)
    device = buf.device
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        buf = buf.cpu()   # gloo has no GPU all-gather: stage through the host (tests / debugging)
    out = torch.empty((world, buf.numel()), dtype=torch.float64, device=buf.device)
    try:
        dist.all_gather_into_tensor(out, buf, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        out = torch.stack(parts)
    out = out.to(device)
    returns_all = out[:, :2 * chunk].reshape(-1)[:2 * n_dir].contiguous()
    moments_all = None
    if moments_local is not None:
        moments_all = out[:, 2 * chunk:].reshape(world * rows_chunk, w).contiguous()
    return returns_all, moments_all


# experiment knob: issue the collective even for a single rank (measures the framework +
# RCCL launch overhead an iteration pays when world > 1, on a one-GPU box)
import os as _os
_FORCE_COLLECTIVE = bool(_os.environ.get("SWIMMER_FORCE_COLLECTIVE"))


def segment_len(chunk, rows_chunk, width):
    """Doubles in one rank's packed segment [2*chunk returns | rows_chunk x width moments]."""
    return 2 * chunk + rows_chunk * width


def all_gather_segments(send, gathered, world, group=None):
    """gathered[r*L:(r+1)*L] <- rank r's `send` (L doubles).  One collective, no repacking:
    the rollout kernel writes returns and moment rows straight into `send`, and the update
    kernel indexes `gathered` in place (sw_ars_update_gathered_f64)."""
    if world == 1 and not _FORCE_COLLECTIVE:
        return send
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no GPU all-gather: stage through the host (tests / debugging only)
        parts = [torch.empty(send.numel(), dtype=send.dtype) for _ in range(world)]
        dist.all_gather(parts, send.cpu(), group=group)
        gathered.copy_(torch.cat(parts))
        return gathered
    dist.all_gather_into_tensor(gathered, send, group=group)
    return gathered


def returns_from_segments(gathered, n_dir, world, chunk):
    """The [2*n_dir] returns in direction order out of the gathered segments."""
    if world == 1:
        return gathered[:2 * n_dir]
    seg = gathered.numel() // world
    return gathered.view(world, seg)[:, :2 * chunk].reshape(-1)[:2 * n_dir].contiguous()
