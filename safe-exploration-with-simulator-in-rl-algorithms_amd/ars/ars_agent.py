"""Augmented Random Search driver: mirror of the reference `ARSAgent`
(ars/ars_agent.py:16-220, the unsafe V1/V2 path).

Same constructor arguments, same public methods (runOneIteration, runTraining,
sort_directions, update_policy) and attributes (policy, mean, covariance, agent_param,
database), same random stream (NumPy's global legacy generator, seeded in __init__, deltas
drawn as 2*rand(m, d)-1, ars_agent.py:95,137).  What differs is where the work happens:

  * the N perturbations and the 2N rollouts of an iteration are ONE launch of the fused
    rollout kernel (sw_ars_rollouts_f64) on this rank's shard of the directions;
  * returns and V2 moment rows are exchanged with ONE all-gather (RCCL) per iteration;
  * sigma_R, the policy step and the running V2 statistics are ONE launch of
    sw_ars_update_f64, run redundantly and deterministically on every rank;
  * the full state covariance (only its diagonal feeds the policy, ars/environment.py:32)
    is one HBM-bound pass over the recorded trajectories (sw_traj_moments_f64) that rides
    along in the next iteration's rollout launch.  With more than one rank every rank holds
    the sums of its own shard: `reduce_covariance()` is the (explicit) collective that adds
    them up; the `covariance` attribute itself never communicates.

The safe-exploration gate (ars_agent.py:144-157) is sequential by construction and is not
part of this path: agent_param.safe=True raises NotImplementedError.

Streams: every launch goes to torch's CURRENT stream.  With a NCCL process group alive, work on
the null (default) stream is implicitly ordered against the group's streams, which costs an
iteration ~13 us of device-side waits (measured); run distributed training under a stream of
its own, e.g. torch.cuda.set_stream(torch.cuda.Stream()), as bench.py does.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import kernels
from .._lib import (SwParams, SwimmerHipError, kernel_flags, numpy_global_uniform_pm1,
                    require_gpu)
from .database import Database
from .environment import Environment
from . import sharding as _sh
from .sharding import (all_gather_segments, returns_from_segments, segment_len,
                       shard_bounds)


class ARSAgent(object):

    def __init__(self, real_env_param, agent_param, data_path=None, seed=None,
                 guess_param=None, approx_error=None, sim_thresh=None, *, device=None,
                 process_group=None, record_trajectories=False, full_covariance=True,
                 top_b=0, rollout_kernel="auto", direct_rccl=None):
        if agent_param.safe:
            raise NotImplementedError(
                "safe exploration (ars_agent.py:144-157) gates every real rollout on a "
                "simulator rollout, one at a time; it is outside the data-parallel path")
        require_gpu()
        self.distributed = dist.is_available() and dist.is_initialized()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        if device is None:
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        self.device = torch.device(device)

        self.real_env_param = real_env_param
        self.real_world = Environment(real_env_param, device=device)
        self.agent_param = agent_param
        self.database = Database()
        self.record_trajectories = record_trajectories
        self.full_covariance = full_covariance and not agent_param.V1
        self.top_b = int(top_b)

        n = real_env_param.n
        self.m, self.d = n - 1, 2 * n + 2
        # direction / max_u keep the env defaults, as in the reference (environment.py:15-17)
        self.params = SwParams.make(n, real_env_param.l_i, real_env_param.m_i,
                                    real_env_param.k, real_env_param.h, (1.0, 0.0),
                                    flags=kernel_flags(rollout_kernel))
        if agent_param.initial_w == 'Zero':
            policy = np.zeros((self.m, self.d))
        else:
            policy = np.load(agent_param.initial_w)
            assert policy.shape == (self.m, self.d)

        f64 = dict(dtype=torch.float64, device=self.device)
        self._policy = torch.as_tensor(np.ascontiguousarray(policy, dtype=np.float64),
                                       device=self.device)
        self.v2 = not agent_param.V1
        self._mean = torch.zeros(self.d, **f64) if self.v2 else None
        self._inv_std = torch.ones(self.d, **f64) if self.v2 else None
        self._running = torch.zeros(1 + 2 * self.d, **f64) if self.v2 else None
        self._sigma = torch.zeros(1, **f64)
        self.n_saved_states = 0

        N, H = agent_param.N, agent_param.H
        self.lo, self.hi, self.chunk = shard_bounds(N, self.rank, self.world)
        self.n_local = self.hi - self.lo
        # covariance sums of this rank's shard [count | sum x | sum x x^T] + the pass's scratch
        self._cov_sums = 1 + self.d + self.d * self.d
        self._cov_acc = (kernels.new_cov_acc(self.params, 2 * self.n_local, H, self.device)
                         if self.full_covariance else None)
        self._coll_events = None
        self._cov_total = None       # all ranks' sums as of iteration _cov_total_it (world > 1)
        self._cov_total_it = -1
        self.rows_chunk = kernels.moments_blocks(2 * self.chunk) if self.v2 else 0
        # Ring-buffered pipeline (slot = iteration mod ring depth), enqueued from native code
        # (sw_ars_pipeline, include/swimmer_hip.h):
        #   copy stream : H2D of the deltas, overlaps the tail of the previous iteration
        #   main stream : rollouts -> all-gather -> update   (the critical path, kernels only);
        #                 the full-covariance pass over iteration i's trajectories rides along
        #                 in the rollout launch of iteration i + 1 (only diag(cov) feeds the
        #                 policy, and that comes from the moments fused into the rollout kernel)
        self._pipe = kernels.ArsPipeline()
        ring = self._pipe.slots
        self._deltas2 = [torch.empty((N, self.m, self.d), **f64) for _ in range(ring)]
        self._deltas = self._deltas2[0]
        self._deltas_host = [torch.empty((N, self.m, self.d), dtype=torch.float64).pin_memory()
                             for _ in range(ring)]
        self._deltas_host_np = [t.numpy() for t in self._deltas_host]
        self._it = 0
        # This rank's results live in ONE packed segment [2*chunk returns | moment rows]
        # (zero padded): the kernels write into views of it, the all-gather ships it as is,
        # and the update kernel indexes the gathered buffer in place.
        width = 2 * self.d
        self._seg_len = segment_len(self.chunk, self.rows_chunk, width)
        self._send = torch.zeros(self._seg_len, **f64)
        self._gathered = (torch.zeros(self.world * self._seg_len, **f64)
                          if (self.world > 1 or _sh._FORCE_COLLECTIVE) else self._send)
        self._returns_local = self._send[:2 * self.n_local]
        rows_local = kernels.moments_blocks(2 * self.n_local) if self.v2 else 0
        self._moments_local = (self._send[2 * self.chunk:2 * self.chunk + rows_local * width]
                               .view(rows_local, width) if self.v2 else None)
        need_traj = self.full_covariance or record_trajectories
        self._traj2 = ([torch.empty((H, self.d, 2 * self.n_local), **f64) for _ in range(ring)]
                       if need_traj and self.n_local > 0 else [None] * ring)
        self._traj = self._traj2[0]
        self._status = torch.zeros(max(1, 2 * self.n_local), dtype=torch.int32,
                                   device=self.device)

        # The exchange: torch.distributed (default), or -- direct_rccl=True / SWIMMER_DIRECT_RCCL=1 --
        # ncclAllGather called from native code on the critical stream with a communicator of this
        # agent's own (sw_comm, include/swimmer_hip.h).  Same bytes either way; torch stays the
        # default until the direct path has met more than one rank.
        if direct_rccl is None:
            direct_rccl = os.environ.get("SWIMMER_DIRECT_RCCL", "0") not in ("", "0")
        self._comm = None
        if direct_rccl and (self.world > 1 or _sh._FORCE_COLLECTIVE):
            if self.world > 1 and dist.get_backend(self.group) != "nccl":
                raise SwimmerHipError("direct_rccl needs GPU ranks (backend nccl), one per device")
            # the communicator binds to the CURRENT device: make it this agent's (a torchrun job that never
            # called torch.cuda.set_device would otherwise put every rank's communicator on GPU 0)
            with torch.cuda.device(self.device):
                self._comm = kernels.DirectComm(self.world, self.rank, self._broadcast_bytes, self._all_ranks_ok)

        # Randomness: the reference seeds NumPy's global generator (ars_agent.py:94-95)
        self.n_seed = seed
        np.random.seed(self.n_seed)

    def _broadcast_bytes(self, payload):
        """Rank 0's 128 id bytes to every rank of the group (COLLECTIVE)."""
        if self.world == 1:
            return payload
        t = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src, group=self.group)
        return bytes(t.cpu().numpy().tobytes())

    def _all_ranks_ok(self, ok):
        """COLLECTIVE: True when `ok` holds on every rank of the group."""
        if self.world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _exchange(self):
        if self._comm is not None:
            return self._comm.all_gather(self._send, self._gathered)
        return all_gather_segments(self._send, self._gathered, self.world, self.group)

    # ---- attributes the reference exposes -------------------------------------------
    @property
    def policy(self):
        return self._policy.cpu().numpy()

    @policy.setter
    def policy(self, value):
        self._policy.copy_(torch.as_tensor(np.ascontiguousarray(value, dtype=np.float64)))

    @property
    def mean(self):
        return None if not self.v2 else self._mean.cpu().numpy()

    @property
    def covariance(self):
        """np.cov(all saved states) (ddof = 1, ars_agent.py:182).  The diagonal is the one
        the policy whitening uses; off-diagonals need full_covariance=True.

        Never communicates.  With one rank it is always current.  With several ranks the
        off-diagonal sums are spread over the ranks: call `reduce_covariance()` on EVERY rank
        (a collective) after the iteration whose value is wanted; reading the attribute at any
        other time raises instead of blocking in a collective the other ranks never enter."""
        if not self.v2:
            return None
        if self.n_saved_states == 0:
            return np.identity(self.d)
        if not self.full_covariance:
            return np.diag(self._inv_std.cpu().numpy() ** -2.0)
        if self.world == 1:
            self._pipe.sync_cov()
            return self._cov_matrix(self._cov_acc[:self._cov_sums].cpu().numpy())
        if self._cov_total_it != self._it:
            raise SwimmerHipError(
                "covariance: the off-diagonal sums live on all ranks; call reduce_covariance() "
                "on every rank after the iteration (collective) before reading the attribute")
        return self._cov_matrix(self._cov_total)

    def reduce_covariance(self):
        """COLLECTIVE: add up every rank's covariance sums (they are linear) and return the
        covariance matrix; `covariance` then returns the same matrix on every rank until
        the next iteration.  A no-op beyond the local flush with one rank."""
        if not (self.v2 and self.full_covariance) or self.n_saved_states == 0:
            return self.covariance
        self._pipe.sync_cov()
        acc = self._cov_acc[:self._cov_sums]
        if self.world > 1:
            acc = acc.cpu() if dist.get_backend(self.group) == "gloo" else acc.clone()
            dist.all_reduce(acc, group=self.group)
        self._cov_total = acc.cpu().numpy()
        self._cov_total_it = self._it
        return self._cov_matrix(self._cov_total)

    def _cov_matrix(self, acc):
        n, s1 = acc[0], acc[1:1 + self.d]
        s2 = acc[1 + self.d:].reshape(self.d, self.d)
        cov = (s2 - np.outer(s1, s1) / n) / (n - 1.0)
        # the diagonal is the one the policy uses: the V2 running statistics
        cov[np.diag_indices(self.d)] = self._inv_std.cpu().numpy() ** -2.0
        return cov

    # ---- measurement aid: device time of the per-iteration all-gather ------------------
    def collective_timing(self, on):
        """HIP events on the launch stream around every all-gather from now on (off: None).
        Each pair costs a few microseconds of pipeline bubbles: keep it out of timed loops."""
        self._coll_events = [] if on else None

    def collective_us(self):
        """Mean device time between the end of the rollout launch and the start of the update,
        i.e. the all-gather as the critical stream sees it; None without samples."""
        if not self._coll_events:
            return None
        torch.cuda.synchronize(self.device)
        return 1e3 * sum(a.elapsed_time(b) for a, b in self._coll_events) / len(self._coll_events)

    # ---- pieces of an iteration -----------------------------------------------------
    def sample_deltas(self):
        """N perturbations, uniform on (-1, 1): the reference's exact call sequence
        (N draws of rand(m, d) from the global generator consume the stream exactly like
        one rand(N, m, d))."""
        return 2 * np.random.rand(self.agent_param.N, self.m, self.d) - 1

    def sort_directions(self, deltas, rewards):
        """Directions sorted by max(r+, r-), best first (ars_agent.py:97-108)."""
        r = np.asarray(rewards, dtype=np.float64).reshape(-1, 2)
        return np.argsort(r.max(axis=1)).tolist()[::-1]

    def update_policy(self, deltas, rewards, order):
        """P += alpha / (b sigma_R) sum_{i in order} (r_i+ - r_i-) delta_i
        (ars_agent.py:110-130), on the GPU.  `order` selects the directions used."""
        order = list(order)
        d = kernels._lib.dev_f64(np.asarray(deltas, dtype=np.float64)[order], self.device)
        r = np.asarray(rewards, dtype=np.float64).reshape(-1, 2)[order].reshape(-1)
        r = kernels._lib.dev_f64(r, self.device)
        kernels.ars_update(self.params, r, d, self._policy, self.agent_param.alpha,
                           self.agent_param.b, 0, sigma_out=self._sigma)

    def run_iteration_async(self, deltas=None, want_returns=True):
        """One ARS iteration without synchronising the host; returns the [2N] returns as a
        device tensor (a view of the result segment, valid until the next iteration), or
        None with want_returns=False."""
        ap = self.agent_param
        i = self._pipe.next_slot()               # the pipeline's own count, not self._it
        self._it += 1
        self._pipe.host_slot_wait(i)             # the slot's previous user is done with it
        if self.record_trajectories:
            # the policy the rollouts are about to run with (the update below overwrites it)
            self._policy_snapshot = self._policy.cpu().numpy()
        host = self._deltas_host_np[i]
        if deltas is None:      # the same draws as sample_deltas(), generated natively
            numpy_global_uniform_pm1(host)
        else:
            host[...] = deltas
        self._deltas, self._traj = self._deltas2[i], self._traj2[i]
        self._pipe.rollouts(i, self.params, ap.N, self.lo, self.n_local, ap.H,
                            self._deltas_host[i], self._deltas, self._policy, ap.nu,
                            self._mean, self._inv_std, self._returns_local, self._traj,
                            self._moments_local,
                            self._cov_acc if self._traj is not None else None, self._status)
        if self._coll_events is not None and (self.world > 1 or _sh._FORCE_COLLECTIVE):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            gathered = self._exchange()
            e1.record()
            self._coll_events.append((e0, e1))
        else:
            gathered = self._exchange()
        n_new = 2 * ap.N * ap.H
        self._pipe.update(i, self.params, ap.N, gathered, self.world, self.chunk,
                          self.rows_chunk, self._deltas, self._policy, ap.alpha, ap.b,
                          self.top_b, self._running, n_new, self._mean, self._inv_std,
                          self._sigma)
        if self.v2:
            self.n_saved_states += n_new
        if self.record_trajectories and self._traj is not None:
            self.database.add_device_batch(self._traj.clone(),
                                           self._rollout_policies(host.copy()))
        if not want_returns:
            return None
        return returns_from_segments(gathered, ap.N, self.world, self.chunk)

    def _rollout_policies(self, deltas):
        """Host copy of the 2*n_local perturbed policies of this rank's shard, in rollout
        order (what the reference stores next to each trajectory, ars_agent.py:172)."""
        P = self._policy_snapshot
        out = np.empty((2 * self.n_local, self.m, self.d))
        nd = self.agent_param.nu * np.asarray(deltas)[self.lo:self.hi]
        out[0::2] = P + nd
        out[1::2] = P - nd
        return out

    # ---- checkpoint / resume (the reference only saves the policy, ars_agent.py:218-219) ----
    CHECKPOINT_FORMAT = 2   # 2: running = {n, mean - c, M2};  1: raw sums {n, S1, S2} (converted)

    def save_checkpoint(self, path):
        """Everything needed to continue training bit for bit: policy, V2 running statistics,
        covariance sums (per rank when distributed), iteration count and NumPy's global generator
        state.  Plain arrays in an .npz (no pickle).  Collective when distributed; rank 0 writes.
        Policy, returns and V2 statistics continue bit for bit at any world size whose shards are
        aligned to the 16-rollout moment rows; the full covariance continues bit for bit when
        the world size is the one that saved the file (else equal up to summation order)."""
        torch.cuda.synchronize(self.device)
        kind, key, pos, has_gauss, cached = np.random.get_state()
        assert kind == "MT19937"
        cov = cov_ranks = None
        if self.full_covariance:
            self._pipe.sync_cov()
            cov = self._cov_acc[:self._cov_sums]
            if self.world > 1:
                # every rank's own sums are kept (cov_acc_ranks): restoring them rank by rank keeps
                # the summation order, so a resume at the same world size is bit-exact for the
                # covariance too; cov_acc (their total) serves a resume at another world size
                cov = cov.cpu() if dist.get_backend(self.group) == "gloo" else cov.clone()
                parts = [torch.empty_like(cov) for _ in range(self.world)]
                dist.all_gather(parts, cov, group=self.group)
                cov_ranks = torch.stack(parts).cpu().numpy()
                cov = cov_ranks[0].copy()
                for r in range(1, self.world):
                    cov += cov_ranks[r]
            else:
                cov = cov.cpu().numpy()
        if self.rank == 0:
            data = dict(policy=self.policy, n_saved_states=np.int64(self.n_saved_states),
                        iteration=np.int64(self._it), rng_key=key, rng_pos=np.int64(pos),
                        rng_has_gauss=np.int64(has_gauss), rng_cached=np.float64(cached),
                        v2=np.int64(self.v2), format=np.int64(self.CHECKPOINT_FORMAT))
            if self.v2:
                data.update(mean=self._mean.cpu().numpy(), inv_std=self._inv_std.cpu().numpy(),
                            running=self._running.cpu().numpy())
            if cov is not None:
                data["cov_acc"] = cov
            if cov_ranks is not None:
                data["cov_acc_ranks"] = cov_ranks
            with open(path, "wb") as f:
                np.savez(f, **data)

    def load_checkpoint(self, path):
        z = np.load(path, allow_pickle=False)
        if bool(z["v2"]) != self.v2 or z["policy"].shape != (self.m, self.d):
            raise SwimmerHipError("checkpoint does not match this agent's configuration")
        # nothing of this agent's earlier iterations may still be in flight: the buffers below
        # are read by queued kernels (the update of the last iteration, an owed covariance pass)
        if self.full_covariance:
            self._pipe.sync_cov()
        torch.cuda.synchronize(self.device)
        dev = self.device
        self._policy.copy_(torch.as_tensor(z["policy"], device=dev))
        if self.v2:
            running = np.array(z["running"], dtype=np.float64)
            if "format" not in z.files or int(z["format"]) < 2:
                # round-1 files hold raw sums about the pivot: {n, S1, S2} -> {n, S1/n, S2 - S1^2/n}
                n, s1, s2 = running[0], running[1:1 + self.d], running[1 + self.d:]
                if n > 0:
                    running = np.concatenate(([n], s1 / n, s2 - s1 * (s1 / n)))
            self._mean.copy_(torch.as_tensor(z["mean"], device=dev))
            self._inv_std.copy_(torch.as_tensor(z["inv_std"], device=dev))
            self._running.copy_(torch.as_tensor(running, device=dev))
        if self.full_covariance:
            self._cov_acc.zero_()
            if "cov_acc_ranks" in z.files and z["cov_acc_ranks"].shape[0] == self.world:
                # same world size as the run that saved it: every rank takes its own sums back
                self._cov_acc[:self._cov_sums].copy_(torch.as_tensor(z["cov_acc_ranks"][self.rank], device=dev))
            elif "cov_acc" in z.files and self.rank == 0:
                # another world size (or a one-rank file): the total goes to rank 0 -- the sums are
                # linear, so the covariance is the same up to the order of the additions
                self._cov_acc[:self._cov_sums].copy_(torch.as_tensor(z["cov_acc"], device=dev))
        self._cov_total_it = -1
        self.n_saved_states = int(z["n_saved_states"])
        self._it = int(z["iteration"])
        np.random.set_state(("MT19937", z["rng_key"], int(z["rng_pos"]), int(z["rng_has_gauss"]),
                             float(z["rng_cached"])))
        torch.cuda.synchronize(self.device)

    def runOneIteration(self):
        """One whole ARS iteration (ars_agent.py:132-185); returns the list of 2N returns."""
        rets = self.run_iteration_async()
        out = rets.cpu().numpy()
        if int((self._status != 0).sum().item()):
            raise np.linalg.LinAlgError("Singular matrix / non-finite state in a rollout")
        return out.tolist()

    def runTraining(self, save_data_path=None, save_policy_path=None):
        """1 warm-up iteration + n_iter iterations; curve = mean of the 2N returns
        (ars_agent.py:187-220)."""
        rewards = [np.mean(self.runOneIteration())]
        for j in range(1, self.agent_param.n_iter + 1):
            all_rewards = self.runOneIteration()
            r = np.mean(all_rewards) if len(all_rewards) > 0 else rewards[-1]
            rewards.append(r)
            if j % 10 == 0:
                if self.rank == 0:
                    ap, ep = self.agent_param, self.real_env_param
                    variant = "V1" if ap.V1 else "V2"
                    print(f"[seed {self.n_seed}] ARS {variant} n={ep.n} N={ap.N} b={ap.b} "
                          f"alpha={ap.alpha} nu={ap.nu} h={ep.h} l_i={ep.l_i} m_i={ep.m_i} | "
                          f"iteration {j}/{ap.n_iter}: mean return {r}")
                if save_data_path is not None:
                    # one rank: the reference's single file; several ranks: every rank stores the
                    # rollouts of its own shard (Database.shard_path / Database.load reads them all)
                    self.database.save(save_data_path, rank=self.rank, world=self.world)
        self.real_world.close()
        if save_policy_path is not None and self.rank == 0:
            np.save(save_policy_path, self.policy)
        return np.array(rewards)
