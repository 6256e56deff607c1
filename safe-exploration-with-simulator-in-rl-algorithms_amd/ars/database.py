"""Trajectory store.

Drop-in for the reference's `Database` (ars/database.py:8-37): the same public attributes
(`policies`, `trajectories`, `size`), the same methods (`load`, `add_trajectory`, `save`) and
the same on-disk format -- one .npz holding the arrays `policies` [R, m, d] and
`trajectories` [R, H, d] (database.py:37).

What differs is where rollouts live before anyone looks at them: the rollout kernel produces
one device tensor [H, d, R] per iteration, and copying 65 MB to the host after every ARS
iteration would cost more than the iteration.  `add_device_batch` therefore only keeps a
reference to the device tensor; the transposition to the reference's per-rollout host lists
happens lazily, when `policies` / `trajectories` are read or the store is saved.
"""
import numpy as np


class Database(object):

    def __init__(self):
        self._policies = []
        self._trajectories = []
        self._device_batches = []   # (traj [H, d, R] device tensor, policies [R, m, d] host array)
        self.size = 0

    # ---- lazily materialised views --------------------------------------------------
    def materialize(self):
        """Move every pending device batch into the host lists (rollout-major, [t][field])."""
        batches, self._device_batches = self._device_batches, []
        for traj, pols in batches:
            per_rollout = traj.permute(2, 0, 1).contiguous().cpu().numpy()
            for r, states in enumerate(per_rollout):
                self._trajectories.append(states.tolist())
                self._policies.append(np.array(pols[r]))
        return self

    @property
    def policies(self):
        return self.materialize()._policies

    @property
    def trajectories(self):
        return self.materialize()._trajectories

    # ---- the reference's interface ----------------------------------------------------
    def add_trajectory(self, trajectory, policy):
        self.materialize()
        self._trajectories.append(trajectory)
        self._policies.append(policy)
        self.size += 1

    def add_device_batch(self, traj, policies):
        """All rollouts of one kernel launch: traj [H, d, R] stays on the GPU."""
        if traj.shape[2] != len(policies):
            raise ValueError("one policy per recorded rollout is required")
        self._device_batches.append((traj, policies))
        self.size += traj.shape[2]

    @staticmethod
    def shard_path(path, rank, world):
        """File holding rank `rank`'s rollouts when `world` ranks share a training run."""
        stem = path[:-4] if path.endswith(".npz") else path
        return f"{stem}.rank{rank:03d}of{world:03d}.npz"

    def save(self, path, rank=0, world=1):
        """The reference's single .npz (database.py:36-37).  With several ranks each rank holds
        the rollouts of its own direction shard (65 MB per iteration and rank at the BASELINE
        sizes -- gathering them would cost more than the iterations): every rank writes
        shard_path(path, rank, world), and load(path) reads the shards back in rank order."""
        if world > 1:
            path = self.shard_path(path, rank, world)
        np.savez(path, policies=self.policies, trajectories=self.trajectories)

    def load(self, path, world=None):
        """The reference's single file (database.py:13-29), or -- when `path` itself does not
        exist -- the per-rank shards save(path, rank, world) wrote.  A shard set must be complete
        and unambiguous: exactly one world size W in the file names (or the `world` given) and
        every rank 0 .. W-1 present; anything else raises instead of silently merging the left-overs
        of another run into the store."""
        import glob
        import os
        import re
        if not os.path.exists(path) and not os.path.exists(path + ".npz"):
            stem = path[:-4] if path.endswith(".npz") else path
            shards = sorted(glob.glob(glob.escape(stem) + ".rank[0-9][0-9][0-9]of[0-9][0-9][0-9].npz"))
            if shards:
                found = {}
                for shard in shards:
                    r, w = (int(v) for v in re.search(r"\.rank(\d{3})of(\d{3})\.npz$", shard).groups())
                    found.setdefault(w, {})[r] = shard
                if world is None:
                    if len(found) != 1:
                        raise ValueError(f"{stem}: shards of several runs (world sizes {sorted(found)}); "
                                         "pass world= to choose one")
                    world = next(iter(found))
                ranks = found.get(int(world), {})
                missing = [r for r in range(int(world)) if r not in ranks]
                if missing or any(r >= int(world) for r in ranks):
                    raise ValueError(f"{stem}: incomplete shard set for world {world}: ranks {missing} missing")
                for r in range(int(world)):
                    self._load_one(ranks[r])
                return
        self._load_one(path if os.path.exists(path) else path + ".npz")

    def _load_one(self, path):
        with np.load(path, allow_pickle=False) as stored:
            missing = {"policies", "trajectories"} - set(stored.files)
            if missing:
                raise ValueError(f"{path}: not a trajectory store (missing {sorted(missing)})")
            pols, trajs = stored["policies"], stored["trajectories"]
        if len(pols) != len(trajs):
            raise ValueError(f"{path}: {len(pols)} policies for {len(trajs)} trajectories")
        for trajectory, policy in zip(trajs, pols):
            self.add_trajectory(trajectory, policy)
