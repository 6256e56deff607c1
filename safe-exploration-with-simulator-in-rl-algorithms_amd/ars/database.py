"""Trajectory store: mirror of the reference `Database` (ars/database.py:8-37).

Same attributes (`policies`, `trajectories`, `size`) and the same .npz format
(`policies`, `trajectories` arrays; np.savez, database.py:37).  Trajectories produced by
the rollout kernel arrive as one device tensor [H, d, n_roll] per iteration; they are kept
on the GPU and only transposed to the reference's [rollout][t][d] order when read.
"""
import numpy as np


class Database(object):

    def __init__(self):
        self.policies = []
        self.trajectories = []
        self.size = 0
        self._pending = []  # (traj [H, d, R] device tensor, policies [R, m, d] host array)

    def _flush(self):
        for traj, pols in self._pending:
            host = traj.permute(2, 0, 1).contiguous().cpu().numpy()  # [R, H, d]
            for r in range(host.shape[0]):
                self.trajectories.append(host[r].tolist())
                self.policies.append(np.array(pols[r]))
        self._pending = []

    def load(self, path):
        npzfile = np.load(path)
        assert ('policies' in npzfile.files and 'trajectories' in npzfile.files), \
            "The file loaded doesn't contain the array 'policies' and 'trajectories'"
        policies = npzfile['policies']
        trajectories = npzfile['trajectories']
        assert (len(policies) == len(trajectories)), \
            "'policies' and 'trajectories' doesn't have the same length"
        for policy, trajectory in zip(policies, trajectories):
            self.add_trajectory(trajectory, policy)

    def add_trajectory(self, trajectory, policy):
        self._flush()
        self.trajectories.append(trajectory)
        self.policies.append(policy)
        self.size += 1

    def add_device_batch(self, traj, policies):
        """traj [H, d, R] device tensor (kept as is), policies [R, m, d] host array."""
        self._pending.append((traj, policies))
        self.size += traj.shape[2]

    def materialize(self):
        """Bring every pending device batch to the reference's host lists."""
        self._flush()
        return self

    def save(self, path):
        self._flush()
        np.savez(path, policies=self.policies, trajectories=self.trajectories)
