"""Flocking metrics of the reference's data-generation example as device reductions.

`MRSAnalytics(data)` keeps the reference class (examples/simulating_data/helper/MRSAnalytics.py): same constructor
contract (`data.get_episodes()["X"]`: episodes x episode_length x N x D, float32, short episodes padded with NaN,
Trainer.py:186-207) and the same method names and output shapes.  The O(N^2) metrics (separation, cohesion) and the
per-frame reductions (dist_to_leader, vel_stddev) are one fused kernel over all frames (mrs_flock_metrics); the
elementwise ones are plain tensor expressions on the device.  `data` may also be a RolloutLog or the X tensor itself.
"""
import torch

from . import native


class MRSAnalytics:

    def __init__(self, data, device="cuda"):  # MRSAnalytics.py:6-11
        self.data = data
        X = data if isinstance(data, torch.Tensor) else data.get_episodes()["X"]
        self.X = X.to(device=torch.device(device) if not X.is_cuda else X.device, dtype=torch.float32)
        self.num_episodes, self.episode_length, self.N = self.X.shape[0], self.X.shape[1], self.X.shape[2]
        self._m = None

    def _metrics(self):
        if self._m is None:
            self._m = native.flock_metrics(self.X)
        return self._m

    def velocity(self):  # :105-106
        return self.X[:, :, :, 3:]

    def position(self):  # :110-111
        return self.X[:, :, :, :3]

    def vel_leader_alignment(self):  # :18-23
        vel = self.velocity()
        return (vel - vel[:, :, 0, :][:, :, None, :])[:, :, 1:, :].norm(dim=3)

    def vel_leader_alignment_avg(self):
        return torch.mean(self.vel_leader_alignment())

    def vel_mag(self):  # :33-34
        return self.velocity().norm(dim=3)

    def vel_mag_avg(self):
        return self.vel_mag().mean()

    def vel_stddev(self):  # :44-53
        return self._metrics()["vel_stddev"]

    def vel_stddev_avg(self):
        return torch.mean(self.vel_stddev())

    def separation(self):  # :61-72
        return self._metrics()["separation"]

    def separation_avg(self):
        return torch.mean(self.separation())

    def cohesion(self, exclude_leader=False):  # :82-93
        return self._metrics()["cohesion_noleader" if exclude_leader else "cohesion"]

    def cohesion_avg(self):
        return torch.mean(self.cohesion())

    def dist_to_leader(self):  # :95-101
        return self._metrics()["dist_to_leader"]
