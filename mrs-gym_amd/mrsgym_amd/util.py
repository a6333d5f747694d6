"""The few helpers of mrsgym/Util.py that sit on the reset path:
totensor (Util.py:27-31), randrange (:18-24), CombinedDistribution (:86-132), SphereTransform (:176-195).
Keyboard / camera / GUI helpers are out of scope."""
import torch
from torch.distributions import Distribution, constraints
from torch.distributions.transforms import Transform


def totensor(x):
    return x if isinstance(x, torch.Tensor) else torch.tensor(x)


def randrange(low, high):
    low, high = totensor(low), totensor(high)
    return torch.rand(low.shape) * (high - low) + low


class CombinedDistribution(Distribution):
    """Several distributions sampled together and stacked / concatenated along `dim`."""
    arg_constraints = {}

    def __init__(self, dist, mixer='stack', dim=0):
        super().__init__(validate_args=False)
        self.dist, self.mixer, self.dim = list(dist), mixer, dim

    def _mix(self, parts, lead=0):
        # `dim` counts the axes of ONE sample; with sample_shape in front it moves right by that many axes
        dim = self.dim + lead if self.dim >= 0 else self.dim
        return (torch.stack if self.mixer == 'stack' else torch.cat)(parts, dim=dim)

    def sample(self, sample_shape=torch.Size()):
        sample_shape = torch.Size(sample_shape)
        return self._mix([d.sample(sample_shape) for d in self.dist], len(sample_shape))

    def rsample(self, sample_shape=torch.Size()):
        sample_shape = torch.Size(sample_shape)
        return self._mix([d.rsample(sample_shape) for d in self.dist], len(sample_shape))

    def expand(self, *a, **k):
        for d in self.dist:
            d.expand(*a, **k)

    @property
    def batch_shape(self):
        return self.sample().shape

    @property
    def mean(self):
        return self._mix([d.mean for d in self.dist])

    @property
    def variance(self):
        return self._mix([d.variance for d in self.dist])

    @property
    def stddev(self):
        return self._mix([d.stddev for d in self.dist])

    def __getitem__(self, i):
        return self.dist[i]

    def __setitem__(self, i, item):
        if i < len(self.dist):
            self.dist[i] = item
        else:
            self.dist.append(item)

    def __len__(self):
        return len(self.dist)


class SphereTransform(Transform):
    """within=True: points outside the ball of `radius` are pulled onto its surface, points inside are
    left alone; within=False: everything is projected onto the sphere."""
    domain = constraints.real
    codomain = constraints.real
    bijective = False

    def __init__(self, centre=None, radius=1.0, within=True):
        super().__init__()
        self.centre, self.radius, self.within = centre, radius, within

    def _call(self, x):
        c = 0.0 if self.centre is None else self.centre.unsqueeze(-1)
        s = x - c
        mag = s.norm(dim=-1, keepdim=True)
        if self.within:
            mag = torch.clamp(mag, min=self.radius)
        return s / mag * self.radius + c

    def _inverse(self, y):
        return y

    def log_abs_det_jacobian(self, x, y):
        return torch.zeros(x.shape[:-1])
