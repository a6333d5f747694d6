"""info["A"] of A_FORMAT = "dense": the reference's float32 0/1 stack (MRS.py:102-124), materialised on first use.

The step kernel always writes the bit-packed rows (8 bytes per agent-step at N = 64).  The float32 (E, K+1, N, N) tensor the
reference returns is 256 bytes per agent-step: a loop that never looks at A (most training loops read X) should not pay for it,
one that does must get the identical tensor.  `LazyDenseA` IS a torch.Tensor (isinstance, .shape, .dtype, .device answer without
any work); the first operation on it -- indexing, arithmetic, .cpu(), torch.equal, anything that reaches the dispatcher -- asks
the environment for the real tensor (MRS._materialize_A), which from then on keeps the dense ring up to date inside the step
kernel for as long as A keeps being read.
"""
import torch
from torch.utils._pytree import tree_map


class LazyDenseA(torch.Tensor):
    @staticmethod
    def __new__(cls, make, shape, device):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=device, requires_grad=False)
        r._make, r._real = make, None
        return r

    def materialize(self):
        """The real float32 tensor (idempotent)."""
        if self._real is None:
            self._real = self._make()
            self._make = None
        return self._real

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        un = lambda x: x.materialize() if isinstance(x, LazyDenseA) else x
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    def __repr__(self):
        return "LazyDenseA(%s, %s)" % (tuple(self.shape), "materialised" if self._real is not None else "pending")

    def numpy(self, *a, **k):
        return self.materialize().numpy(*a, **k)

    def __array__(self, dtype=None):
        a = self.materialize().detach().cpu().numpy()
        return a if dtype is None else a.astype(dtype)

    def tolist(self):
        return self.materialize().tolist()

    def data_ptr(self):
        return self.materialize().data_ptr()
