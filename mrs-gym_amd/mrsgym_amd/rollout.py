"""Rollout log in the on-disk format of the reference's data generator.

The reference logs one sample per step through `Trainer.set_state(A, X, done, expert, context)`
(examples/simulating_data/helper/Trainer.py:89-108, called from DataGenerator.py:36) and saves
`torch.save({"history": {"A", "X", "done", "expert", "context": deques}, "iter", "sample_weights",
"sample_idxs"})` (Trainer.py:43-61); `read_data.py` / `analytics.py` load that file back.  With E envs stepping
at once the log is kept on the device -- one preallocated (T, E, ...) buffer per field, appended by plain device
copies, nothing read back per step -- and `save_trainer()` writes the same dict: env after env, each env's
samples in time order with `done` set on its last sample of every episode, A as the dense float32 (N,N) of
MRS.calc_A, tensors on the CPU.
"""
from collections import deque

import torch


class RolloutLog:
    def __init__(self, n_envs, n_agents, state_dim, capacity, out_dim=3, device="cuda"):
        self.E, self.N, self.D, self.T = int(n_envs), int(n_agents), int(state_dim), int(capacity)
        self.device = torch.device(device)
        self.W = (self.N + 63) // 64
        z = dict(device=self.device)
        self.X = torch.zeros(self.T, self.E, self.N, self.D, dtype=torch.float32, **z)
        self.A = torch.zeros(self.T, self.E, self.N, self.W, dtype=torch.int64, **z)      # bit-packed rows
        self.expert = torch.zeros(self.T, self.E, self.N, out_dim, dtype=torch.float32, **z)
        self.done = torch.zeros(self.T, self.E, dtype=torch.bool, **z)
        self.t = 0

    def __len__(self):
        return self.t

    def set_state(self, A, X, done=False, expert=None):
        """One sample per env (Trainer.set_state): X (E,N,D) newest observation, A (E,N,W) packed or (E,N,N) dense
        newest adjacency, done bool or (E,) tensor, expert (E,N,out_dim) or None."""
        if self.t >= self.T:
            raise IndexError("RolloutLog is full (%d steps)" % self.T)
        t = self.t
        self.X[t].copy_(X.reshape(self.E, self.N, self.D))
        A = A.reshape(self.E, self.N, -1)
        if A.dtype == torch.int64 and A.shape[-1] == self.W:
            self.A[t].copy_(A)
        else:       # dense 0/1 rows -> packed words
            bits = (A != 0).to(torch.int64)
            pad = self.W * 64 - self.N
            if pad:
                bits = torch.nn.functional.pad(bits, (0, pad))
            shifts = torch.arange(64, device=bits.device, dtype=torch.int64)
            self.A[t].copy_((bits.reshape(self.E, self.N, self.W, 64) << shifts).sum(-1))
        if isinstance(done, torch.Tensor):
            self.done[t].copy_(done.reshape(-1).expand(self.E) if done.numel() == 1 else done.reshape(self.E))
        else:
            self.done[t].fill_(bool(done))
        if expert is not None:
            self.expert[t].copy_(expert.reshape(self.E, self.N, -1))
        self.t = t + 1

    def dense_A(self, t0=0, t1=None):
        """(t1-t0, E, N, N) float32 0/1 from the packed rows."""
        t1 = self.t if t1 is None else t1
        shifts = torch.arange(64, device=self.device, dtype=torch.int64)
        bits = (self.A[t0:t1].unsqueeze(-1) >> shifts) & 1                # (T,E,N,W,64)
        return bits.reshape(t1 - t0, self.E, self.N, self.W * 64)[..., :self.N].to(torch.float32)

    def trainer_dict(self):
        """The dict Trainer.save_trainer writes (Trainer.py:43-61)."""
        T = self.t
        X = self.X[:T].permute(1, 0, 2, 3).cpu()              # (E,T,N,D): env after env
        A = self.dense_A(0, T).permute(1, 0, 2, 3).cpu()
        ex = self.expert[:T].permute(1, 0, 2, 3).cpu()
        done = self.done[:T].t().cpu().clone()
        if T:
            done[:, -1] = True                                 # Trainer.py:45-46: the last sample closes its episode
        hist = {"A": deque(), "X": deque(), "done": deque(), "expert": deque(), "context": deque()}
        for e in range(self.E):
            for t in range(T):
                hist["A"].append(A[e, t].clone()); hist["X"].append(X[e, t].clone())
                hist["done"].append(bool(done[e, t])); hist["expert"].append(ex[e, t].clone())
                hist["context"].append({})
        return {"history": hist, "iter": 0, "sample_weights": torch.tensor([]), "sample_idxs": []}

    def get_episodes(self):
        """Trainer.get_episodes (Trainer.py:186-207): every episode of every env as one row, padded with NaN to the longest:
        {"X": (episodes, max_len, N, D), "A": (episodes, max_len, N, N), "expert": (episodes, max_len, N, out)} on the device.
        An episode ends with the sample whose `done` is set; the last logged sample closes its env's open episode."""
        T = self.t
        done = self.done[:T].t().clone()                         # (E,T)
        if T:
            done[:, -1] = True
        dn = done.cpu()
        spans = []
        for e in range(self.E):
            start = 0
            for t in range(T):
                if bool(dn[e, t]):
                    spans.append((e, start, t + 1))
                    start = t + 1
        L = max((b - a for _, a, b in spans), default=0)
        nan = float("nan")
        X = torch.full((len(spans), L, self.N, self.D), nan, device=self.device)
        A = torch.full((len(spans), L, self.N, self.N), nan, device=self.device)
        ex = torch.full((len(spans), L, self.N, self.expert.shape[-1]), nan, device=self.device)
        dense = self.dense_A(0, T) if T else None
        for k, (e, a, b) in enumerate(spans):
            X[k, :b - a] = self.X[a:b, e]
            A[k, :b - a] = dense[a:b, e]
            ex[k, :b - a] = self.expert[a:b, e]
        return {"X": X, "A": A, "expert": ex}

    def save_trainer(self, path):
        data = self.trainer_dict()
        with open(path, "wb") as fp:
            torch.save(data, fp)
        return data
