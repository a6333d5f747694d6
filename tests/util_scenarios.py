"""Synthetic swarms of SURVEY.md section 8d, shared by the parity tests and bench.py (numpy only)."""
import numpy as np

HOVER_RPM = 14475.809152959684
ADIM = {"set_speeds": 4, "set_control": 4, "set_target_accel": 3, "set_target_vel": 3, "set_target_pos": 3,
        "set_target_ori": 3}


def grid_spawn(E, N, seed=0, env_base=0, pitch=1.0, jitter=0.2):
    """sqrt(N) x sqrt(N) grid, pitch 1 m, xy jitter U[-.2,.2], z U[1,3]; yaw U[-pi/2,pi/2]; per-env seed
    0x5EED0000 + global env index so that shards reproduce the single-GPU swarm."""
    side = int(np.ceil(np.sqrt(N)))
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:N] - (side - 1) / 2
    pos = np.zeros((E, N, 3), np.float32)
    eul = np.zeros((E, N, 3), np.float32)
    for e in range(E):
        rng = np.random.default_rng(0x5EED0000 + seed * 1000003 + env_base + e)
        pos[e, :, :2] = g * pitch + rng.uniform(-jitter, jitter, (N, 2))
        pos[e, :, 2] = rng.uniform(1.0, 3.0, N)
        eul[e, :, 2] = rng.uniform(-np.pi / 2, np.pi / 2, N)
    return pos, eul


class ActionStream:
    """Deterministic per-step actions for each ACTION_TYPE."""

    def __init__(self, atype, E, N, start_pos, seed=1):
        self.atype, self.E, self.N = atype, E, N
        self.rng = np.random.default_rng(seed)
        self.start = start_pos
        self.cur = None

    def __call__(self, t):
        r, E, N = self.rng, self.E, self.N
        a = self.atype
        if a == "set_speeds":
            self.cur = HOVER_RPM * (1 + 0.05 * r.uniform(-1, 1, (E, N, 4)))
        elif a == "set_target_vel":
            if t % 50 == 0:
                self.cur = r.uniform(-1, 1, (E, N, 3))
        elif a == "set_control":
            if t % 10 == 0:
                self.cur = np.concatenate([9.81 + r.uniform(-1, 1, (E, N, 1)), r.uniform(-1, 1, (E, N, 3))], -1)
        elif a == "set_target_pos":
            if t % 100 == 0:
                self.cur = self.start + r.uniform(-1, 1, (E, N, 3))
        elif a == "set_target_accel":
            if t % 25 == 0:
                self.cur = r.uniform(-1, 1, (E, N, 3))
        elif a == "set_target_ori":
            if t % 40 == 0:
                self.cur = r.uniform(-.2, .2, (E, N, 3))
        return self.cur.astype(np.float32)
