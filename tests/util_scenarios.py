"""Synthetic swarms of SURVEY.md section 8d, shared by the parity tests and bench.py (numpy only)."""
import numpy as np

HOVER_RPM = 14475.809152959684
ADIM = {"set_speeds": 4, "set_control": 4, "set_target_accel": 3, "set_target_vel": 3, "set_target_pos": 3,
        "set_target_ori": 3}


def grid_spawn(E, N, seed=0, env_base=0, pitch=1.0, jitter=0.2, yaw_range=np.pi / 2):
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
        eul[e, :, 2] = rng.uniform(-yaw_range, yaw_range, N)
    return pos, eul


class ActionStream:
    """Deterministic per-step actions for each ACTION_TYPE.

    coherent=False is the throughput workload of SURVEY.md 8d (independent per-agent targets).  That
    workload is CHAOTIC by construction: quadcopters cross under each other and the reference's
    downwash term grows like 1/dz^2 with no quad-quad collision to stop dz -> 0, and drones reach the
    ground; two implementations that differ in the last bit of one float32 exp() then decorrelate.
    coherent=True is the tolerance-test variant: the whole env shares a target (plus a small
    per-agent offset), vertical targets alternate in sign / are floored, so the swarm stays a
    formation in free flight and errors stay at the float32-noise floor of the reference's design.
    Target steps are kept small (x0.4 m/s, +-0.3 m) and the tests spawn with |yaw| <= 0.8: the
    reference's cascade feeds the WORLD-frame angular velocity into the body-rate D term
    (Quadcopter.py:54), which destabilises the attitude loop for |yaw| >~ 1.1 rad (measured on the
    oracle) -- again chaos, not an implementation difference.
    """

    def __init__(self, atype, E, N, start_pos, seed=1, coherent=False):
        self.atype, self.E, self.N, self.coherent, self.k = atype, E, N, coherent, 0
        self.rng = np.random.default_rng(seed)
        self.start = start_pos
        self.cur = None

    def _u(self, lo, hi, d):
        r, E, N = self.rng, self.E, self.N
        if self.coherent:
            return r.uniform(lo, hi, (E, 1, d)) + 0.05 * r.uniform(lo, hi, (E, N, d))
        return r.uniform(lo, hi, (E, N, d))

    def __call__(self, t):
        r, E, N = self.rng, self.E, self.N
        a = self.atype
        if a == "set_speeds":
            # open loop: +-5 % makes every drone tumble and fall through its neighbours' downwash
            # cones within ~150 steps; the tolerance variant keeps the swarm near hover
            self.cur = HOVER_RPM * (1 + (0.002 if self.coherent else 0.05) * r.uniform(-1, 1, (E, N, 4)))
        elif a == "set_target_vel":
            if t % 50 == 0:
                self.cur = self._u(-1, 1, 3)
                if self.coherent:
                    self.k += 1
                    self.cur *= 0.4
                    self.cur[..., 2] = (-1) ** self.k * 0.3 * np.abs(self.cur[..., 2])
        elif a == "set_control":
            if t % 10 == 0:
                self.cur = np.concatenate([9.81 + r.uniform(-1, 1, (E, N, 1)), r.uniform(-1, 1, (E, N, 3))], -1)
        elif a == "set_target_pos":
            if t % 100 == 0:
                self.cur = self.start + self._u(-1, 1, 3) * (0.3 if self.coherent else 1.0)
                if self.coherent:
                    self.cur[..., 2] = np.maximum(self.cur[..., 2], 1.2)
        elif a == "set_target_accel":
            if t % 25 == 0:
                self.cur = self._u(-1, 1, 3)
                if self.coherent:
                    self.k += 1
                    self.cur = (-1) ** self.k * np.abs(self.cur) * 0.3
        elif a == "set_target_ori":
            if t % 40 == 0:
                self.cur = self._u(-.2, .2, 3)
        return self.cur.astype(np.float32)
