"""Diagnostic: which buffer's PLACEMENT makes one swarm of a process 0.45 us per step slower than another of the same build?
(tools/ab_inproc.py: eight swarms of one build fall into two groups, 22.45 and 22.95 us.)  K swarms with bitwise identical
states are stepped in interleaved blocks; then the fastest and the slowest exchange one group of buffers at a time (the
tensors only -- the values are identical) and both are measured again: the group that carries the slowness with it is the one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn

E, N, K = 4096, 64, int(os.environ.get("K", 8))
ROLLIN, BLOCK, ROUNDS = 700, 100, int(os.environ.get("ROUNDS", 10))
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(400)]
AT = ACT["set_target_vel"]


class Run:
    def __init__(self):
        self.sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
        self.sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
        self.obs = torch.zeros(E, N, self.sh.D, device="cuda:0")
        self.adj = torch.zeros(E, N, self.sh.W, dtype=torch.int64, device="cuda:0")
        self.t = 0

    def advance(self, n):
        sh, o, a = self.sh, self.obs.data_ptr(), self.adj.data_ptr()
        for _ in range(n):
            sh.step_ptr(table[(self.t // 50) % len(table)], AT, o, a, 5.0); self.t += 1


def measure(runs, rounds=ROUNDS):
    us = [[] for _ in runs]
    for rnd in range(rounds):
        for j in range(len(runs)):
            i = (j + rnd) % len(runs)
            r = runs[i]
            r.advance(3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.advance(BLOCK - 3); e1.record(); torch.cuda.synchronize()
            us[i].append(e0.elapsed_time(e1) / (BLOCK - 3) * 1e3)
    return np.array([np.mean(u) for u in us]), np.array([np.std(u) / np.sqrt(len(u)) for u in us])


GROUPS = {"obs": ["obs"], "adj": ["adj"], "state planes": ["pos", "quat", "vel", "angvel"], "controller memory": ["pid"], "status": ["status"]}


def swap(a, b, names):
    for n in names:
        if n in ("obs", "adj"):
            x, y = getattr(a, n), getattr(b, n); setattr(a, n, y); setattr(b, n, x)
        else:
            x, y = getattr(a.sh, n), getattr(b.sh, n); setattr(a.sh, n, y); setattr(b.sh, n, x)
    for r in (a, b):
        r.sh._pb = r.sh._buffers(); r.sh._pb_ref = __import__("ctypes").byref(r.sh._pb)


runs = [Run() for _ in range(K)]
for r in runs:
    r.advance(ROLLIN)
torch.cuda.synchronize()
m, s = measure(runs)
print("the %d swarms: %s us per step (+- %.2f)" % (K, " ".join("%.2f" % x for x in m), s.mean()))
f, sl = int(np.argmin(m)), int(np.argmax(m))
F, S = runs[f], runs[sl]
print("fastest #%d %.2f, slowest #%d %.2f" % (f, m[f], sl, m[sl]))
m2, _ = measure([F, S])
print("the two alone, again:                       fast %.2f  slow %.2f" % tuple(m2))
for g, names in GROUPS.items():
    swap(F, S, names)
    m3, _ = measure([F, S])
    print("after exchanging %-18s        fast-handle %.2f  slow-handle %.2f   %s" % (g + ":", m3[0], m3[1], "<- the slowness moved" if m3[0] - m3[1] > 0.5 * (m2[1] - m2[0]) else ""))
    swap(F, S, names)
allnames = sum(GROUPS.values(), [])
swap(F, S, allnames)
m4, _ = measure([F, S])
print("after exchanging every caller-owned buffer:   fast-handle %.2f  slow-handle %.2f   (what stays is the handle's own workspace)" % tuple(m4))
