"""Diagnostic: does replaying the step's three launches from a hipGraph shorten the gaps between them?

Captures 2 consecutive mrs_step calls (the contact counters alternate by step parity, so an even number of
steps makes a replayable unit) into a torch CUDAGraph on a side stream and compares stream time per step
with plain launches.  Workload = bench.py's (C3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn

E = int(os.environ.get("E", 4096)); N = int(os.environ.get("N", 64)); K = int(os.environ.get("K", 1000))
UNIT = int(os.environ.get("UNIT", 2))
pos, eul = grid_spawn(E, N)
z = np.zeros((E, N, 3), np.float32)
acts = ActionStream("set_target_vel", E, N, pos, seed=11)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(K // 50 + 1)]
at = ACT["set_target_vel"]


def fresh():
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
    return sh, obs, adj


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    fn()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6


# plain launches
sh, obs, adj = fresh()
a_static = table[0].clone()
for t in range(100): sh.step_ptr(a_static, at, obs.data_ptr(), adj.data_ptr(), 5.0)
def plain():
    for t in range(K):
        if t % 50 == 0: a_static.copy_(table[t // 50])
        sh.step_ptr(a_static, at, obs.data_ptr(), adj.data_ptr(), 5.0)
ev, wall = timed(plain, K)
ref_pos = sh.pos.clone()
print("plain launches : %.1f us/step events, %.1f us/step wall" % (ev, wall), flush=True)

# graph replay
sh, obs, adj = fresh()
a_static = table[0].clone()
for t in range(100): sh.step_ptr(a_static, at, obs.data_ptr(), adj.data_ptr(), 5.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    # the captured unit advances the simulation by UNIT steps; state is rewound after capture
    snap = {k: v.clone() for k, v in sh.state_dict().items()}
    with torch.cuda.graph(g, stream=s):
        for _ in range(UNIT): sh.step_ptr(a_static, at, obs.data_ptr(), adj.data_ptr(), 5.0)
    sh.load_state_dict(snap)
torch.cuda.synchronize()
def replay():
    for t in range(0, K, UNIT):
        if t % 50 == 0: a_static.copy_(table[t // 50])
        g.replay()
ev, wall = timed(replay, K)
print("graph replay   : %.1f us/step events, %.1f us/step wall (unit = %d steps)" % (ev, wall, UNIT), flush=True)
print("state identical to plain run:", bool(torch.equal(ref_pos, sh.pos)))
