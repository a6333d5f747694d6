"""Diagnostic: A/B several builds of the library IN ONE PROCESS, interleaved in short blocks, so that box-to-box and
minute-to-minute drift (about +-0.5 us on this pool: profiles/README.md) cancels out of the comparison.

    python tools/ab_inproc.py name1 name2[:SOLVER_ITERS=6] ...      (build/abl/lib<name>.so, tools/abl_build.sh)

Every build gets its own SwarmShard with the same spawn and the same action table (bench workload unless E / N / ATYPE /
NOADJ say otherwise), so all of them walk through the same states; after ROLLIN untimed steps the builds take turns,
BLOCK steps each between two HIP events, ROUNDS times, the order rotating by one every round.  Prints per build the mean
step time, the mean PAIRED difference to the first build and its standard error, and a checksum of the final positions."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from util_scenarios import ActionStream, grid_spawn

E = int(os.environ.get("E", 4096)); N = int(os.environ.get("N", 64))
ROLLIN = int(os.environ.get("ROLLIN", 700)); BLOCK = int(os.environ.get("BLOCK", 100)); ROUNDS = int(os.environ.get("ROUNDS", 12))
ATYPE = os.environ.get("ATYPE", "set_target_vel")
NATIVE = os.path.join(ROOT, "mrs-gym_amd", "mrsgym_amd", "native.py")


def load(spec, k):
    name = spec.split(":")[0]
    os.environ["MRS_HIP_LIB"] = os.path.join(ROOT, "build", "abl", "lib%s.so" % name)
    sp = importlib.util.spec_from_file_location("native_ab_%d" % k, NATIVE)      # a private copy of the binding per library
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    sh = m.SwarmShard(E, N, "cuda:0")
    prm = m.default_params()
    for kv in spec.split(":")[1:]:
        key, val = kv.split("=")
        setattr(prm, key.lower(), int(val))
    sh.set_params(prm)
    return m, sh


pos, eul = grid_spawn(E, N)
z = np.zeros((E, N, 3), np.float32)
acts = ActionStream(ATYPE, E, N, pos, seed=1000)
total = ROLLIN + BLOCK * ROUNDS
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(total // 50 + 2)]
specs = sys.argv[1:]
runs = []
for k, spec in enumerate(specs):
    m, sh = load(spec, k)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    obs = torch.zeros(E, N, sh.D, device="cuda:0")
    adj = None if os.environ.get("NOADJ") else torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
    runs.append(dict(spec=spec, sh=sh, at=m.ACT[ATYPE], obs=obs, adj=adj, ADJ=adj.data_ptr() if adj is not None else 0,
                     CR=5.0 if adj is not None else float("nan"), t=0, us=[]))


def advance(r, n):
    sh, at, o, a, cr = r["sh"], r["at"], r["obs"].data_ptr(), r["ADJ"], r["CR"]
    t = r["t"]
    for _ in range(n):
        sh.step_ptr(table[t // 50], at, o, a, cr); t += 1
    r["t"] = t


for r in runs:
    advance(r, ROLLIN)
torch.cuda.synchronize()
for rnd in range(ROUNDS):
    for j in range(len(runs)):
        r = runs[(j + rnd) % len(runs)]
        advance(r, 3)                                         # the first launches after another build's block pull the state back into the cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); advance(r, BLOCK - 3); e1.record(); torch.cuda.synchronize()
        r["us"].append(e0.elapsed_time(e1) / (BLOCK - 3) * 1e3)
base = np.array(runs[0]["us"])
for r in runs:
    u = np.array(r["us"]); d = u - base
    chk = float(r["sh"].pos.double().abs().sum())
    print("%-28s %6.2f us/step  (min %.2f)   vs %s: %+.3f +- %.3f   checksum %.10e" % (
        r["spec"], u.mean(), u.min(), runs[0]["spec"], d.mean(), d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else 0.0, chk), flush=True)
    if os.environ.get("ADDR"):      # where the buffers of this slot live (the same build differs by +-0.4 us from slot to slot)
        sh = r["sh"]
        print("    " + " ".join("%s=%x" % (k, t.data_ptr()) for k, t in (("pos", sh.pos), ("quat", sh.quat), ("vel", sh.vel), ("angvel", sh.angvel),
                                                                      ("pid", sh.pid), ("obs", r["obs"]), ("adj", r["adj"])) if t is not None))
