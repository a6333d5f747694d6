import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E, N = 6, 64
pos, eul = grid_spawn(E, N, seed=4)
z = np.zeros((E, N, 3), np.float32)
shards = []
for split in ("0", "1"):
    os.environ["MRS_STEP_SPLIT"] = split
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    os.environ.pop("MRS_STEP_SPLIT", None)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    shards.append((sh, torch.zeros(E, N, sh.D, device="cuda:0"), torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")))
atype = "set_target_vel"
acts = ActionStream(atype, E, N, pos, seed=9)
for t in range(300):
    a = torch.from_numpy(acts(t)).cuda()
    for sh, obs, adj in shards:
        sh.step(a, atype, obs_out=obs, adj_out=adj, comm_range=2.5)
    s0, s1 = shards[0][0], shards[1][0]
    bad = False
    for name in ("pos", "quat", "vel", "angvel", "pid"):
        d = (getattr(s0, name) - getattr(s1, name)).abs()
        d = torch.nan_to_num(d, nan=0.0)
        if float(d.max()) > 0:
            idx = int(d.max(dim=0)[0].argmax())
            print("t=%d %s maxdiff %.3e at agent %d (z=%.4f) count %d" % (t, name, float(d.max()), idx, float(s0.pos[2, idx]), int((d > 0).sum())))
            bad = True
    if bad: break
