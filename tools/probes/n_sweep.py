"""Diagnostic: step every ACTION_TYPE at awkward swarm sizes (workgroup tails, ring exchange odd/even, 1024-thread path)
and check against the CPU oracle for a few steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd, oracle
from util_scenarios import ActionStream, grid_spawn
worst = 0.0
for N in (2, 31, 63, 65, 66, 127, 128, 129, 200, 255, 257, 512, 1000, 1024):
    E = 5 if N < 300 else 2
    pos, eul = grid_spawn(E, N, seed=N)
    pos[..., 2] = 0.55 + 0.5 * (pos[..., 2] - 1.0)            # low: part of the swarm reaches the ground within the run
    z = np.zeros((E, N, 3), np.float32)
    for atype in ("set_target_vel", "set_speeds", "set_control"):
        sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
        sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
        sw = oracle.OracleSwarm(E, N)
        sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z.astype(np.float64), angvel=z.astype(np.float64))
        acts = ActionStream(atype, E, N, pos, seed=3, coherent=True)
        obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
        dense = torch.zeros(E, N, N, device="cuda:0")
        for t in range(25):
            a = acts(t)
            sh.step(torch.from_numpy(a).cuda(), atype, obs_out=obs, adj_out=adj, comm_range=2.0)
            sw.step(a, atype)
        torch.cuda.synchronize()
        err = float(np.abs(sh.view(sh.pos).cpu().numpy() - sw.pos).max())
        sh.adjacency_expand(adj, dense)
        A_or = sw.adjacency(2.0)
        bad = int((dense.cpu().numpy() != A_or).sum())
        worst = max(worst, err)
        print("N=%4d E=%d %-15s pos err %.2e  adjacency mismatches %d  min z %.3f" % (N, E, atype, err, bad, float(sh.pos[2].min())), flush=True)
print("worst position error %.2e" % worst)
