import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/mrs-gym_amd', '/root/repo/tests']
import numpy as np, torch, mrsgym_amd, oracle
from util_scenarios import ActionStream, grid_spawn
def run(scale, yaw, pitch, zabs, steps=1000, E=4, N=64):
    pos, eul = grid_spawn(E, N, seed=0, yaw_range=yaw, pitch=pitch)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0"); z = np.zeros((E, N, 3), np.float32)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    sw = oracle.OracleSwarm(E, N, nthreads=8)
    sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z.astype(np.float64), angvel=z.astype(np.float64))
    acts = ActionStream("set_target_vel", E, N, pos, seed=11, coherent=False)
    worst = 0; a = None
    for t in range(steps):
        a0 = acts(t)
        a = a0 * scale
        if zabs: a[..., 2] = np.where((t // 50) % 2 == 0, 1, -1) * np.abs(a[..., 2])
        sh.step(torch.from_numpy(a.astype(np.float32)).cuda(), "set_target_vel"); sw.step(a.astype(np.float32), "set_target_vel")
        if t % 100 == 99:
            e = max(np.abs(sh.view(getattr(sh, k)).cpu().numpy() - getattr(sw, k)).max() for k in ("pos", "quat", "vel"))
            if e > 1e-4 and worst <= 1e-4: print("   first above 1e-4 at step", t, "(%.1e)" % e)
            worst = max(worst, e)
    print("scale %.2f yaw %.2f pitch %.1f zabs %d: worst |gpu-oracle| pos/quat/vel %.2e, angvel %.2e, min z %.2f" % (scale, yaw, pitch, zabs, worst, np.abs(sh.view(sh.angvel).cpu().numpy() - sw.angvel).max(), sw.pos[..., 2].min()), flush=True)
for args in ((0.4, 0.8, 1.0, 1), (0.4, 0.8, 2.0, 1), (1.0, 0.8, 2.0, 1), (0.4, 1.0, 2.0, 1), (0.4, 1.2, 3.0, 1), (0.2, 1.2, 2.0, 1), (0.4, 0.8, 2.0, 0)):
    run(*args)
