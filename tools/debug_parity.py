"""GPU-vs-oracle error growth per ACTION_TYPE (diagnostic, run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
import oracle
import mrsgym_amd
from util_scenarios import ActionStream, grid_spawn

E, N = 4, 64
coherent = os.environ.get("COHERENT", "1") == "1"
steps = int(os.environ.get("STEPS", "1000"))
OPEN = ("set_speeds", "set_control", "set_target_accel", "set_target_ori")
for atype in sys.argv[1:] or ["set_target_vel", "set_target_pos", "set_speeds", "set_control", "set_target_ori", "set_target_accel"]:
    pos, eul = grid_spawn(E, N, yaw_range=0.8 if coherent else np.pi / 2)
    z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0", want_rpm=True)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    sw = oracle.OracleSwarm(E, N, nthreads=8)
    sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z.astype(np.float64), angvel=z.astype(np.float64))
    if atype in OPEN:
        p = mrsgym_amd.default_params(); p.enable_contact = 0; p.ground_z = -1e9
        sh.set_params(p); sw.p.enable_contact = 0; sw.p.ground_z = -1e9
    acts = ActionStream(atype, E, N, pos, seed=11, coherent=coherent)
    for t in range(steps):
        a = acts(t)
        sh.step(torch.from_numpy(a).cuda(), atype)
        sw.step(a, atype)
        if t in (0, 9, 49, 99, 199, 299, 499, 699, 999):
            g = {k: sh.view(getattr(sh, k)).cpu().numpy() for k in ("pos", "quat", "vel", "angvel")}
            d = np.linalg.norm(sw.pos[:, :, None] - sw.pos[:, None], axis=-1) + np.eye(N) * 1e9
            print(atype, t, "pos %.1e vel %.1e ang %.1e quat %.1e" % (
                np.abs(g["pos"] - sw.pos).max(), np.abs(g["vel"] - sw.vel).max(), np.abs(g["angvel"] - sw.angvel).max(),
                np.abs(g["quat"] - sw.quat).max()), "zmin %.2f mindist %.2f |v|max %.1f" % (
                sw.pos[..., 2].min(), d.min(), np.abs(sw.vel).max()), flush=True)
