"""Diagnostic: per-component deviation of one teacher-forced F6 step (tests/test_gpu_parity.py::test_reference_trajectories_F6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
name, T0 = sys.argv[1], int(sys.argv[2])
d = np.load(os.path.join(ROOT, "tests", "golden", "F6_step_%s.npz" % name))
N, atype = int(name.split("_")[0][1:]), name.split("_", 1)[1]
sh = mrsgym_amd.SwarmShard(1, N, "cuda:0")
np.set_printoptions(linewidth=200, precision=3)
for t in range(max(T0 - 3, 1), min(T0 + 4, d["actions"].shape[0])):
    s0 = d["state"][t - 1]
    sh.set_state_f64(pos=s0[None, :, 0:3], quat=s0[None, :, 3:7], vel=s0[None, :, 7:10], angvel=s0[None, :, 10:13])
    sh.step(torch.from_numpy(d["actions"][t][None]).cuda(), atype)
    st = np.concatenate([sh.view(sh.pos)[0].cpu().numpy(), sh.view(sh.quat)[0].cpu().numpy(), sh.view(sh.vel)[0].cpu().numpy(), sh.view(sh.angvel)[0].cpu().numpy()], 1)
    s = d["state"][t]
    err = np.abs(st - s)
    i = np.unravel_index(err.argmax(), err.shape)
    print("t=%d max err %.3e at agent %d comp %d; value %.4f; z=%.4f |w|=%.2f |v|=%.2f" % (t, err.max(), i[0], i[1], s[i], s[i[0], 2], np.linalg.norm(s[i[0], 10:13]), np.linalg.norm(s[i[0], 7:10])))
