"""Diagnostic: per-wave phase timestamps of the fused step kernel (profiles/r01_timeline.txt).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=14 -DMRS_TIMELINE \
          mrs-gym_amd/csrc/mrs_kernels.hip -o build/abl/libmrs_tl.so
    MRS_HIP_LIB=build/abl/libmrs_tl.so python tools/probes/timeline_probe.py        (on the GPU box)

In that build lane 0 of every wave writes clock64() deltas at the phase boundaries into the rpm buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = int(os.environ.get("E", 4096)), int(os.environ.get("N", 64))
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0", want_rpm=True)
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(20)]
obs = torch.zeros(E, N, sh.D, device="cuda"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda")   # W words per row: N > 64 has several
names = ["loads+tile+vote", "downwash", "controller", "forces+intvel", "barrier1", "own solve", "barrier2", "pose+store", "obs+adj(end)"]
for T0 in (200, 800):
    for t in range(T0 if T0 == 200 else 600):
        sh.step_ptr(table[(t // 50) % 20], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
    torch.cuda.synchronize()
    W = E if N == 64 else -(-E // (256 // N)) * 4          # waves of the launch (N = 64: one per env; else 256-thread workgroups)
    full = sh.rpm.flatten()[:W * 16].view(W, 16).cpu().numpy()
    tl = full[:, :9]      # one wave per env at N=64
    if full[:, 9].max() > 0:
        print("  head of the kernel: arguments arrived + loads issued at %.0f [%.0f .. %.0f], positions staged at %.0f [%.0f .. %.0f]" % (
            full[:, 9].mean(), full[:, 9].min(), full[:, 9].max(), full[:, 10].mean(), full[:, 10].min(), full[:, 10].max()))
        t0, t1 = full[:, 11], full[:, 12]          # 10 ns ticks, 20 bits: one launch fits without a wrap almost always
        if t1.max() - t0.min() < 1e4:
            first = t0.min()
            print("  wave starts after the first wave's: median %.2f us, 90%% %.2f us, last %.2f us; wave ends: first %.2f us, median %.2f us, last %.2f us" % (
                np.median(t0 - first) / 100, np.percentile(t0 - first, 90) / 100, (t0.max() - first) / 100,
                (t1.min() - first) / 100, np.median(t1 - first) / 100, (t1.max() - first) / 100))
    print("after %d steps (clock64 ticks; mean over %d waves, [min..max] of the cumulative stamp)" % (T0, len(tl)))
    prev = np.zeros(len(tl))
    for k, nm in enumerate(names):
        col = tl[:, k]
        valid = col > 0
        d = (col - prev)[valid]
        print("  %-18s +%8.0f   cumulative %8.0f [%8.0f .. %8.0f]" % (nm, d.mean() if d.size else 0, col[valid].mean() if valid.any() else 0, col[valid].min() if valid.any() else 0, col[valid].max() if valid.any() else 0))
        prev = np.where(valid, col, prev)
