"""Diagnostic: the bench workload's step time over ten seconds of continuous stepping (blocks of 2000 launches), with the shader
clock rocm-smi reports between blocks: is a slow box slow from the start, does it ramp, does it throttle?"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(40)]
obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
AT = ACT["set_target_vel"]


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Average Graphics Package Power", "Current Socket Graphics Package Power", "Temperature (Sensor junction)", "mclk"))]
        return " | ".join(x.split(":", 1)[-1].strip() if ":" in x else x for x in keep[:5])
    except Exception as e:
        return "rocm-smi: %s" % e


t = 0
print("idle:", smi(), flush=True)
T0 = time.time()
for blk in range(int(os.environ.get("BLOCKS", 24))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000):
        sh.step_ptr(table[(t // 50) % 40], AT, obs.data_ptr(), adj.data_ptr(), 5.0); t += 1
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 2000 * 1e3
    # re-spawn every 4 blocks so that the workload stays the bench's (a quarter on the ground), not an all-crashed swarm
    if blk % 4 == 3:
        sh.set_state(pos=pos, ori=eul, vel=z, angvel=z); sh.pid_reset(); t = 0
    print("t=%5.1f s  block %2d  %.2f us/step  grounded %.2f  %s" % (time.time() - T0, blk, us, float((sh.pos[2] < 0.6).float().mean()), smi() if blk % 3 == 0 else ""), flush=True)
