"""Diagnostic: steady-state duration of one mrs_step launch for ONE build of the library (MRS_HIP_LIB selects it).

    MRS_HIP_LIB=build/abl/libX.so python tools/steady_bench.py [tag]

Bench workload (N=64 x 4096 envs, set_target_vel, obs + packed adjacency), ROLLIN untimed steps so that ~1/4 of the
swarm is on the ground, then K timed steps between two HIP events on the launch stream (no host work in between but
the ctypes call).  Prints one line: tag, us per step, and a checksum of the final positions (same code path =>
same checksum; a variant that changes arithmetic shows up here)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("MRS_HIP_LIB", "default"))
E = int(os.environ.get("E", 4096)); N = int(os.environ.get("N", 64))
ROLLIN = int(os.environ.get("ROLLIN", 700)); K = int(os.environ.get("K", 500)); REPS = int(os.environ.get("REPS", 3))
ATYPE = os.environ.get("ATYPE", "set_target_vel")
pos, eul = grid_spawn(E, N)
z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
if os.environ.get("NOCONTACT"):     # what the step costs without the ground: no contact hand-off, nothing to solve
    prm = mrsgym_amd.default_params(); prm.enable_contact = 0; prm.ground_z = -1e9
    sh.set_params(prm)
if os.environ.get("SOLVER_ITERS"):  # what the step would cost if the contact sweeps stopped after this many (the run's physics changes)
    prm = mrsgym_amd.default_params(); prm.solver_iters = int(os.environ["SOLVER_ITERS"])
    sh.set_params(prm)
if os.environ.get("PAIR") == "0":    # without quad-quad contact (MrsParams.pair_contact)
    prm = mrsgym_amd.default_params(); prm.pair_contact = 0
    sh.set_params(prm)
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream(ATYPE, E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range((ROLLIN + K * REPS) // 50 + 2)]
obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
at = ACT[ATYPE]
if os.environ.get("NOADJ"):          # RETURN_A = False (BASELINE config 2): no adjacency rows
    adj = None
ADJ = adj.data_ptr() if adj is not None else 0
CR = 5.0 if adj is not None else float("nan")
t = 0
for _ in range(ROLLIN):
    sh.step_ptr(table[t // 50], at, obs.data_ptr(), ADJ, CR); t += 1
torch.cuda.synchronize()
if os.environ.get("KO"):            # a -DMRS_KO=-1 build: parts of the step switched off for the timed steps only (tools/abl_run.sh "name:KO=3")
    os.environ["MRS_KO"] = os.environ["KO"]
res = []
for r in range(REPS):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        sh.step_ptr(table[t // 50], at, obs.data_ptr(), ADJ, CR); t += 1
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / K * 1e3)
grounded = float((sh.pos[2] < 0.6).float().mean())
chk = float(sh.pos.double().abs().sum())
print("%-28s %s us/step  (min %.2f)  grounded %.3f  checksum %.10e" % (tag, " ".join("%.2f" % x for x in res), min(res), grounded, chk), flush=True)
