"""Diagnostic: long rollouts on the GPU box -- (1) the bench workload twice for STEPS steps: no NaN, no error status, final
state bitwise equal between the two runs; (2) a vectorised data-generation loop with AUTO_RESET (masked device spawn every few
steps, which marks only those envs' contact flags unknown) for STEPS steps: no exception, spawn separation kept, finite state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
STEPS = int(os.environ.get("STEPS", 5000))

def run(E, N):
    pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
    table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(40)]
    obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda")
    for t in range(STEPS):
        sh.step_ptr(table[(t // 50) % 40], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
    torch.cuda.synchronize()
    st = {k: sh.view(getattr(sh, k)).clone() for k in ("pos", "quat", "vel", "angvel")}
    return st, int(sh.status.max()), obs.clone(), adj.clone()

for E, N in ((4096, 64), (1024, 12), (256, 130), (1024, 256), (512, 192), (1024, 128)):   # the last three: envs of whole waves (block scheme)
    t0 = time.time()
    a, sa, oa, aa = run(E, N)
    b, sb, ob, ab = run(E, N)
    fin = all(bool(torch.isfinite(v).all()) for v in a.values())
    same = all(torch.equal(a[k], b[k]) for k in a) and torch.equal(oa, ob) and torch.equal(aa, ab)
    print("E=%d N=%d: %d steps x2 in %.1f s; finite %s, status %d/%d, runs bitwise equal %s, grounded %.2f, max |v| %.1f" % (
        E, N, STEPS, time.time() - t0, fin, sa, sb, same, float((a["pos"][..., 2] < 0.6).float().mean()), float(a["vel"].abs().max())), flush=True)
    assert fin and same and sa == 0 and sb == 0

def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
from mrsgym_amd.util import CombinedDistribution
from torch.distributions import Normal, Uniform
E, N = 512, 12
dist = CombinedDistribution([Normal(torch.zeros(N, 2), 1.25), Uniform(2.0 * torch.ones(N, 1), 5.0 * torch.ones(N, 1))], mixer='cat', dim=1)
mrs = None
def done_fn(A=None, **kw):
    return (A[:, 0].sum(dim=-1) == 0).any(dim=-1) | (mrs.env_steps() + 1 >= 40)
mrs = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1, COMM_RANGE=2.5, START_POS=dist,
                      ACTION_TYPE='set_target_vel', done_fn=done_fn, AUTO_RESET=True, SEED=3)
model = mrsgym_amd.Reynolds(N=N, D=6, K=1, OUT_DIM=3)
resets = 0; t0 = time.time()
for t in range(min(STEPS, 2000)):
    action = model.from_env(mrs)
    action[:, 0, :] = torch.tensor([0.3, 0.0, 0.0], device=action.device)
    X, r, done, info = mrs.step(action)
    resets += int(done.sum())
mrs.check_errors()
print("AUTO_RESET loop: %d steps, %d env resets in %.1f s; finite %s" % (min(STEPS, 2000), resets, time.time() - t0, bool(torch.isfinite(X).all())), flush=True)
assert bool(torch.isfinite(X).all()) and resets > 0
print("soak ok")
