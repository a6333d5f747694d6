"""Diagnostic: host-side profile (cProfile) of a vectorised AUTO_RESET loop with a user START_POS distribution."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import torch, mrsgym_amd
from mrsgym_amd.util import CombinedDistribution
from torch.distributions import Normal, Uniform
E, N = int(os.environ.get("E", 512)), 12
def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
dist = CombinedDistribution([Normal(torch.zeros(N, 2), 1.25), Uniform(2.0 * torch.ones(N, 1), 5.0 * torch.ones(N, 1))], mixer='cat', dim=1)
mrs = None
def done_fn(A=None, **kw):
    return (A[:, 0].sum(dim=-1) == 0).any(dim=-1) | (mrs.env_steps() + 1 >= 40)
mrs = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1, COMM_RANGE=2.5, START_POS=dist,
                      ACTION_TYPE='set_target_vel', done_fn=done_fn, AUTO_RESET=True, SEED=3)
model = mrsgym_amd.Reynolds(N=N, D=6, K=1, OUT_DIM=3)
def loop(n):
    r = 0
    for t in range(n):
        action = model.from_env(mrs)
        action[:, 0, :] = torch.tensor([0.3, 0.0, 0.0], device=action.device)
        X, rew, done, info = mrs.step(action)
        r += int(done.sum())
    return r
loop(100)
torch.cuda.synchronize(); t0 = time.time(); r = loop(300); torch.cuda.synchronize()
print("%.2f ms per step, %d resets" % ((time.time() - t0) / 300 * 1e3, r))
pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
