"""The PRODUCT's N > 1 path on the GPU: two ranks launched by torch.distributed.run, each a SwarmShard of its
env block, the joint observation all-gathered every step.  With two or more GPUs on the box the ranks sit on cuda:0 /
cuda:1 and the collective is RCCL; on a one-GPU box both ranks share cuda:0 and gloo carries the collective (RCCL cannot
put two ranks on one device) -- the sharding, the side-stream gather with its back-pressure and the unequal shard sizes
(7 envs over 2 ranks) are the same code either way.  The concatenated shard states and every gathered observation must
equal the single-process run BITWISE: envs are independent."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from util_scenarios import ActionStream, grid_spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E_TOTAL, N, STEPS, ATYPE = 7, 64, 40, "set_target_vel"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("ranks,mode", [(2, "collective"), (1, "collective"), (1, "direct"), (2, "direct")])
def test_product_ranks_equal_single_gpu_bitwise(tmp_path, ranks, mode):
    """ranks = 2: see the module text.  ranks = 1: the same worker as ONE rank, which always sits on RCCL (backend "nccl") --
    on the one-GPU boxes of this pool the only way the RCCL leg of ObsAllGather / gather_global_state (communicator set-up,
    all_gather_into_tensor on the side stream, the event hand-shake with the step stream) runs at all."""
    import mrsgym_amd
    if mode == "direct" and ranks > 1 and torch.cuda.device_count() < ranks:
        pytest.skip("the one-shot form (grouped point-to-point operations) needs RCCL, i.e. one device per rank; its logic runs on "
                    "the CPU in tests/test_dist_gloo.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MRS_DIST_ALLGATHER=mode)
    if ranks == 1:
        env["MRS_DIST_FORCE"] = "1"             # mrsgym_amd.dist: the collective path with one rank
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_product_worker.py"), str(tmp_path), str(E_TOTAL), str(N),
           str(STEPS), ATYPE]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = np.load(os.path.join(str(tmp_path), "dist_product.npz"))
    if torch.cuda.device_count() >= ranks:
        assert str(d["backend"]) == "nccl"      # RCCL carried the gather
    # the same swarm in one process on one GPU
    pos, eul = grid_spawn(E_TOTAL, N)
    acts = ActionStream(ATYPE, E_TOTAL, N, pos, seed=3)
    sh = mrsgym_amd.SwarmShard(E_TOTAL, N, "cuda:0")
    z = np.zeros((E_TOTAL, N, 3), np.float32)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    obs = torch.zeros(E_TOTAL, N, sh.D, device="cuda:0")
    adj = torch.zeros(E_TOTAL, N, sh.W, dtype=torch.int64, device="cuda:0")
    joint = []
    for t in range(STEPS):
        sh.step(torch.from_numpy(acts(t)).cuda(), ATYPE, obs_out=obs, adj_out=adj, comm_range=2.5)
        if t % 5 == 4 or t == STEPS - 1:
            joint.append(obs.cpu().numpy().copy())
    state = torch.cat([sh.view(sh.pos), sh.view(sh.quat), sh.view(sh.vel), sh.view(sh.angvel)], -1).cpu().numpy()
    assert np.array_equal(d["joint"], np.stack(joint))       # every gathered joint observation
    assert np.array_equal(d["state"], state)                 # concatenated shard states == the single-GPU state
    assert np.array_equal(d["adj"], adj.cpu().numpy())
