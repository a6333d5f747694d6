"""The N>1 path on CPU: world_size-2 gloo.  Env instances shard by contiguous global-index blocks, no
data-path collective inside step(), one all-gather of the newest observation slice.
The per-rank swarm here is the CPU oracle (the product kernels need a GPU); what is under test is the
product's sharding / gather logic in mrsgym_amd.dist, and that a sharded run equals the single-process
run BITWISE (envs are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util_scenarios import ActionStream, grid_spawn

E_TOTAL, N, STEPS, ATYPE = 7, 12, 25, "set_target_vel"   # 7 envs over 2 ranks: shards of 4 and 3 (unequal on purpose)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_block(lo, hi, all_actions):
    import oracle
    E = hi - lo
    pos, eul = grid_spawn(E, N, env_base=lo)
    sw = oracle.OracleSwarm(E, N)
    z = np.zeros((E, N, 3))
    sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z, angvel=z)
    obs = []
    for t in range(STEPS):
        sw.step(all_actions[t][lo:hi], ATYPE)
        o = sw.observe()
        obs.append(np.concatenate([o["pos"], o["vel"]], -1))
    return sw, obs


def _actions():
    pos, _ = grid_spawn(E_TOTAL, N)
    acts = ActionStream(ATYPE, E_TOTAL, N, pos, seed=3)
    return [acts(t) for t in range(STEPS)]


def _worker(rank, world, port, out_dir, mode="collective", every=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mrsgym_amd import dist as mdist
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = mdist.shard_range(E_TOTAL, rank, world)
    all_actions = _actions()
    sw, obs = _run_block(lo, hi, all_actions)
    gather = mdist.ObsAllGather(hi - lo, N, 6, "cpu", mode=mode, every=every)
    joint = []
    for t in range(STEPS):
        out = gather.gather(torch.from_numpy(obs[t]).contiguous())
        gather.wait()
        if every == 1 or t % every == every - 1:
            joint.append(out.clone().numpy())
        elif t < every - 1:
            assert out is None                                        # nothing has been exchanged yet
        else:
            assert np.array_equal(out.numpy(), joint[-1])             # between exchanges: the last joint tensor
    state = mdist.gather_global_state(torch.from_numpy(np.concatenate([sw.pos, sw.quat, sw.vel, sw.angvel], -1)))
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist.npz"), joint=np.stack(joint), state=state.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from mrsgym_amd.dist import shard_range
    for E in (1, 7, 8, 4096, 32768):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(E, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == E
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("mode,every", [("collective", 1), ("direct", 1), ("direct", 4), ("collective", 5)])
def test_two_rank_run_equals_single_process_bitwise(tmp_path, mode, every):
    """mode: the all-gather collective, or the one-shot form (every rank sends its slice to every peer, grouped point-to-point
    operations: SURVEY.md section 5); every = k: only every k-th step exchanges the joint observation."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode, every), nprocs=2, join=True)
    d = np.load(os.path.join(str(tmp_path), "dist.npz"))
    sw, obs = _run_block(0, E_TOTAL, _actions())
    assert np.array_equal(d["joint"], np.stack(obs)[every - 1::every])    # gathered joint observation, every k-th step
    want = np.concatenate([sw.pos, sw.quat, sw.vel, sw.angvel], -1)
    assert np.array_equal(d["state"], want)                               # concatenated shard state == single run


def test_single_process_gather_is_identity():
    from mrsgym_amd.dist import ObsAllGather
    g = ObsAllGather(3, 4, 6, "cpu")
    x = torch.randn(3, 4, 6)
    assert torch.equal(g.gather(x), x)
    g = ObsAllGather(3, 4, 6, "cpu", mode="direct", every=3)
    assert g.gather(x) is None and g.gather(x) is None and torch.equal(g.gather(x), x) and torch.equal(g.gather(x + 1), x)
    with pytest.raises(ValueError):
        ObsAllGather(3, 4, 6, "cpu", mode="ring")
