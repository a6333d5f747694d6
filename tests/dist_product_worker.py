"""Worker of tests/test_gpu_dist.py (launched by torch.distributed.run, one process per rank): the PRODUCT's sharded path.

Each rank owns the contiguous block of envs mrsgym_amd.dist.shard_range gives it, as a SwarmShard on its GPU (rank r on
cuda:r under RCCL when the box has that many devices; otherwise every rank on cuda:0 with gloo carrying the collective --
the one-GPU rehearsal), steps it with the global-index-seeded actions, all-gathers the newest observation slice every step
through ObsAllGather (side stream, double-buffered) and finally the shard states.  Rank 0 writes what it gathered."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mrs-gym_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.distributed as dist


def main():
    out_dir, e_total, n, steps, atype = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import mrsgym_amd
    from mrsgym_amd import dist as mdist
    from util_scenarios import ActionStream, grid_spawn
    world = int(os.environ["WORLD_SIZE"])
    multi = torch.cuda.device_count() >= world
    rank, world, local = mdist.init_from_env(backend="nccl" if multi else "gloo")
    dev = torch.device("cuda", local if multi else 0)
    torch.cuda.set_device(dev)
    lo, hi = mdist.shard_range(e_total, rank, world)
    pos, eul = grid_spawn(e_total, n)                      # the whole swarm's spawn; this rank takes its block
    acts = ActionStream(atype, e_total, n, pos, seed=3)
    sh = mrsgym_amd.SwarmShard(hi - lo, n, dev)
    z = np.zeros((hi - lo, n, 3), np.float32)
    sh.set_state(pos=pos[lo:hi], ori=eul[lo:hi], vel=z, angvel=z)
    ring = [torch.zeros(hi - lo, n, sh.D, device=dev) for _ in range(4)]
    adj = torch.zeros(hi - lo, n, sh.W, dtype=torch.int64, device=dev)
    gather = mdist.ObsAllGather(hi - lo, n, sh.D, dev)
    joint = []
    for t in range(steps):
        a = torch.from_numpy(acts(t)[lo:hi]).to(dev)
        obs = ring[t % 4]
        sh.step(a, atype, obs_out=obs, adj_out=adj, comm_range=2.5)
        out = gather.gather(obs)
        if t % 5 == 4 or t == steps - 1:                   # a consumer that looks at the joint tensor now and then
            gather.wait()
            torch.cuda.current_stream(dev).synchronize()
            joint.append(out.clone().cpu().numpy())
    gather.wait()
    state = torch.cat([sh.view(sh.pos), sh.view(sh.quat), sh.view(sh.vel), sh.view(sh.angvel)], -1).contiguous()
    state = mdist.gather_global_state(state)
    adj_all = mdist.gather_global_state(adj)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist_product.npz"), joint=np.stack(joint), state=state.cpu().numpy(), adj=adj_all.cpu().numpy(),
                 backend=np.array("nccl" if multi else "gloo"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
