#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the MRS.step() hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by torch.distributed.run, one rank per GPU; reads RANK/LOCAL_RANK/WORLD_SIZE)

Workload (BASELINE.json configs[2], the one the metric is quoted on): N_AGENTS=64 x 4096 envs per GPU,
ACTION_TYPE=set_target_vel (cascaded PID), RETURN_A=True, COMM_RANGE=5.0, K_HOPS=3, state_fn =
cat(pos, vel); synthetic swarm of SURVEY.md 8d (grid spawn, per-agent U[-1,1]^3 m/s targets held 50
steps, pre-generated on the device).  One "step" = one env.step(actions) of the product API over the
whole batch: fused controller + rotor/aero forces + downwash + 6-DoF integration + ground contact +
newest observation slice + bit-packed newest adjacency rows, written into the K_HOPS history ring.
Weak scaling: every GPU owns 4096 envs and step() has no exchange between them, so `value` times the
sharded path alone; with N>1 the same K steps are then repeated with the joint observation tensor
all-gathered over RCCL every step (side stream, double-buffered) and reported as `with_obs_allgather`.

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel against HBM (algorithmic bytes,
SURVEY.md 8d: 268 B per agent-step at this config); `cpu_baseline` times the CPU oracle (a C port of
the reference's algorithm, OpenMP over envs) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mrs-gym_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

N_AGENTS, ENVS_PER_GPU, K_HOPS, COMM_RANGE, ATYPE = 64, 4096, 3, 5.0, "set_target_vel"
# state r+w 104, action 12, PID state (set_target_vel, 15 words r+w) 120, obs slice 24, packed adjacency row 8
ALGO_BYTES_PER_AGENT_STEP = 104 + 12 + 120 + 24 + 8
HBM_PEAK_GBS = 8000.0


def state_fn(quad):  # README.md:28-29
    return torch.cat([quad.get_pos(), quad.get_vel()])


def cpu_baseline(seconds=15.0):
    """The CPU oracle (kind "port") on 32 envs per thread of the same workload: ~15 s of work on the host cores."""
    import oracle
    from util_scenarios import ActionStream, grid_spawn
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MRS_CPU_BASELINE_THREADS", "16"))))   # a 1-GPU box's CPU share
    E = 32 * cores
    pos, eul = grid_spawn(E, N_AGENTS)
    z = np.zeros((E, N_AGENTS, 3))
    sw = oracle.OracleSwarm(E, N_AGENTS, nthreads=cores)
    sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z, angvel=z)
    acts = ActionStream(ATYPE, E, N_AGENTS, pos, seed=1)
    a = acts(0)
    sw.step_full(a, ATYPE, COMM_RANGE)          # thread-pool warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        if n % 50 == 0:
            a = acts(n)
        sw.step_full(a, ATYPE, COMM_RANGE)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 6000:
            break
    return {"value": E * N_AGENTS * n / dt, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d agents x %d steps of the same workload (C oracle, OpenMP over envs, %.1f s)" % (E, N_AGENTS, n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--dense-a", action="store_true", help="materialise the float32 (E,K+1,N,N) adjacency the reference returns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import mrsgym_amd
    from mrsgym_amd import dist as mdist
    from util_scenarios import ActionStream, grid_spawn
    import torch.distributed as dist

    rank, world, local = mdist.init_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # MRS_BENCH_SINGLE_DEVICE=1 (+ MRS_DIST_BACKEND=gloo): rehearse the N>1 code path on a one-GPU box
    single = os.environ.get("MRS_BENCH_SINGLE_DEVICE") == "1"
    dev = torch.device("cuda", local if (world > 1 and not single) else 0)
    torch.cuda.set_device(dev)
    E, N = args.envs_per_gpu, N_AGENTS
    base = rank * E
    pos, eul = grid_spawn(E, N, env_base=base)

    def make_env():
        env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K_HOPS, COMM_RANGE=COMM_RANGE,
                              RETURN_A=True, ACTION_TYPE=ATYPE, HEADLESS=True, START_POS=torch.from_numpy(pos),
                              A_FORMAT="dense" if args.dense_a else "packed", ENV_INDEX_BASE=base, DEVICE=str(dev),
                              CHECK_NAN="lazy")
        env.reset(ori=torch.from_numpy(eul))
        return env

    acts = ActionStream(ATYPE, E, N, pos, seed=1000 + rank)
    total = args.warmup + args.steps
    table = [torch.from_numpy(acts(50 * k)).to(dev) for k in range((2 * total if world > 1 else total) // 50 + 1)]
    # Process/device warm-up on a SCRATCH swarm, before the W warm-up steps of the measured one: the first
    # ~0.1 s of launches of a fresh process run ~20 % slow (62 vs 52 us/step measured; clock ramp + the HIP
    # runtime growing its signal/kernarg pools), and on a fresh box the first process is slower still.  The
    # measured swarm then starts from its spawn state exactly as the workload definition says.
    prewarm_s = float(os.environ.get("MRS_BENCH_PREWARM_S", "1.0"))
    if prewarm_s > 0:
        scratch = make_env()
        t_end = time.perf_counter() + prewarm_s
        while time.perf_counter() < t_end:
            for t in range(500):
                scratch.step(table[0])
            torch.cuda.synchronize()
        del scratch
    env = make_env()
    assert env._obs.fused, "cat(pos, vel) must take the fused observation path"
    gather = mdist.ObsAllGather(E, N, 6, dev) if world > 1 else None

    def one_step(t, with_gather):
        X, r, d, info = env.step(table[(t // 50) % len(table)])
        if with_gather:
            gather.gather(env._Xring.newest())
        return X, info

    # HIP events on the stream the step kernels are launched on (torch's current stream) bracket SPANS of
    # EV_SPAN consecutive mrs_step launches, one span every EV_EVERY steps: on this stack a timing-event pair
    # costs the stream ~60 us (measured 137 us per step with a pair on every step against 76 us with none),
    # so the per-launch duration is sampled and the pair's cost amortised over the span.
    EV_EVERY = int(os.environ.get("MRS_BENCH_EVENT_EVERY", "50"))
    EV_SPAN = max(1, min(int(os.environ.get("MRS_BENCH_EVENT_SPAN", "10")), EV_EVERY, args.steps))
    shard_step = env.shard.step_ptr

    def timed_region(t_first, with_gather):
        """W warm-up steps, barrier + synchronize, EXACTLY K timed steps, synchronize + barrier; max over ranks."""
        for t in range(t_first, t_first + args.warmup):
            one_step(t, with_gather)
        if with_gather:
            gather.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(max(1, (args.steps - EV_SPAN) // EV_EVERY + 1))]

        def timed_step(*a, **k):
            i = timed_step.i
            timed_step.i = i + 1
            j, r = divmod(i, EV_EVERY)
            if j < len(ev) and r == 0:
                ev[j][0].record()
            shard_step(*a, **k)
            if j < len(ev) and r == EV_SPAN - 1:
                ev[j][1].record()
        timed_step.i = 0
        env.shard.step_ptr = timed_step
        t0 = time.perf_counter()
        for t in range(t_first + args.warmup, t_first + total):
            one_step(t, with_gather)
        host_elapsed = time.perf_counter() - t0     # launch loop only: equals `elapsed` when the host is the limit
        if with_gather:
            gather.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        env.shard.step_ptr = shard_step
        env.check_errors()
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / EV_SPAN
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, host_elapsed, kernel_ms

    # `value`: the sharded path itself -- envs are independent, so step() has no exchange and none is timed.
    elapsed, host_elapsed, kernel_ms = timed_region(0, False)
    # N > 1, reported beside it: the same K steps with the joint observation tensor all-gathered over RCCL every
    # step (side stream, double-buffered), for consumers that want every rank to hold all of X.
    gathered = None
    if world > 1 and os.environ.get("MRS_BENCH_OBS_ALLGATHER", "1") == "1":
        g_elapsed, _, _ = timed_region(total, True)
        gathered = {"value": float(E) * N * args.steps * world / g_elapsed, "unit": "agent-steps/s",
                    "ms_per_step": g_elapsed / args.steps * 1e3,
                    "bytes_sent_per_rank_per_step": E * N * 6 * 4 * (world - 1),
                    "what": "newest observation slice (E_local,N,6) float32 all-gathered to every rank each step (RCCL)"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    agent_steps = float(E) * N * args.steps * world
    value = agent_steps / elapsed
    algo_bytes_launch = ALGO_BYTES_PER_AGENT_STEP * E * N
    achieved = algo_bytes_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")     # written from the rocprofv3 --pmc passes (see profiles/README.md)
    if os.path.exists(tp) and E == ENVS_PER_GPU and not args.dense_a:   # counters were collected on the default workload
        try:
            traffic = json.load(open(tp)).get("mrs_step_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "agent-steps/sec (whole node) at N_AGENTS=64 x4096 envs", "value": value, "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "host_ms_per_step": host_elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "N_AGENTS=64 x %d envs/GPU, ACTION_TYPE=set_target_vel (PID), RETURN_A=True COMM_RANGE=5.0, "
                               "K_HOPS=3, state_fn=cat(pos,vel), A %s" % (E, "dense fp32" if args.dense_a else "bit-packed"),
                   "n_agents": N, "n_envs_per_gpu": E, "k_hops": K_HOPS, "comm_range": COMM_RANGE,
                   "parallelism": "env-sharded x%d, no exchange inside step()" % world if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "k_step<set_target_vel>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": algo_bytes_launch},
    }
    if gathered is not None:
        out["with_obs_allgather"] = gathered
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
