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
Before anything is timed the swarm is rolled in for ROLLIN untimed steps (default 700, key `rollin_steps`) to
the workload's steady state -- a part of the swarm on the ground -- so that a short run (--steps 20) times the
same regime as a long one and as the rocprofv3 summaries under profiles/.
Weak scaling: every GPU owns 4096 envs.  N = 1: `value` is the sharded path itself.  N > 1: `value` is BASELINE
config 5's definition -- every step is followed by the RCCL all-gather of the joint observation tensor (side
stream, double-buffered, back-pressured); the same K steps without the exchange are reported beside it as
`no_exchange`, with the xGMI floor of the gather.

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel against HBM (algorithmic bytes,
SURVEY.md 8d: 268 B per agent-step at this config); `cpu_baseline` times the CPU oracle (a C port of
the reference's algorithm, OpenMP over envs) on a bounded sample of the same workload.
"""
import argparse
import gc
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
    return {"value": E * N_AGENTS * n / dt, "unit": "agent-steps/s", "cores": cores, "kind": "port", "leg": "(b)",
            "sample": "%d envs x %d agents x %d steps of the same workload (C oracle, OpenMP over envs, %.1f s)" % (E, N_AGENTS, n, dt)}


def literal_reference_leg():
    """SURVEY.md 8d / BASELINE.md section 4, leg (a): the untouched reference needs `pybullet` and `gym` (setup.py:5) AND its
    own source tree.  Probed, never installed or fetched; the reference tree does not travel to the GPU box, so even with
    both importable this leg can only say so."""
    missing = []
    for mod in ("pybullet", "gym"):
        try:
            __import__(mod)
        except Exception as e:                                   # ImportError, or a broken install
            missing.append("%s (%s)" % (mod, type(e).__name__))
    try:
        import mrsgym  # noqa: F401  (an installed copy of the reference package, if the box happens to have one)
        have_ref = True
    except Exception:
        have_ref = False
    if missing or not have_ref:
        return {"leg": "(a)", "status": "unavailable",
                "why": "not importable on this box: " + ", ".join(missing + ([] if have_ref else ["mrsgym (the reference package itself)"]))}
    # both importable and an installed mrsgym: time the README example (N=3) and N=64 x 1 env, 200 steps, one process
    import time as _t
    out = {"leg": "(a)", "status": "measured", "cores": 1}
    try:
        import gym
        for n_agents, key in ((3, "n3"), (64, "n64")):
            env = gym.make('mrs-v0', state_fn=lambda quad: torch.cat([quad.get_pos(), quad.get_vel()]), N_AGENTS=n_agents,
                           ACTION_TYPE='set_target_vel', HEADLESS=True,
                           **({} if n_agents <= 32 else {"START_POS": torch.from_numpy(__import__("util_scenarios").grid_spawn(1, n_agents)[0][0])}))
            act = torch.zeros(n_agents, 3)
            t0 = _t.perf_counter()
            for _ in range(200):
                env.step(act)
            out[key + "_agent_steps_per_s_per_core"] = n_agents * 200 / (_t.perf_counter() - t0)
            env.close()
    except Exception as e:
        out = {"leg": "(a)", "status": "unavailable", "why": "reference raised %s: %s" % (type(e).__name__, e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--rollin", type=int, default=int(os.environ.get("MRS_BENCH_ROLLIN", "700")),
                    help="untimed steps before the warm-up: brings the swarm to the workload's steady state")
    ap.add_argument("--no-dense-a", action="store_true", help="skip the extra dense-adjacency leg (N=1)")
    ap.add_argument("--dense-a", action="store_true", help="materialise the float32 (E,K+1,N,N) adjacency the reference returns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--double-buffer", action="store_true", help="also run the two-half-swarms-on-two-streams leg (N=1; opt-in: it "
                    "no longer beats the one-launch step, DESIGN.md section 6)")
    ap.add_argument("--no-double-buffer", action="store_true", help="(accepted, no effect: the leg is opt-in since round 4)")
    ap.add_argument("--no-model-legs", action="store_true", help="skip the `literal` and `solver6` legs beside `value` (N=1)")
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

    def make_env(a_format, lo=0, hi=None, atype=ATYPE, **model):
        """**model: the fidelity knobs of the product (SOLVER_ITERS, ROUND_EULER_READBACK, REST_SHORTCUT, QUAD_CONTACT); none =
        the library's defaults, which is what `value` is measured with."""
        hi = E if hi is None else hi
        env = mrsgym_amd.make('mrs-v0', N_ENVS=hi - lo, N_AGENTS=N, state_fn=state_fn, K_HOPS=K_HOPS, COMM_RANGE=COMM_RANGE,
                              RETURN_A=True, ACTION_TYPE=atype, HEADLESS=True, START_POS=torch.from_numpy(pos[lo:hi]),
                              A_FORMAT=a_format, ENV_INDEX_BASE=base + lo, DEVICE=str(dev), CHECK_NAN="lazy",
                              HISTORY_SLOTS=int(os.environ.get("MRS_BENCH_HISTORY_SLOTS", "0")), **model)
        env.reset(ori=torch.from_numpy(eul[lo:hi]))
        return env

    acts = ActionStream(ATYPE, E, N, pos, seed=1000 + rank)
    total = args.warmup + args.steps
    n_regions = 3 if world == 1 else 2
    table = [torch.from_numpy(acts(50 * k)).to(dev) for k in range((args.rollin + n_regions * total) // 50 + 2)]
    # Device warm-up on a SCRATCH swarm, run immediately before every measured swarm starts (its roll-in follows with no
    # host-side gap): after the GPU has been idle for tens of milliseconds -- process start, but also the allocations
    # and the reset of a new env -- this GPU runs the next 30-50 ms of kernels up to 20 % slow (rocprofv3 kernel
    # durations 26.7 us before a 52 ms gap, 29 -> 33 us over the following 100 launches, back to 26.5 after ~1000), and
    # the first process on a fresh box more so.  The measured swarm is created BEFORE its warm-up for that reason.
    prewarm_s = float(os.environ.get("MRS_BENCH_PREWARM_S", "1.0"))
    # MRS_BENCH_PREWARM_ATYPE (tools/profile_round.sh: set_target_pos): the scratch swarm on another instantiation of the step
    # kernel, so that under rocprofv3 the rows of `k_step<set_target_vel>` hold the measured swarm's launches only -- at the
    # clock the sustained load settles at (per-wave shader-clock stamps: the same 37 k ticks per wave take 20.8 us 200 launches
    # after an idle start and 17.9 us after 800; without the warm-up a profile of 1800 launches is a profile of that ramp)
    scratch = make_env("packed", atype=os.environ.get("MRS_BENCH_PREWARM_ATYPE", ATYPE)) if prewarm_s > 0 else None

    def warm(seconds):
        if scratch is None:
            return
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for t in range(500):
                scratch.step(table[0])
            torch.cuda.synchronize()
    gather = mdist.ObsAllGather(E, N, 6, dev) if world > 1 else None
    # What the collective layer saw: gathered once, before anything is timed, so that a scaling run can be checked
    # for "RCCL saw N ranks, one device each" from the JSON line alone.
    comm_info = None
    if world > 1:
        try:
            mine = {"rank": rank, "device": str(dev), "device_name": torch.cuda.get_device_name(dev), "pid": os.getpid(),
                    "visible_devices": torch.cuda.device_count()}
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            try:
                ver = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as exc:   # noqa: BLE001
                ver = "unavailable (%s)" % type(exc).__name__
            comm_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_devices": [e["device"] for e in everyone],
                         "ranks": everyone, "rccl_version": ver, "distinct_devices": len({e["device"] for e in everyone}),
                         "obs_allgather_mode": gather.mode, "single_device_rehearsal": single}
        except Exception as exc:       # noqa: BLE001 -- reported, never fatal
            comm_info = {"error": "%s: %s" % (type(exc).__name__, exc)}
    # HIP events on the stream the step kernels are launched on (torch's current stream) bracket SPANS of
    # EV_SPAN consecutive mrs_step launches, one span every EV_EVERY steps: on this stack a timing-event pair
    # costs the stream ~60 us (measured 137 us per step with a pair on every step against 76 us with none),
    # so the per-launch duration is sampled and the pair's cost amortised over the span.
    # Round 3: the spans cover the whole timed region (50 of every 50 launches; the driver's 20-step form: one span of 20).  With
    # spans of 10 the average came out 0.3 - 0.4 us ABOVE the wall-clock time per step of the same region -- the two records'
    # own cost lands inside the span and was divided by 10 (tools/probes/ev_span_test.sh: span 10: kernel 22.80 / step 22.46 us;
    # span 25 ... 50: 22.29 / 22.28, 22.35 / 22.50) -- and a rocprofv3 kernel trace of the same run sat 0.9 us below it.
    # Round 5: a span's opening event is recorded AFTER the span's first launch and the span is divided by the EV_SPAN - 1 launches
    # it brackets.  Recording it first put its host cost (12 - 18 us on an idle queue, step-by-step stamps under
    # MRS_BENCH_DEBUG_STEPS=1) between the synchronize and the first launch of the timed region: 0.85 us per step of `value` in
    # the 20-step form, and the first launch's own submission (9 us with the GPU idle) inside the span.  Behind a launch the record
    # is hidden by the running kernel and its timestamp is that kernel's end.  It is recorded after the span's THIRD launch (EV_OPEN):
    # the first record after a synchronize costs the host ~20 us, and behind one queued kernel (22 us) the GPU still ran dry for
    # ~10 us; behind three the queue holds two kernels more.  A span brackets EV_SPAN - EV_OPEN launches.
    EV_EVERY = int(os.environ.get("MRS_BENCH_EVENT_EVERY", "50"))
    EV_SPAN = max(2, min(int(os.environ.get("MRS_BENCH_EVENT_SPAN", "50")), EV_EVERY, args.steps))
    EV_OPEN = 3 if EV_SPAN >= 10 else 1

    def rollin(env):
        """ROLLIN untimed steps from the spawn state: the workload's steady state (a part of the swarm grounded).
        Returns the grounded share as a DEVICE scalar, read after the timed region: no host wait between roll-in and
        warm-up.  (After an idle gap of tens of milliseconds this GPU runs the next ~30 ms of kernels 10-20 % slow --
        rocprofv3 kernel durations 27 -> 33 us across such a gap, decaying over ~1000 launches; the timed steps would
        be measuring that ramp.)"""
        for t in range(args.rollin):
            env.step(table[t // 50])
        return (env.shard.pos[2] < 0.6).float().mean()

    def timed_region(env, t_first, with_gather, read_A=False):
        """W warm-up steps, barrier + synchronize, EXACTLY K timed steps, synchronize + barrier; max over ranks.
        read_A: the consumer looks at info["A"] every step (A_FORMAT = "dense" materialises the float32 stack only for one that does)."""
        shard_step = env.shard.step_ptr

        def one_step(t):
            out = env.step(table[t // 50])
            if read_A:
                out[3]["A"].materialize()
            if with_gather:
                gather.gather(env._Xring.newest())
        # everything the timed loop needs exists before the warm-up, so that nothing but the barrier + synchronize
        # stands between the last warm-up step and the first timed one
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(max(1, (args.steps - EV_SPAN) // EV_EVERY + 1))]
        # the warm-up goes through what the timed steps go through, timing events included: on a fresh box the first
        # hipEventRecord of a process runs library code that is not paged in yet
        warm_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        warm_ev[0].record()
        for a_, b_ in ev:      # torch creates the HIP event at its first record(): here, not inside the timed steps
            a_.record()
            b_.record()
        for t in range(t_first, t_first + args.warmup):
            one_step(t)
        warm_ev[1].record()
        if with_gather:
            gather.wait()

        def timed_step(*a, **k):
            i = timed_step.i
            timed_step.i = i + 1
            j, r = divmod(i, EV_EVERY)
            if DBG: tt = [time.perf_counter()]
            shard_step(*a, **k)
            if DBG: tt.append(time.perf_counter())
            if j < len(ev) and r == EV_OPEN - 1:
                ev[j][0].record()
            if DBG: tt.append(time.perf_counter())
            if j < len(ev) and r == EV_SPAN - 1:
                ev[j][1].record()
            if DBG: tt.append(time.perf_counter()); dbg.append(tt)
        timed_step.i = 0
        DBG = bool(os.environ.get("MRS_BENCH_DEBUG_STEPS")); dbg = []
        if not os.environ.get("MRS_BENCH_DEBUG_NOSYNC"):   # diagnostic only: the contract requires this synchronize
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        env.shard.step_ptr = timed_step
        t0 = time.perf_counter()
        tp, host_max = t0, 0.0
        for t in range(t_first + args.warmup, t_first + total):
            one_step(t)
            tn = time.perf_counter()
            if tn - tp > host_max:                  # the longest single pass of the launch loop: a host stall shows here
                host_max, timed_region.host_max_at = tn - tp, t - t_first - args.warmup
            tp = tn
        host_elapsed = tp - t0                      # launch loop only: equals `elapsed` when the host is the limit
        timed_region.host_max = host_max
        if with_gather:
            gather.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        env.shard.step_ptr = shard_step
        env.check_errors()
        if DBG:
            print("t0->", " | ".join("%.1f %.1f %.1f %.1f" % tuple((x - t0) * 1e6 for x in tt) for tt in dbg[:3] + dbg[-2:]), "end %.1f" % (elapsed * 1e6), file=sys.stderr)
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / (EV_SPAN - EV_OPEN)
        if args.steps < 2:          # no second launch to close a span with: the synchronized wall time of the one step
            kernel_ms = elapsed * 1e3
        if os.environ.get("MRS_BENCH_DEBUG_SPANS"):   # diagnostic: the sampled spans in order (us per launch)
            print("spans:", " ".join("%.1f" % (a.elapsed_time(b) / (EV_SPAN - EV_OPEN) * 1e3) for a, b in ev), file=sys.stderr, flush=True)
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, host_elapsed, kernel_ms

    def model_of(e):
        prm = e.sim.params
        return {"solver_iters": int(prm.solver_iters), "round_euler_readback": int(prm.round_euler_readback), "pair_contact": int(prm.pair_contact),
                "rest_shortcut": int(prm.rest_shortcut), "enable_contact": int(prm.enable_contact),
                "contact_sweeps_dtype": "float32", "state_dtype": "float64"}

    env = make_env("dense" if args.dense_a else "packed")
    model = model_of(env)
    assert env._obs.fused, "cat(pos, vel) must take the fused observation path"
    # the torch kernels of rollin()'s grounded-share expression are loaded here, not at their first use between roll-in and
    # warm-up (a code-object load is ~55 ms of host time with the GPU idle: kernel trace, tools/probes/trace_bench.sh)
    float((env.shard.pos[2] < 0.6).float().mean())
    # no collector pause between here and the end of the timed region (in its 20-step form, 0.5 ms, one pause would be a
    # tenth of it); collected now, while the GPU has nothing queued, not between roll-in and warm-up
    gc.collect()
    gc.disable()
    warm(prewarm_s)
    grounded = rollin(env)
    agent_steps = float(E) * N * args.steps * world
    extra = {}
    if world == 1:
        # `value`: the step itself -- one GPU, nothing to exchange
        elapsed, host_elapsed, kernel_ms = timed_region(env, args.rollin, False, read_A=args.dense_a)
        host_max, host_max_at = timed_region.host_max, timed_region.host_max_at
        if not args.no_dense_a and not args.dense_a:
            # the reference's return format: float32 0/1 (E,K+1,N,N) adjacency materialised every step (a second launch)
            del env
            denv = make_env("dense")
            warm(min(prewarm_s, 0.3))
            rollin(denv)
            d_elapsed, _, _ = timed_region(denv, args.rollin, False, read_A=True)
            extra["dense_a"] = {"value": agent_steps / d_elapsed, "unit": "agent-steps/s", "ms_per_step": d_elapsed / args.steps * 1e3,
                                "what": "same K steps with info['A'] as the dense float32 (E,K+1,N,N) tensor the reference returns "
                                        "(516 B per agent-step algorithmic, SURVEY.md 8d) instead of bit-packed rows, READ every step"}
            # the library's default construction (A_FORMAT = "dense") in a loop that never looks at info["A"]: the float32 stack is
            # materialised on first use only (mrsgym_amd/lazy.py), so this must cost what the packed rows cost
            del denv
            denv = make_env("dense")          # a fresh one: nobody has ever looked at its info["A"]
            warm(min(prewarm_s, 0.3))
            rollin(denv)
            u_elapsed, _, _ = timed_region(denv, args.rollin, False, read_A=False)
            extra["dense_a_unread"] = {"value": agent_steps / u_elapsed, "unit": "agent-steps/s", "ms_per_step": u_elapsed / args.steps * 1e3,
                                       "what": "A_FORMAT='dense' (the default), info['A'] never touched: packed rows only, the float32 stack is lazy"}
            del denv
        if not args.no_model_legs and not args.dense_a:
            # The headline must not hang on the fidelity knobs: the same K steps with every knob at its literal
            # setting, and with the sweep cap of round 3's headline, each with its own event-timed kernel duration.
            for key, knobs, what in (
                    ("literal", dict(SOLVER_ITERS=10, ROUND_EULER_READBACK=True, REST_SHORTCUT=False),
                     "every fidelity knob at its literal setting: 10 contact sweeps, the controller's float32 rounding of the Euler "
                     "read-back (Object.py:97 -> QuadControl.py:99), every body near the ground through the sweeps (no closed forms for flat bodies)"),
                    ("solver6", dict(SOLVER_ITERS=6),
                     "library defaults except a cap of 6 contact sweeps (round 3's headline setting; accuracy: tests/golden/F6c)")) + (
                    (("default_again", dict(), "diagnostic (MRS_BENCH_REPEAT_DEFAULT=1): the library defaults once more, as the last leg"),)
                    if os.environ.get("MRS_BENCH_REPEAT_DEFAULT") == "1" else ()):
                lenv = make_env("packed", **knobs)
                warm(min(prewarm_s, 0.3))
                rollin(lenv)
                l_elapsed, _, l_kernel_ms = timed_region(lenv, args.rollin, False)
                l_ach = ALGO_BYTES_PER_AGENT_STEP * E * N / (l_kernel_ms * 1e-3) / 1e9
                extra[key] = {"value": agent_steps / l_elapsed, "unit": "agent-steps/s", "ms_per_step": l_elapsed / args.steps * 1e3,
                              "kernel_ms": l_kernel_ms, "roofline_frac": l_ach / HBM_PEAK_GBS, "fidelity": model_of(lenv), "what": what}
                del lenv
        if args.double_buffer and E % 2 == 0:
            # BESIDE `value`, never instead of it: the same swarm as two half-swarms on two streams, each stepping on its own
            # (the EnvPool / Sample-Factory pattern: a closed loop may use it, the policy for half A runs while half B steps).
            # The two kernels overlap, half A's tail and hand-off bubbles under half B's forces phase and vice versa.
            # Driven at the C-ABI level (SwarmShard.step_ptr, fixed observation / adjacency buffers): two env.step() calls per round
            # cost 2 x ~13 us of Python, more than the two kernels take (measured: 29.5 us per round through env.step()).
            from mrsgym_amd.native import ACT as _ACT
            halves = []
            for lo, hi in ((0, E // 2), (E // 2, E)):
                sh = mrsgym_amd.SwarmShard(hi - lo, N, dev)
                zz = np.zeros((hi - lo, N, 3), np.float32)
                sh.set_state(pos=pos[lo:hi], ori=eul[lo:hi], vel=zz, angvel=zz)
                halves.append((sh, torch.zeros(hi - lo, N, sh.D, device=dev), torch.zeros(hi - lo, N, sh.W, dtype=torch.int64, device=dev),
                               [t_[lo:hi].contiguous() for t_ in table], torch.cuda.Stream(device=dev)))
            at = _ACT[ATYPE]

            def round_(t):
                for sh, o_, a_, tb, st_ in halves:
                    with torch.cuda.stream(st_):
                        sh.step_ptr(tb[t // 50], at, o_.data_ptr(), a_.data_ptr(), COMM_RANGE)
            torch.cuda.synchronize()
            for t in range(args.rollin + args.warmup):
                round_(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(args.rollin + args.warmup, args.rollin + total):
                round_(t)
            torch.cuda.synchronize()
            db_elapsed = time.perf_counter() - t0
            extra["double_buffered"] = {"value": agent_steps / db_elapsed, "unit": "agent-steps/s", "ms_per_step": db_elapsed / args.steps * 1e3,
                                        "what": "two half-swarms of %d envs on two HIP streams, free-running (each half's step t+1 follows its own "
                                                "step t; the halves are never joined), driven through the C-ABI (mrs_step) with fixed output buffers: "
                                                "K rounds = K steps of both halves; reported beside the synchronous `value`, which is one launch "
                                                "over all envs per env.step()" % (E // 2)}
            del halves
    else:
        # `value`: BASELINE config 5 / north_star -- every step followed by the RCCL all-gather of the joint observation
        elapsed, host_elapsed, kernel_ms = timed_region(env, args.rollin, True)
        host_max, host_max_at = timed_region.host_max, timed_region.host_max_at
        # the kernel's own duration for the roofline object comes from the region without the gather: with it the event
        # spans on the step stream include the back-pressure waits for the side stream
        n_elapsed, _, kernel_ms = timed_region(env, args.rollin + total, False)
        per_rank = E * N * 6 * 4
        extra["no_exchange"] = {"value": agent_steps / n_elapsed, "unit": "agent-steps/s", "ms_per_step": n_elapsed / args.steps * 1e3,
                                "what": "the same K steps without the all-gather: step() itself has no exchange between shards"}
        # A consumer that needs the joint observation every k-th step only (ObsAllGather(every=k)), and the one-shot form of the
        # exchange (grouped point-to-point operations) -- beside `value`, each in its own timed region;
        # a leg that cannot run on this box says why instead of taking the bench down
        # (the one-shot form has never met more than one rank on hardware: it runs only when asked for, MRS_BENCH_DIRECT=1, so that
        # an untried collective pattern cannot take the scaling run down with it)
        legs = [("gather_every_4", dict(every=4)), ("gather_every_16", dict(every=16))]
        if os.environ.get("MRS_BENCH_DIRECT") == "1":
            legs.append(("direct_p2p", dict(mode="direct")))
        for key, kw in legs:
            try:
                saved = gather
                gather = mdist.ObsAllGather(E, N, 6, dev, **kw)
                g_elapsed, _, _ = timed_region(env, args.rollin, True)
                extra[key] = {"value": agent_steps / g_elapsed, "unit": "agent-steps/s", "ms_per_step": g_elapsed / args.steps * 1e3,
                              "what": {"gather_every_4": "the joint observation gathered on every 4th step only",
                                       "gather_every_16": "the joint observation gathered on every 16th step only",
                                       "direct_p2p": "every step, as one grouped batch of point-to-point sends / receives (the one-shot form)"}[key]}
            except Exception as exc:          # noqa: BLE001 -- reported, never fatal
                extra[key] = {"value": None, "error": "%s: %s" % (type(exc).__name__, exc)}
            finally:
                gather = saved
        floor_ms = per_rank * (world - 1) / (min(world - 1, 7) * 76e9) * 1e3
        one_gpu_step_ms = n_elapsed / args.steps * 1e3
        extra["obs_allgather"] = {"bytes_sent_per_rank_per_step": per_rank * (world - 1), "bytes_received_per_rank_per_step": per_rank * (world - 1),
                                  "xgmi_floor_ms": floor_ms,
                                  # what DESIGN.md section 8 predicts for this N from the link arithmetic alone, so that a measured line can be
                                  # checked against it: the step hides under the gather once the gather is the longer of the two
                                  "predicted": {"value": float(E) * N * world / (max(floor_ms, one_gpu_step_ms) * 1e-3),
                                                "no_exchange": float(E) * N * world / (one_gpu_step_ms * 1e-3),
                                                "xgmi_floor_ms": floor_ms,
                                                "what": "value = E*N*world / max(xGMI floor, measured no-exchange step); at N = 8: 44 MB in per rank per step "
                                                        "over 7 links x ~76 GB/s >= 83 us against a ~25 us step => ~2.5e10 agent-steps/s (~2.5x one GPU), "
                                                        "no_exchange ~8x"},
                                  "what": "newest observation slice (E_local,N,6) float32 to every rank each step (RCCL, side stream, "
                                          "double-buffered); floor = bytes received / (links driven x ~76 GB/s per xGMI link direction)"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = agent_steps / elapsed
    algo_bytes_launch = ALGO_BYTES_PER_AGENT_STEP * E * N
    achieved = algo_bytes_launch / (kernel_ms * 1e-3) / 1e9
    # HBM traffic and VALU occupancy need hardware counters: taken from the committed rocprofv3 --pmc passes of this
    # workload (profiles/README.md says which commit), never measured inside this process
    traffic = valu_busy = prof_src = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp) and E == ENVS_PER_GPU and not args.dense_a:
        try:
            tj = json.load(open(tp))
            traffic, valu_busy, prof_src = tj.get("mrs_step_bytes_per_launch"), tj.get("valu_busy"), tj.get("source")
        except Exception:
            pass
    out = {
        "metric": "agent-steps/sec (whole node) at N_AGENTS=64 x4096 envs", "value": value, "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rollin_steps": args.rollin,
        "grounded_fraction_after_rollin": float(grounded), "ms_per_step": elapsed / args.steps * 1e3,
        "host_ms_per_step": host_elapsed / args.steps * 1e3, "host_ms_longest_step": host_max * 1e3, "host_longest_step_index": host_max_at,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "precision": "float64 rigid-body state, controller and force assembly (as Bullet / numpy in the reference); float32 where the "
                     "reference is float32 (read-backs, downwash pair terms, adjacency distances) and, the build's own choice, in the "
                     "ground-contact sweeps and the quad-quad contact terms (velocity changes added to the float64 state; tested against "
                     "the float64 oracle at 1e-4 per step)",
        "config": {"workload": "N_AGENTS=64 x %d envs/GPU, ACTION_TYPE=set_target_vel (PID), RETURN_A=True COMM_RANGE=5.0, "
                               "K_HOPS=3, state_fn=cat(pos,vel), A %s%s" % (E, "dense fp32" if args.dense_a else "bit-packed",
                                                                             ", joint observation all-gathered every step" if world > 1 else ""),
                   "n_agents": N, "n_envs_per_gpu": E, "k_hops": K_HOPS, "comm_range": COMM_RANGE,
                   # the product's fidelity knobs as `value` was measured (the library's defaults; `literal` / `solver6` beside it)
                   "fidelity": model,
                   "parallelism": "env-sharded x%d, RCCL all-gather of the newest observation slice per step" % world if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "k_step<set_target_vel>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": algo_bytes_launch,
                     "limiter": "the instruction stream itself: ~2000 vector instructions per wave, a third of them float64, at 77 % VALU occupancy with 4 waves per SIMD (a lone wave needs 12 us for the step; the many-round asymptote is 21 us per 4096 envs) -- not HBM bandwidth, never MFMA: see DESIGN.md section 3", "valu_busy": valu_busy,
                     "counters_from": prof_src},
    }
    out.update(extra)
    if comm_info is not None:
        out["comm"] = comm_info
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline()
        # the literal reference cannot run here (pybullet absent); its Python controller alone
        # was measured at 305 us per agent-step on one core => an upper bound on what the reference itself could reach
        cb["literal_reference"] = literal_reference_leg()
        cb["reference_python_bracket"] = {"leg": "(c)", "us_per_agent_step_controller_only": 305.0, "agent_steps_per_s_per_core": 1e6 / 305.0,
                                          "agent_steps_per_s_all_cores": cb["cores"] * 1e6 / 305.0,
                                          "what": "kind (c) of SURVEY.md 8d: lower bound on the reference's cost (QuadControl Python only, "
                                                  "no Bullet, no downwash loop); the reference is single-threaded"}
        out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
