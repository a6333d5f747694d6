"""Teacher-forced per-step parity on the literal BASELINE workloads (SURVEY.md section 8d inputs).

The GPU swarm free-runs -- through touchdown, rest, tumbling on the ground and quad-quad contact, wherever the (chaotic)
workload takes it -- and EVERY step the CPU oracle is re-seeded with the GPU's pre-step state (float64 state words and
the controller memory) and takes the same one step: what is compared is one step of the kernel against one step of the
restatement on identical inputs, at the state mix the benchmark times, not two diverging samples of a chaotic system.

Test infrastructure (uses oracle/); shared by tests/test_gpu_teacher.py and tools/teacher_probe.py.
"""
import numpy as np

import oracle
from util_scenarios import ActionStream, grid_spawn

# C2 set_speeds / RETURN_A=False, C3 set_target_vel / A / K=3, C4 set_control / A (N=256), C5 set_target_pos / A
CONFIGS = {
    "C2": dict(N=64, atype="set_speeds", comm_range=None),
    "C3": dict(N=64, atype="set_target_vel", comm_range=5.0),
    "C4": dict(N=256, atype="set_control", comm_range=5.0),
    "C5": dict(N=64, atype="set_target_pos", comm_range=5.0),
    # beyond BASELINE's four: the remaining ACTION_TYPEs and the other code paths of the pair loops (several envs per wave; the ring
    # of an env that spans waves but is not whole 64-agent blocks; three blocks), same literal spawn, same chaotic inputs
    "X12": dict(N=12, atype="set_target_accel", comm_range=2.5),
    "X100": dict(N=100, atype="set_target_ori", comm_range=5.0),
    "X192": dict(N=192, atype="set_target_vel", comm_range=float("inf")),
}
PHASES = ("free", "rest", "listed", "pair")


def pid_to_oracle(pid_records, out):
    """MrsBuffers.pid (5, T, 4) float32 records (include/mrs_hip.h) -> the oracle's per-agent OrcPid array."""
    r = pid_records
    out["integral_pos_e"] = r[0][:, :3]
    out["d_vel_e"] = r[1][:, :3]
    out["integral_vel_e"] = np.stack([r[1][:, 3], r[2][:, 0], r[2][:, 1]], -1)
    out["last_vel_e"] = np.stack([r[2][:, 2], r[2][:, 3], r[3][:, 0]], -1)
    out["last_target_vel"] = r[3][:, 1:4]
    out["integral_ori_e"] = r[4][:, :3]


def classify(pre, P, N):
    """Phase of every body from its PRE-step state (E,N,.): the branch the step takes for it.
    pair: within quad-quad contact range of another agent (float32 positions, the kernel's and the oracle's test);
    rest: near the ground, lying flat and still (the kernel's own-lane shortcut, contact_at_rest: |R20|,|R21|,|w|,|v_xy| < 1e-6);
    listed: near the ground otherwise (the sequential-impulse sweeps); free: none of these."""
    pos, quat, vel, ang = pre["pos"], pre["quat"], pre["vel"], pre["angvel"]
    park_z = P.ground_z + np.sqrt(P.coll_radius ** 2 + P.coll_half_len ** 2) + P.contact_threshold
    near = pos[..., 2] <= park_z
    x, y, z, w = (quat[..., k] for k in range(4))
    r20, r21 = 2 * (x * z - w * y), 2 * (y * z + w * x)
    still = (np.abs(r20) < 1e-6) & (np.abs(r21) < 1e-6) & (np.abs(ang).max(-1) < 1e-6) & (np.abs(vel[..., :2]).max(-1) < 1e-6)
    p32 = pos.astype(np.float32)
    d = p32[:, :, None, :] - p32[:, None, :, :]
    d2 = d[..., 2] * d[..., 2] + (d[..., 1] * d[..., 1] + d[..., 0] * d[..., 0])
    rc = np.float32(2 * P.coll_radius) + np.float32(P.contact_threshold)
    d2[:, np.arange(N), np.arange(N)] = np.inf
    pair = (d2 <= rc * rc * np.float32(1.0001)).any(-1)
    ph = np.zeros(pos.shape[:2], np.int8)          # free
    ph[near & still] = 1
    ph[near & ~still] = 2
    ph[pair] = 3
    return ph


def run(torch, mrsgym_amd, cfg, E, steps, seed=0, check_adj_every=1, params=None, nthreads=8, progress=None, dump=None, unconstrained=False,
        oracle_params=None):
    """Returns dict: err[phase] = array of per-body per-step errors (max over the 13 state words, relative above
    magnitude 1; abs_err[phase]: the same words' largest ABSOLUTE difference), visited counters, adjacency mismatches (must be 0).  unconstrained=True: the oracle takes every step a
    second time with both contact models off, and vunc[phase] holds, aligned with err[phase], the largest velocity word
    (m/s) the contact solve was handed -- what it has to cancel in float32 on the GPU."""
    c = CONFIGS[cfg]
    N, atype, R = c["N"], c["atype"], c["comm_range"]
    pos, eul = grid_spawn(E, N, seed=seed)                      # 1 m pitch, |yaw| <= pi/2: the literal spawn
    z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    if params is not None:
        sh.set_params(params)
    P = sh.params
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    sw = oracle.OracleSwarm(E, N, nthreads=nthreads)
    for k in ("solver_iters", "enable_contact", "pair_contact"):
        setattr(sw.p, k, int(getattr(P, k)))
    for k, v in (oracle_params or {}).items():                  # the oracle at OTHER settings than the kernel (e.g. rest_shortcut = 0)
        setattr(sw.p, k, v)
    acts = ActionStream(atype, E, N, pos, seed=1000 + seed)      # coherent=False: independent per-agent targets
    obs = torch.zeros(E, N, sh.D, device="cuda:0")
    adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0") if R is not None else None
    dense = torch.zeros(E, N, N, device="cuda:0") if R is not None else None

    def grab():
        return {k: sh.view(getattr(sh, k)).cpu().numpy() for k in ("pos", "quat", "vel", "angvel")}

    errs = {p: [] for p in PHASES}
    aerrs = {p: [] for p in PHASES}
    vuncs = {p: [] for p in PHASES}
    worst = {p: (0.0, None) for p in PHASES}
    visited = dict(touchdown=0, rest=0, tumbling=0, pair=0, listed=0, nnls=0)
    adj_bad = 0
    pre = grab()
    was_near = np.zeros((E, N), bool)
    for t in range(steps):
        pid = sh.pid.cpu().numpy()
        a = acts(t)
        sh.step(torch.from_numpy(a).cuda(), atype, obs_out=obs, adj_out=adj, comm_range=R if R is not None else float("nan"))
        post = grab()
        # the oracle: this step from the GPU's pre-step state
        sw.pos[...] = pre["pos"]; sw.quat[...] = pre["quat"]; sw.vel[...] = pre["vel"]; sw.angvel[...] = pre["angvel"]
        pid_to_oracle(pid, sw.pid)
        sw.step(a, atype)
        ph = classify(pre, P, N)
        if unconstrained:
            keep = {k: getattr(sw, k).copy() for k in ("pos", "quat", "vel", "angvel")}
            sw.pos[...] = pre["pos"]; sw.quat[...] = pre["quat"]; sw.vel[...] = pre["vel"]; sw.angvel[...] = pre["angvel"]
            pid_to_oracle(pid, sw.pid)
            flags = (sw.p.enable_contact, sw.p.pair_contact)
            sw.p.enable_contact, sw.p.pair_contact = 0, 0
            sw.step(a, atype)
            sw.p.enable_contact, sw.p.pair_contact = flags
            vu = np.abs(sw.vel).max(-1)
            for k, v in keep.items():
                getattr(sw, k)[...] = v
        e = np.zeros((E, N))
        ea = np.zeros((E, N))                                     # the same, ABSOLUTE (north_star's metric): max |delta| over the 13 words
        for k, ref in (("pos", sw.pos), ("quat", sw.quat), ("vel", sw.vel), ("angvel", sw.angvel)):
            d = np.abs(post[k] - ref)
            e = np.maximum(e, (d / np.maximum(1.0, np.abs(ref))).max(-1))
            ea = np.maximum(ea, d.max(-1))
        e = np.where(np.isfinite(e), e, np.inf)
        ea = np.where(np.isfinite(ea), ea, np.inf)
        if dump is not None:   # diagnostic: the worst contact-phase cases with everything needed to replay them on the host
            thr = dump.get("thr", 5e-5)
            pick = (e > thr) & (ph > 0)
            if unconstrained and "vmax" in dump:       # only bodies whose contact solve was handed less than this (m/s)
                pick &= vu < dump["vmax"]
            for (ee, ii) in zip(*np.nonzero(pick)):
                if len(dump.setdefault("cases", [])) < dump.get("max", 400):
                    dump["cases"].append(dict(t=t, env=int(ee), agent=int(ii), phase=int(ph[ee, ii]), err=float(e[ee, ii]),
                                              pre=np.concatenate([pre[k][ee, ii] for k in ("pos", "quat", "vel", "angvel")]),
                                              gpu=np.concatenate([post[k][ee, ii] for k in ("pos", "quat", "vel", "angvel")]),
                                              orc=np.concatenate([sw.pos[ee, ii], sw.quat[ee, ii], sw.vel[ee, ii], sw.angvel[ee, ii]]),
                                              wrench=sw.wrench[ee, ii].copy(), action=a[ee, ii].copy()))
        for i, p in enumerate(PHASES):
            m = ph == i
            if m.any():
                errs[p].append(e[m])
                aerrs[p].append(ea[m])
                if unconstrained:
                    vuncs[p].append(vu[m])
                j = np.argmax(np.where(m, e, -1))
                if e.flat[j] > worst[p][0]:
                    worst[p] = (float(e.flat[j]), (t, int(j // N), int(j % N)))
        near = ph >= 1
        visited["touchdown"] += int(((ph == 2) & ~was_near & (pre["vel"][..., 2] < -0.1)).sum())
        visited["rest"] += int((ph == 1).sum())
        visited["listed"] += int((ph == 2).sum())
        visited["tumbling"] += int(((ph == 2) & (np.abs(pre["angvel"]).max(-1) > 1.0)).sum())
        visited["pair"] += int((ph == 3).sum())
        was_near = near & (ph != 3) | (was_near & (ph == 3))
        if R is not None and t % check_adj_every == 0:
            sh.adjacency_expand(adj, dense)
            p32 = np.ascontiguousarray(post["pos"].astype(np.float32))
            want = np.stack([oracle.adjacency(p32[k], R) for k in range(E)])
            adj_bad += int((dense.cpu().numpy() != want).sum())
            o = obs.cpu().numpy()
            adj_bad += int((o[..., :3] != p32).sum()) + int((o[..., 3:6] != post["vel"].astype(np.float32)).sum())
        pre = post
        if progress and t % 100 == 99:
            progress(t + 1)
    out = dict(cfg=cfg, E=E, N=N, steps=steps, visited=visited, adj_bad=adj_bad, worst=worst,
               err={p: (np.concatenate(errs[p]) if errs[p] else np.zeros(0)) for p in PHASES},
               abs_err={p: (np.concatenate(aerrs[p]) if aerrs[p] else np.zeros(0)) for p in PHASES},
               vunc={p: (np.concatenate(vuncs[p]) if vuncs[p] else np.zeros(0)) for p in PHASES},
               grounded_share=float((pre["pos"][..., 2] < 0.6).mean()))
    return out


def quantiles(x):
    if x.size == 0:
        return None
    q = np.quantile(x, [0.5, 0.99, 0.999])
    return dict(n=int(x.size), q50=float(q[0]), q99=float(q[1]), q999=float(q[2]), max=float(x.max()))
