"""Offline laboratory for the ground-contact solver (CPU only, no GPU needed).

    python tools/contact_lab.py capture [E] [steps]   -> /tmp/contact_problems.npy
    python tools/contact_lab.py study

capture: runs the bench workload (N = 64, set_target_vel, grid spawn) on the CPU oracle, single-threaded, with the
oracle's diagnostic tap switched on, and keeps the inputs (z, R, v, w) of every contact problem of the last 100 steps.
study: solves the captured problems with numpy restatements of solver variants (vectorised over the problems) and
reports, per variant, how many sweeps a body needs and how far its result is from the converged one."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

OUT = "/tmp/contact_problems.npy"


def capture(E=24, steps=800, keep=100):
    import oracle
    from util_scenarios import ActionStream, grid_spawn
    N = 64
    pos, eul = grid_spawn(E, N)
    z = np.zeros((E, N, 3))
    sw = oracle.OracleSwarm(E, N, nthreads=1)
    sw.set_state(pos=pos.astype(np.float64), euler=eul, vel=z, angvel=z)
    acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
    lib = oracle.lib()
    lib.orc_contact_dump.argtypes = [C.POINTER(C.c_double), C.c_long]
    lib.orc_contact_dump_count.restype = C.c_long
    cap = E * N * keep
    buf = np.zeros((cap, 17))
    a = None
    for t in range(steps):
        if t % 50 == 0:
            a = acts(t)
        if t == steps - keep:
            lib.orc_contact_dump(buf.ctypes.data_as(C.POINTER(C.c_double)), cap)
        sw.step(a, "set_target_vel")
    n = lib.orc_contact_dump_count()
    lib.orc_contact_dump(None, 0)
    np.save(OUT, buf[:n])
    print("captured %d contact problems (%.1f %% of the agent-steps of the last %d steps)" % (n, 100.0 * n / cap, keep))


class Prob:
    """The captured problems as arrays (M = number of problems), default parameters of the library."""
    mass, r, hl = 0.027, 0.06, 0.0125
    dt, g, gz, mu, erp, thr = 0.01, 9.81, 0.5, 0.75, 0.2, 0.02

    def __init__(self, d):
        lx, lz = 2 * (self.r + 0.002), 2 * (self.hl + 0.002)
        I0 = self.mass / 12.0 * (lx * lx + lz * lz)
        I2 = self.mass / 12.0 * (2 * lx * lx)
        self.z = d[:, 0]
        self.R = d[:, 1:10].reshape(-1, 3, 3)
        self.v = d[:, 10:13].copy()
        self.w = d[:, 13:16].copy()
        M = len(d)
        Ib = np.array([1 / I0, 1 / I0, 1 / I2])
        self.Iw = np.einsum("mij,j,mkj->mik", self.R, Ib, self.R)
        c = self.r * 0.70710678118654752440
        sg = np.where(self.R[:, 2, 2] >= 0, -1.0, 1.0)
        pts = []
        for k in range(4):
            pb = np.stack([np.full(M, -c if (k & 1) else c), np.full(M, -c if (k & 2) else c), sg * self.hl], -1)
            pts.append(np.einsum("mij,mj->mi", self.R, pb))
        self.rk = np.stack(pts, 1)                                   # (M,4,3)
        self.dist = self.z[:, None] + self.rk[:, :, 2] - self.gz
        self.act = self.dist <= self.thr
        vn = self.v[:, None, 2] + self.w[:, None, 0] * self.rk[:, :, 1] - self.w[:, None, 1] * self.rk[:, :, 0]
        self.rhs = -vn - self.dist * np.where(self.dist > 0, 1 / self.dt, self.erp / self.dt)
        self.im = 1 / self.mass
        self.M = M


def rows_per_point(P):
    """the 12 rows of the shipped model: per point normal z, friction x, friction y; returns (dirs, levers)"""
    return None


def pgs12(P, iters, start="equal", order="point", tol=1e-7, stag=0.5, check_every=2, ret_hist=False, vel_tol=None, vel_abs=None):
    """The shipped model (oracle/mrs_oracle.c contact_solve): 4 points x (normal, friction x, friction y)."""
    M = P.M
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    ln = np.zeros((M, 4)); lt = np.zeros((M, 4, 2))
    r = P.rk; act = P.act
    E3 = np.eye(3)
    def ang(k, d):   # Iw (r_k x d)
        return np.einsum("mij,mj->mi", P.Iw, np.cross(r[:, k], np.broadcast_to(d, (M, 3))))
    an = [ang(k, E3[2]) for k in range(4)]; ax = [ang(k, E3[0]) for k in range(4)]; ay = [ang(k, E3[1]) for k in range(4)]
    def keff(k, d, a):
        u = np.cross(r[:, k], np.broadcast_to(d, (M, 3)))
        return 1.0 / (P.im + np.einsum("mi,mi->m", u, a))
    Kn = [np.where(act[:, k], keff(k, E3[2], an[k]), 0.0) for k in range(4)]
    Kx = [np.where(act[:, k], keff(k, E3[0], ax[k]), 0.0) for k in range(4)]
    Ky = [np.where(act[:, k], keff(k, E3[1], ay[k]), 0.0) for k in range(4)]
    if start == "equal":
        nact = act.sum(1)
        rsum = np.where(act, P.rhs, 0).sum(1)
        l0 = np.where(nact > 0, P.mass * np.maximum(rsum, 0) / np.maximum(nact, 1) ** 2, 0.0)
        for k in range(4):
            l = np.where(act[:, k], l0, 0.0)
            ln[:, k] = l
            dv[:, 2] += l * P.im
            dw += an[k] * l[:, None]
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int)
    prev = np.full(M, 3e38); moved = np.zeros(M)
    hist = []
    for it in range(iters):
        if it % check_every == 0:
            moved = np.zeros(M)
        if vel_tol is not None and it % check_every == check_every - 1:
            dv_before, dw_before = dv.copy(), dw.copy()          # the velocity rule looks at what the pair's second sweep changed
        live = ~done
        for k in range(4):
            rk = r[:, k]
            dvn = dv[:, 2] + dw[:, 0] * rk[:, 1] - dw[:, 1] * rk[:, 0]
            nl = np.maximum(ln[:, k] + Kn[k] * (P.rhs[:, k] - dvn), 0)
            dl = np.where(live, nl - ln[:, k], 0); ln[:, k] += dl
            moved = np.maximum(moved, np.abs(dl))
            dv[:, 2] += dl * P.im; dw += an[k] * dl[:, None]
            lim = P.mu * ln[:, k]
            for a, (K, A, d) in enumerate(((Kx, ax, 0), (Ky, ay, 1))):
                vtot = P.v + dv; wtot = P.w + dw
                vt = (vtot + np.cross(wtot, rk))[:, d]
                nl = np.clip(lt[:, k, a] - K[k] * vt, -lim, lim)
                dl = np.where(live, nl - lt[:, k, a], 0); lt[:, k, a] += dl
                moved = np.maximum(moved, np.abs(dl))
                dv[:, d] += dl * P.im; dw += A[k] * dl[:, None]
        used += live
        if ret_hist:
            hist.append((dv.copy(), dw.copy()))
        if it % check_every == check_every - 1:
            lmax = ln.max(1)
            if vel_tol is not None:     # stop on the body's VELOCITY change (m/s; rotation at the rim) instead of the impulses'
                dvel = np.maximum(np.abs(dv - dv_before).max(1), P.r * np.abs(dw - dw_before).max(1))
                scale = np.maximum(P.g * P.dt, np.maximum(np.abs(dv).max(1), P.r * np.abs(dw).max(1)))
                if vel_abs is not None:   # absolute on every velocity word (m/s, rad/s) + relative to the word sizes of the solve's change
                    dvel = np.maximum(np.abs(dv - dv_before).max(1), np.abs(dw - dw_before).max(1))
                    scale = vel_abs + vel_tol * np.maximum(np.abs(dv).max(1), np.abs(dw).max(1))
                    done |= dvel <= scale
                else:
                    done |= dvel <= vel_tol * scale
            else:
                conv = moved <= np.maximum(tolv, tol * lmax)
                stg = moved >= stag * prev
                done |= conv | stg
            prev = moved.copy()
    return dv, dw, used, (ln, lt), hist


def report(name, used, dv, dw, ref):
    e = np.maximum(np.abs(dv - ref[0]).max(1), 0.06 * np.abs(dw - ref[1]).max(1))
    h = np.bincount(used, minlength=11)
    wave = []
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(used))
    for i in range(0, len(used) - 63, 64):
        wave.append(used[perm[i:i + 64]].max())
    print("%-34s sweeps/body mean %.2f  hist %s  per-wave-of-64 mean %.2f | err vs converged: median %.1e  99%% %.1e  max %.1e"
          % (name, used.mean(), dict((i, int(c)) for i, c in enumerate(h) if c), np.mean(wave), np.median(e), np.quantile(e, 0.99), e.max()))


def study():
    d = np.load(OUT)
    P = Prob(d)
    anyact = P.act.any(1)
    print("%d problems, %d with an active point; active counts %s" % (P.M, anyact.sum(), np.bincount(P.act.sum(1))))
    d = d[anyact]
    P = Prob(d)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    dv, dw, used, _, _ = pgs12(P, 10)
    report("shipped: equal start, <=10, stag .5", used, dv, dw, ref)
    for it in (2, 4, 6):
        dv, dw, used, _, _ = pgs12(P, it)
        report("shipped, <=%d sweeps" % it, used, dv, dw, ref)
    dv, dw, used, _, _ = pgs12(P, 10, start="cold")
    report("cold start, <=10", used, dv, dw, ref)
    return P, ref


def flip_study():
    """What a stopping decision that falls the other way costs (float32 sweeps against float64 ones decide differently now and
    then): per body, the distance between its result at the sweep count the rule stops it at and the result one pair of sweeps
    later.  The impulse rules (tolerance + stagnation) against a rule on the body's velocity change."""
    d = np.load(OUT)
    P = Prob(d)
    d = d[P.act.any(1)]
    P = Prob(d)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    hist = pgs12(P, 14, tol=0, stag=2.0, ret_hist=True)[4]        # every body's iterate after 1..14 sweeps (no stopping)
    def consequence(used):
        idx = np.arange(P.M)
        a = np.stack([h[0] for h in hist]); b = np.stack([h[1] for h in hist])      # (14, M, 3)
        at = lambda k: (a[k - 1, idx], b[k - 1, idx])
        v0, w0 = at(used); v1, w1 = at(np.minimum(used + 2, 14))
        return np.maximum(np.abs(v1 - v0).max(1), np.abs(w1 - w0).max(1))
    for name, kw in (("shipped: impulse tol 1e-7 + stagnation .5", {}), ("impulse tol only (no stagnation rule)", dict(stag=2.0)),
                     ("velocity rule 1e-6", dict(vel_tol=1e-6)), ("velocity rule 1e-5", dict(vel_tol=1e-5)), ("velocity rule 1e-4", dict(vel_tol=1e-4)),
                     ("abs 1e-6 + 1e-6 |dv,dw|", dict(vel_tol=1e-6, vel_abs=1e-6)), ("abs 1e-5 + 1e-6 |dv,dw|", dict(vel_tol=1e-6, vel_abs=1e-5)),
                     ("abs 1e-6 + 1e-5 |dv,dw|", dict(vel_tol=1e-5, vel_abs=1e-6)), ("abs 1e-5 + 1e-5 |dv,dw|", dict(vel_tol=1e-5, vel_abs=1e-5))):
        dv, dw, used, _, _ = pgs12(P, 10, **kw)
        report(name, used, dv, dw, ref)
        c = consequence(used)
        early = used < 10
        print("    a flipped decision moves the result (max over v, w) by: median %.1e  99%% %.1e  max %.1e   (bodies stopped before the cap: %d of %d; "
              "moved by more than 1e-5 / 1e-4 / 1e-3: %d / %d / %d)"
              % (np.median(c[early]), np.quantile(c[early], 0.99), c[early].max(), early.sum(), P.M, (c[early] > 1e-5).sum(), (c[early] > 1e-4).sum(),
                 (c[early] > 1e-3).sum()))


if len(sys.argv) > 1 and sys.argv[1] == "flip":
    flip_study()
    sys.exit(0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "capture":
        capture(*(int(x) for x in sys.argv[2:]))
    else:
        study()


def slow_anatomy():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    dv, dw, used, (ln, lt), _ = pgs12(P, 10)
    slow = used > 2
    nact = P.act.sum(1)
    tilt = np.degrees(np.arccos(np.clip(np.abs(P.R[:, 2, 2]), 0, 1)))
    wn = np.linalg.norm(P.w, axis=1); vxy = np.linalg.norm(P.v[:, :2], axis=1)
    print("slow %d of %d" % (slow.sum(), len(slow)))
    for name, x in (("nact", nact), ("tilt deg", tilt), ("|w|", wn), ("|v_xy|", vxy), ("v_z", P.v[:, 2]), ("z", P.z)):
        q = lambda a: " ".join("%9.3g" % v for v in np.quantile(a, [0, .1, .5, .9, 1]))
        print("  %-9s fast: %s | slow: %s" % (name, q(x[~slow]), q(x[slow])))
    print("  nact among slow:", np.bincount(nact[slow], minlength=5), " among fast:", np.bincount(nact[~slow], minlength=5))
    # friction saturation: how many tangential rows sit at the pyramid's clamp
    lim = P.mu * ln
    sat = (np.abs(lt) >= lim[:, :, None] * (1 - 1e-9)) & (lim[:, :, None] > 0)
    print("  saturated friction rows per body: fast mean %.2f, slow mean %.2f" % (sat.sum((1, 2))[~slow].mean(), sat.sum((1, 2))[slow].mean()))
    lifted = (ln == 0) & P.act
    print("  active points with zero normal impulse per body: fast mean %.2f, slow mean %.2f" % (lifted.sum(1)[~slow].mean(), lifted.sum(1)[slow].mean()))


if len(sys.argv) > 1 and sys.argv[1] == "slow":
    slow_anatomy()


def tol_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    for tol in (1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
        for stag in (0.5,):
            dv, dw, used, _, _ = pgs12(P, 10, tol=tol, stag=stag)
            report("tol %g stag %g" % (tol, stag), used, dv, dw, ref)
    for tol in (1e-5, 1e-4):
        dv, dw, used, _, _ = pgs12(P, 10, tol=tol, stag=0.5, check_every=1)
        report("tol %g, checked every sweep" % tol, used, dv, dw, ref)


if len(sys.argv) > 1 and sys.argv[1] == "tol":
    tol_study()


def pgs7(P, iters, tol=1e-7, stag=0.5, check_every=2, torsion=True, nsub=1, ln0=None, fblock=False):
    """Candidate: 4 normal rows at the rim points + friction x, y at the centroid of the active points + a torsional row
    about the vertical (limit mu * rho * sum of the normal impulses, rho = rms horizontal distance of the active points
    from their centroid).  nsub: sweeps over the normal rows per sweep over the friction rows."""
    M = P.M
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    ln = np.zeros((M, 4)); lf = np.zeros((M, 3))
    r = P.rk; act = P.act
    E3 = np.eye(3)
    nact = act.sum(1)
    rc = (np.where(act[:, :, None], r, 0).sum(1)) / np.maximum(nact, 1)[:, None]
    rho = np.sqrt(np.maximum((np.where(act, ((r[:, :, :2] - rc[:, None, :2]) ** 2).sum(2), 0).sum(1)) / np.maximum(nact, 1), 0))
    def ang(rr, d):
        return np.einsum("mij,mj->mi", P.Iw, np.cross(rr, np.broadcast_to(d, (M, 3))))
    an = [ang(r[:, k], E3[2]) for k in range(4)]
    Kn = [np.where(act[:, k], 1.0 / (P.im + np.einsum("mi,mi->m", np.cross(r[:, k], np.broadcast_to(E3[2], (M, 3))), an[k])), 0.0) for k in range(4)]
    af = [ang(rc, E3[0]), ang(rc, E3[1]), P.Iw[:, :, 2]]
    Kf = [1.0 / (P.im + np.einsum("mi,mi->m", np.cross(rc, np.broadcast_to(E3[0], (M, 3))), af[0])),
          1.0 / (P.im + np.einsum("mi,mi->m", np.cross(rc, np.broadcast_to(E3[1], (M, 3))), af[1])),
          1.0 / P.Iw[:, 2, 2]]
    rsum = np.where(act, P.rhs, 0).sum(1)
    l0 = np.where(nact > 0, P.mass * np.maximum(rsum, 0) / np.maximum(nact, 1) ** 2, 0.0)
    for k in range(4):
        l = np.where(act[:, k], l0, 0.0) if ln0 is None else np.maximum(np.where(act[:, k], ln0[:, k], 0), 0)
        ln[:, k] = l; dv[:, 2] += l * P.im; dw += an[k] * l[:, None]
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int); prev = np.full(M, 3e38); moved = np.zeros(M)
    # friction block: the three rows (x, y at the centroid, torsion) solved together, then clamped
    Jf = np.zeros((M, 3, 6))                                    # rows acting on (v, w)
    Jf[:, 0, 0] = 1; Jf[:, 0, 3:] = np.cross(rc, np.broadcast_to(E3[0], (M, 3)))
    Jf[:, 1, 1] = 1; Jf[:, 1, 3:] = np.cross(rc, np.broadcast_to(E3[1], (M, 3)))
    Jf[:, 2, 5] = 1
    Minv6 = np.zeros((M, 6, 6)); Minv6[:, 0, 0] = Minv6[:, 1, 1] = Minv6[:, 2, 2] = P.im; Minv6[:, 3:, 3:] = P.Iw
    Wf = np.einsum("mai,mij,mbj->mab", Jf, Minv6, Jf)
    Wfi = np.linalg.inv(Wf)
    for it in range(iters):
        if it % check_every == 0:
            moved = np.zeros(M)
        live = ~done
        for sub in range(nsub):
            for k in range(4):
                rk = r[:, k]
                dvn = dv[:, 2] + dw[:, 0] * rk[:, 1] - dw[:, 1] * rk[:, 0]
                nl = np.maximum(ln[:, k] + Kn[k] * (P.rhs[:, k] - dvn), 0)
                dl = np.where(live, nl - ln[:, k], 0); ln[:, k] += dl
                moved = np.maximum(moved, np.abs(dl))
                dv[:, 2] += dl * P.im; dw += an[k] * dl[:, None]
        N = ln.sum(1)
        if fblock:
            u = np.concatenate([P.v + dv, P.w + dw], 1)
            vt = np.einsum("mai,mi->ma", Jf, u)
            nl = lf - np.einsum("mab,mb->ma", Wfi, vt)
            lim = np.stack([P.mu * N, P.mu * N, P.mu * N * rho], 1)
            nl = np.clip(nl, -lim, lim)
            dl = np.where(live[:, None], nl - lf, 0); lf += dl
            moved = np.maximum(moved, np.abs(dl[:, :2]).max(1)); moved = np.maximum(moved, np.abs(dl[:, 2]) / 0.06)
            du = np.einsum("mij,maj,ma->mi", Minv6, Jf, dl)
            dv += du[:, :3]; dw += du[:, 3:]
        for a in range(2 if not fblock else 0):
            vt = (P.v + dv + np.cross(P.w + dw, rc))[:, a]
            lim = P.mu * N
            nl = np.clip(lf[:, a] - Kf[a] * vt, -lim, lim)
            dl = np.where(live, nl - lf[:, a], 0); lf[:, a] += dl
            moved = np.maximum(moved, np.abs(dl))
            dv[:, a] += dl * P.im; dw += af[a] * dl[:, None]
        if torsion and not fblock:
            wz = (P.w + dw)[:, 2]
            lim = P.mu * N * rho
            nl = np.clip(lf[:, 2] - Kf[2] * wz, -lim, lim)
            dl = np.where(live, nl - lf[:, 2], 0); lf[:, 2] += dl
            moved = np.maximum(moved, np.abs(dl) / 0.06)
            dw += af[2] * dl[:, None]
        used += live
        if it % check_every == check_every - 1:
            conv = moved <= np.maximum(tolv, tol * ln.max(1))
            done |= conv | (moved >= stag * prev)
            prev = moved.copy()
    return dv, dw, used, (ln, lf), None


def b_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    ref7 = pgs7(P, 600, tol=0, stag=2.0)[:2]
    ref12 = pgs12(P, 400, tol=0, stag=2.0)[:2]
    e = np.maximum(np.abs(ref7[0] - ref12[0]).max(1), 0.06 * np.abs(ref7[1] - ref12[1]).max(1))
    print("converged 7-row vs converged 12-row model: median %.1e 90%% %.1e 99%% %.1e max %.1e" % (np.median(e), np.quantile(e, .9), np.quantile(e, .99), e.max()))
    for tol in (1e-7, 1e-5):
        dv, dw, used, _, _ = pgs7(P, 10, tol=tol)
        report("7 rows, tol %g" % tol, used, dv, dw, ref7)
    dv, dw, used, _, _ = pgs7(P, 10, tol=1e-6, nsub=2)
    report("7 rows, 2 normal passes per sweep", used, dv, dw, ref7)
    dv, dw, used, _, _ = pgs7(P, 10, tol=1e-6, check_every=1)
    report("7 rows, tol 1e-6 every sweep", used, dv, dw, ref7)


if len(sys.argv) > 1 and sys.argv[1] == "b":
    b_study()


def why_slow():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    mu = Prob.mu
    for m in (0.0, 0.75, 1e9):
        Prob.mu = m; P.mu = m
        ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
        dv, dw, used, _, _ = pgs12(P, 10, stag=2.0)
        report("12 rows, mu = %g, no stagnation stop" % m, used, dv, dw, ref)
    Prob.mu = mu


if len(sys.argv) > 1 and sys.argv[1] == "why":
    why_slow()


def fit_start(P):
    """normal impulses of the unconstrained least-squares solution over the active points (pseudo-inverse of W_AA), clipped at 0"""
    M = P.M
    r = P.rk
    a = np.stack([np.ones((M, 4)), r[:, :, 1], -r[:, :, 0]], -1)            # (M,4,3): a_k = (1, ry, -rx)
    Minv = np.zeros((M, 3, 3)); Minv[:, 0, 0] = P.im; Minv[:, 1:, 1:] = P.Iw[:, :2, :2]
    W = np.einsum("mki,mij,mlj->mkl", a, Minv, a)
    act = P.act
    W = np.where(act[:, :, None] & act[:, None, :], W, 0)
    b = np.where(act, P.rhs, 0)
    lam = np.zeros((M, 4))
    for m0 in range(0, M, 4096):
        sl = slice(m0, m0 + 4096)
        lam[sl] = np.einsum("mkl,ml->mk", np.linalg.pinv(W[sl], rcond=1e-9, hermitian=True), b[sl])
    return lam


def pgs12_from(P, ln0, iters, tol=1e-7, stag=0.5, mu=None):
    """pgs12 with a given start of the normal impulses (monkey-patches the equal-share start)"""
    M = P.M
    mu = P.mu if mu is None else mu
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    ln = np.maximum(np.where(P.act, ln0, 0), 0); lt = np.zeros((M, 4, 2))
    r = P.rk; act = P.act; E3 = np.eye(3)
    def ang(k, d):
        return np.einsum("mij,mj->mi", P.Iw, np.cross(r[:, k], np.broadcast_to(d, (M, 3))))
    an = [ang(k, E3[2]) for k in range(4)]; ax = [ang(k, E3[0]) for k in range(4)]; ay = [ang(k, E3[1]) for k in range(4)]
    def keff(k, d, a):
        return 1.0 / (P.im + np.einsum("mi,mi->m", np.cross(r[:, k], np.broadcast_to(d, (M, 3))), a))
    Kn = [np.where(act[:, k], keff(k, E3[2], an[k]), 0.0) for k in range(4)]
    Kx = [np.where(act[:, k], keff(k, E3[0], ax[k]), 0.0) for k in range(4)]
    Ky = [np.where(act[:, k], keff(k, E3[1], ay[k]), 0.0) for k in range(4)]
    for k in range(4):
        dv[:, 2] += ln[:, k] * P.im; dw += an[k] * ln[:, k, None]
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int); prev = np.full(M, 3e38); moved = np.zeros(M)
    for it in range(iters):
        if it % 2 == 0:
            moved = np.zeros(M)
        live = ~done
        for k in range(4):
            rk = r[:, k]
            dvn = dv[:, 2] + dw[:, 0] * rk[:, 1] - dw[:, 1] * rk[:, 0]
            nl = np.maximum(ln[:, k] + Kn[k] * (P.rhs[:, k] - dvn), 0)
            dl = np.where(live, nl - ln[:, k], 0); ln[:, k] += dl
            moved = np.maximum(moved, np.abs(dl))
            dv[:, 2] += dl * P.im; dw += an[k] * dl[:, None]
            lim = mu * ln[:, k]
            for a, (K, A, d) in enumerate(((Kx, ax, 0), (Ky, ay, 1))):
                vt = (P.v + dv + np.cross(P.w + dw, rk))[:, d]
                nl = np.clip(lt[:, k, a] - K[k] * vt, -lim, lim)
                dl = np.where(live, nl - lt[:, k, a], 0); lt[:, k, a] += dl
                moved = np.maximum(moved, np.abs(dl))
                dv[:, d] += dl * P.im; dw += A[k] * dl[:, None]
        used += live
        if it % 2 == 1:
            conv = moved <= np.maximum(tolv, tol * ln.max(1))
            done |= conv | (moved >= stag * prev)
            prev = moved.copy()
    return dv, dw, used


def fit_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    l0 = fit_start(P)
    print("fit start: bodies with a negative component: %d of %d" % ((np.where(P.act, l0, 0) < -1e-12).any(1).sum(), P.M))
    for mu in (0.0, 0.75):
        P.mu = mu
        ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
        dv, dw, used = pgs12_from(P, l0, 10, mu=mu)
        report("least-squares start, mu %g" % mu, used, dv, dw, ref)
        dv, dw, used = pgs12_from(P, l0, 10, mu=mu, tol=1e-5)
        report("least-squares start, tol 1e-5, mu %g" % mu, used, dv, dw, ref)
    P.mu = 0.75


if len(sys.argv) > 1 and sys.argv[1] == "fit":
    fit_study()


def combo_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    l0 = fit_start(P)
    ref7 = pgs7(P, 600, tol=0, stag=2.0)[:2]
    for kw in (dict(), dict(ln0=l0), dict(ln0=l0, fblock=True), dict(fblock=True), dict(ln0=l0, fblock=True, tol=1e-5), dict(ln0=l0, fblock=True, nsub=2)):
        dv, dw, used, _, _ = pgs7(P, 10, **kw)
        report("7 rows %s" % ",".join("%s" % k for k in kw), used, dv, dw, ref7)


if len(sys.argv) > 1 and sys.argv[1] == "combo":
    combo_study()


def bgs12(P, iters, tol=1e-7, stag=0.5, ln0=None):
    """block Gauss-Seidel: the three rows of a point solved together (3x3), then normal >= 0 and the pyramid clamp"""
    M = P.M
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    lam = np.zeros((M, 4, 3))                                    # (x, y, z) impulse per point
    r = P.rk; act = P.act
    def skew(a):
        S = np.zeros(a.shape[:-1] + (3, 3))
        S[..., 0, 1] = -a[..., 2]; S[..., 0, 2] = a[..., 1]; S[..., 1, 0] = a[..., 2]; S[..., 1, 2] = -a[..., 0]; S[..., 2, 0] = -a[..., 1]; S[..., 2, 1] = a[..., 0]
        return S
    Kinv = []
    for k in range(4):
        S = skew(r[:, k])
        K = P.im * np.eye(3)[None] - np.einsum("mij,mjk,mkl->mil", S, P.Iw, S)
        Kinv.append(np.linalg.inv(K))
    nact = act.sum(1); rsum = np.where(act, P.rhs, 0).sum(1)
    l0 = np.where(nact > 0, P.mass * np.maximum(rsum, 0) / np.maximum(nact, 1) ** 2, 0.0)
    for k in range(4):
        l = np.where(act[:, k], l0, 0.0) if ln0 is None else np.maximum(np.where(act[:, k], ln0[:, k], 0), 0)
        lam[:, k, 2] = l
        dv[:, 2] += l * P.im; dw += np.einsum("mij,mj->mi", P.Iw, np.cross(r[:, k], np.stack([0 * l, 0 * l, l], -1)))
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int); prev = np.full(M, 3e38); moved = np.zeros(M)
    vn0 = P.v[:, None, 2] + P.w[:, None, 0] * r[:, :, 1] - P.w[:, None, 1] * r[:, :, 0]
    for it in range(iters):
        if it % 2 == 0:
            moved = np.zeros(M)
        live = ~done
        for k in range(4):
            rk = r[:, k]
            pv = P.v + dv + np.cross(P.w + dw, rk)                # current point velocity
            tgt = np.zeros((M, 3)); tgt[:, 2] = vn0[:, k] + P.rhs[:, k]
            nl = lam[:, k] + np.einsum("mij,mj->mi", Kinv[k], tgt - pv)
            nl[:, 2] = np.maximum(nl[:, 2], 0)
            lim = P.mu * nl[:, 2]
            nl[:, 0] = np.clip(nl[:, 0], -lim, lim); nl[:, 1] = np.clip(nl[:, 1], -lim, lim)
            dl = np.where((live & act[:, k])[:, None], nl - lam[:, k], 0); lam[:, k] += dl
            moved = np.maximum(moved, np.abs(dl).max(1))
            dv += dl * P.im; dw += np.einsum("mij,mj->mi", P.Iw, np.cross(rk, dl))
        used += live
        if it % 2 == 1:
            conv = moved <= np.maximum(tolv, tol * lam[:, :, 2].max(1))
            done |= conv | (moved >= stag * prev)
            prev = moved.copy()
    return dv, dw, used


def bgs_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    refb = bgs12(P, 400, tol=0, stag=2.0)[:2]
    e = np.maximum(np.abs(refb[0] - ref[0]).max(1), 0.06 * np.abs(refb[1] - ref[1]).max(1))
    print("converged block-GS vs converged row-GS: median %.1e 99%% %.1e max %.1e" % (np.median(e), np.quantile(e, .99), e.max()))
    for kw in (dict(), dict(tol=1e-5), dict(ln0=fit_start(P))):
        dv, dw, used = bgs12(P, 10, **kw)
        report("block GS %s" % ",".join(kw), used, dv, dw, refb)
    # how good is each after a fixed number of sweeps, measured against the SAME converged solution (row-GS)?
    for n in (2, 4):
        dv, dw, used = bgs12(P, n, stag=2.0, tol=0)
        report("block GS, exactly %d sweeps (vs row ref)" % n, used, dv, dw, ref)
        dv, dw, used, _, _ = pgs12(P, n, stag=2.0, tol=0)
        report("row GS, exactly %d sweeps (vs row ref)" % n, used, dv, dw, ref)


if len(sys.argv) > 1 and sys.argv[1] == "bgs":
    bgs_study()


def rest_predictor():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    dv, dw, used, _, _ = pgs12(P, 10)
    slow = used > 2
    tilt = 1 - np.abs(P.R[:, 2, 2]); wn = np.abs(P.w).max(1); vxy = np.abs(P.v[:, :2]).max(1); n4 = P.act.all(1)
    for thr in (1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4):
        rest = n4 & (tilt < thr) & (wn < thr) & (vxy < thr)
        print("thr %g: predicted at rest %d (%.1f %%), of which slow %d; not at rest %d of which slow %d" %
              (thr, rest.sum(), 100.0 * rest.mean(), (rest & slow).sum(), (~rest).sum(), (~rest & slow).sum()))
    # at-rest closed form: every point carries rhs_k * m / 4 ... error against the solver's result
    rest = n4 & (tilt < 1e-7) & (wn < 1e-7) & (vxy < 1e-7)
    cf_dvz = P.rhs[rest].mean(1)
    print("closed form for bodies at rest: max |dv_z - solver| %.2e, max |dv_xy| %.2e, max |dw| %.2e" %
          (np.abs(cf_dvz - dv[rest, 2]).max(), np.abs(dv[rest, :2]).max(), np.abs(dw[rest]).max()))


if len(sys.argv) > 1 and sys.argv[1] == "rest":
    rest_predictor()


def pgs12_order(P, iters, order="point", tol=1e-7, stag=0.5):
    """pgs12 with other row orders: 'point' (shipped), 'sym' (points 0..3 then 3..0 on alternate sweeps), 'nf' (the four
    normal rows, then the eight friction rows), 'nfsym' (both)"""
    M = P.M
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    ln = np.zeros((M, 4)); lt = np.zeros((M, 4, 2))
    r = P.rk; act = P.act; E3 = np.eye(3)
    def ang(k, d):
        return np.einsum("mij,mj->mi", P.Iw, np.cross(r[:, k], np.broadcast_to(d, (M, 3))))
    an = [ang(k, E3[2]) for k in range(4)]; ax = [ang(k, E3[0]) for k in range(4)]; ay = [ang(k, E3[1]) for k in range(4)]
    def keff(k, d, a):
        return 1.0 / (P.im + np.einsum("mi,mi->m", np.cross(r[:, k], np.broadcast_to(d, (M, 3))), a))
    Kn = [np.where(act[:, k], keff(k, E3[2], an[k]), 0.0) for k in range(4)]
    Kt = [[np.where(act[:, k], keff(k, E3[0], ax[k]), 0.0), np.where(act[:, k], keff(k, E3[1], ay[k]), 0.0)] for k in range(4)]
    At = [[ax[k], ay[k]] for k in range(4)]
    nact = act.sum(1); rsum = np.where(act, P.rhs, 0).sum(1)
    l0 = np.where(nact > 0, P.mass * np.maximum(rsum, 0) / np.maximum(nact, 1) ** 2, 0.0)
    for k in range(4):
        l = np.where(act[:, k], l0, 0.0); ln[:, k] = l; dv[:, 2] += l * P.im; dw += an[k] * l[:, None]
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int); prev = np.full(M, 3e38); moved = np.zeros(M)
    state = dict(moved=moved)
    def nrow(k, live):
        rk = r[:, k]
        dvn = dv[:, 2] + dw[:, 0] * rk[:, 1] - dw[:, 1] * rk[:, 0]
        nl = np.maximum(ln[:, k] + Kn[k] * (P.rhs[:, k] - dvn), 0)
        dl = np.where(live, nl - ln[:, k], 0); ln[:, k] += dl
        state["moved"] = np.maximum(state["moved"], np.abs(dl))
        dv[:, 2] += dl * P.im; dw[:] += an[k] * dl[:, None]
    def frow(k, a, live):
        rk = r[:, k]
        vt = (P.v + dv + np.cross(P.w + dw, rk))[:, a]
        lim = P.mu * ln[:, k]
        nl = np.clip(lt[:, k, a] - Kt[k][a] * vt, -lim, lim)
        dl = np.where(live, nl - lt[:, k, a], 0); lt[:, k, a] += dl
        state["moved"] = np.maximum(state["moved"], np.abs(dl))
        dv[:, a] += dl * P.im; dw[:] += At[k][a] * dl[:, None]
    for it in range(iters):
        if it % 2 == 0:
            state["moved"] = np.zeros(M)
        live = ~done
        pts = range(4) if (order in ("point", "nf") or it % 2 == 0) else range(3, -1, -1)
        if order in ("point", "sym"):
            for k in pts:
                nrow(k, live); frow(k, 0, live); frow(k, 1, live)
        else:
            for k in pts:
                nrow(k, live)
            for k in pts:
                frow(k, 0, live); frow(k, 1, live)
        used += live
        if it % 2 == 1:
            mv = state["moved"]
            done |= (mv <= np.maximum(tolv, tol * ln.max(1))) | (mv >= stag * prev)
            prev = mv.copy()
    return dv, dw, used


def order_study():
    d = np.load(OUT); P = Prob(d); d = d[P.act.any(1)]; P = Prob(d)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    for order in ("point", "sym", "nf", "nfsym"):
        for n in (4, 6, 10):
            dv, dw, used = pgs12_order(P, n, order=order, stag=2.0, tol=0)
            report("order %-6s exactly %2d sweeps" % (order, n), used, dv, dw, ref)


if len(sys.argv) > 1 and sys.argv[1] == "order":
    order_study()


def warm_study():
    """Warm start across steps (what Bullet does): the impulses a body's rows ended the previous step with are the start of
    this step's sweeps.  Bodies are matched across the captured steps by identity; a body without a problem in the step
    before starts from the equal share."""
    d = np.load(OUT)
    ids = d[:, 16]
    # split into steps: the capture is in step order, ids ascending within a step
    brk = np.flatnonzero(np.diff(ids) < 0) + 1
    steps = np.split(np.arange(len(d)), brk)
    print("%d steps captured" % len(steps))
    prev = {}
    hist_w = np.zeros(12, int); hist_c = np.zeros(12, int); errs_w = []; errs_c = []
    for si, idx in enumerate(steps):
        P = Prob(d[idx]); act = P.act.any(1)
        if not act.any():
            prev = {}; continue
        sub = idx[act]; P = Prob(d[sub])
        ref = pgs12(P, 300, tol=0, stag=2.0)
        l0n = np.zeros((P.M, 4)); l0t = np.zeros((P.M, 4, 2)); have = np.zeros(P.M, bool)
        for m, b in enumerate(d[sub, 16]):
            if b in prev:
                l0n[m], l0t[m] = prev[b]; have[m] = True
        dvw, dww, usedw, lam = pgs12_warm(P, 10, l0n, l0t, have)
        dvc, dwc, usedc, _, _ = pgs12(P, 10)
        prev = {b: (lam[0][m].copy(), lam[1][m].copy()) for m, b in enumerate(d[sub, 16])}
        if si >= 5:
            hist_w += np.bincount(usedw, minlength=12)[:12]; hist_c += np.bincount(usedc, minlength=12)[:12]
            errs_w.append(np.maximum(np.abs(dvw - ref[0]).max(1), 0.06 * np.abs(dww - ref[1]).max(1)))
            errs_c.append(np.maximum(np.abs(dvc - ref[0]).max(1), 0.06 * np.abs(dwc - ref[1]).max(1)))
    ew, ec = np.concatenate(errs_w), np.concatenate(errs_c)
    print("cold (equal share)  sweeps hist %s  err 99%% %.1e 99.9%% %.1e" % (dict((i, int(c)) for i, c in enumerate(hist_c) if c), np.quantile(ec, .99), np.quantile(ec, .999)))
    print("warm start          sweeps hist %s  err 99%% %.1e 99.9%% %.1e" % (dict((i, int(c)) for i, c in enumerate(hist_w) if c), np.quantile(ew, .99), np.quantile(ew, .999)))


def pgs12_warm(P, iters, l0n, l0t, have, tol=1e-7, stag=0.5):
    M = P.M
    dv = np.zeros((M, 3)); dw = np.zeros((M, 3))
    r = P.rk; act = P.act; E3 = np.eye(3)
    def ang(k, d):
        return np.einsum("mij,mj->mi", P.Iw, np.cross(r[:, k], np.broadcast_to(d, (M, 3))))
    an = [ang(k, E3[2]) for k in range(4)]; ax = [ang(k, E3[0]) for k in range(4)]; ay = [ang(k, E3[1]) for k in range(4)]
    def keff(k, d, a):
        return 1.0 / (P.im + np.einsum("mi,mi->m", np.cross(r[:, k], np.broadcast_to(d, (M, 3))), a))
    Kn = [np.where(act[:, k], keff(k, E3[2], an[k]), 0.0) for k in range(4)]
    Kt = [[np.where(act[:, k], keff(k, E3[0], ax[k]), 0.0), np.where(act[:, k], keff(k, E3[1], ay[k]), 0.0)] for k in range(4)]
    At = [[ax[k], ay[k]] for k in range(4)]
    nact = act.sum(1); rsum = np.where(act, P.rhs, 0).sum(1)
    eq = np.where(nact > 0, P.mass * np.maximum(rsum, 0) / np.maximum(nact, 1) ** 2, 0.0)
    ln = np.where(have[:, None], l0n, eq[:, None]) * act
    lt = np.where(have[:, None, None], l0t, 0.0) * act[:, :, None]
    for k in range(4):
        dv[:, 2] += ln[:, k] * P.im; dw += an[k] * ln[:, k, None]
        for a in range(2):
            dv[:, a] += lt[:, k, a] * P.im; dw += At[k][a] * lt[:, k, a, None]
    tolv = tol * P.mass * P.g * P.dt
    done = np.zeros(M, bool); used = np.zeros(M, int); prev = np.full(M, 3e38); moved = np.zeros(M)
    for it in range(iters):
        if it % 2 == 0:
            moved = np.zeros(M)
        live = ~done
        for k in range(4):
            rk = r[:, k]
            dvn = dv[:, 2] + dw[:, 0] * rk[:, 1] - dw[:, 1] * rk[:, 0]
            nl = np.maximum(ln[:, k] + Kn[k] * (P.rhs[:, k] - dvn), 0)
            dl = np.where(live, nl - ln[:, k], 0); ln[:, k] += dl
            moved = np.maximum(moved, np.abs(dl)); dv[:, 2] += dl * P.im; dw += an[k] * dl[:, None]
            lim = P.mu * ln[:, k]
            for a in range(2):
                vt = (P.v + dv + np.cross(P.w + dw, rk))[:, a]
                nl = np.clip(lt[:, k, a] - Kt[k][a] * vt, -lim, lim)
                dl = np.where(live, nl - lt[:, k, a], 0); lt[:, k, a] += dl
                moved = np.maximum(moved, np.abs(dl)); dv[:, a] += dl * P.im; dw += At[k][a] * dl[:, None]
        used += live
        if it % 2 == 1:
            done |= (moved <= np.maximum(tolv, tol * ln.max(1))) | (moved >= stag * prev)
            prev = moved.copy()
    return dv, dw, used, (ln, lt)


if len(sys.argv) > 1 and sys.argv[1] == "warm":
    warm_study()


def table():
    """What DESIGN.md section 5 quotes (profiles/r03_contact_lab.txt)."""
    d = np.load(OUT); P = Prob(d)
    print("# tools/contact_lab.py table: %d contact problems of the bench workload on the CPU oracle (24 envs x 64 agents, steps 700-800)" % P.M)
    print("active rim points per problem: %s" % dict(enumerate(np.bincount(P.act.sum(1), minlength=5).tolist())))
    d = d[P.act.any(1)]; P = Prob(d)
    tilt = np.maximum(np.abs(P.R[:, 2, 0]), np.abs(P.R[:, 2, 1])); wn = np.abs(P.w).max(1); vxy = np.abs(P.v[:, :2]).max(1)
    for eps in (1e-9, 1e-6, 1e-5):
        rest = P.act.all(1) & (tilt < eps) & (wn < eps) & (vxy < eps)
        print("at rest (|R20|, |R21|, |w|, |v_xy| < %g, four active points): %.1f %% of the problems" % (eps, 100.0 * rest.mean()))
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    print("error of the velocity change against the converged solution (400 sweeps), max(|dv|, 0.06 |dw|) in m/s, and sweeps run:")
    for cap in (2, 4, 6, 8, 10):
        dv, dw, used, _, _ = pgs12(P, cap)
        e = np.maximum(np.abs(dv - ref[0]).max(1), 0.06 * np.abs(dw - ref[1]).max(1))
        print("  at most %2d sweeps: mean sweeps %.2f | error median %.1e  90 %% %.1e  99 %% %.1e  99.9 %% %.1e  max %.1e" % (
            cap, used.mean(), np.median(e), np.quantile(e, .9), np.quantile(e, .99), np.quantile(e, .999), e.max()))
    dv, dw, used, _, _ = pgs12(P, 10)
    rest = P.act.all(1) & (tilt < 1e-6) & (wn < 1e-6) & (vxy < 1e-6)
    print("sweeps wanted under the cap of 10 (convergence or stagnation): all %s; not at rest %s" % (
        dict((i, int(c)) for i, c in enumerate(np.bincount(used)) if c), dict((i, int(c)) for i, c in enumerate(np.bincount(used[~rest])) if c)))


if len(sys.argv) > 1 and sys.argv[1] == "table":
    table()


def ls_study():
    """Bodies standing on all four rim points with friction holding ("sticking"): the rows then ask every rim point for the velocity
    (0, 0, u_k) -- twelve equations, six unknowns -- and what the sweeps converge to (over hundreds of sweeps; the impulses keep
    drifting in their null space) is the LEAST-SQUARES velocity, which has a closed form because the four points are symmetric
    about the cap's centre:  v_centre = (0, 0, mean u),  w_body = diag(1/4c^2, 1/4c^2, 1/8c^2) (sum_k u_k s_k) x z_body.
    How many of the listed bodies is that, and how do the capped sweeps compare with it?"""
    d = np.load(OUT)
    P = Prob(d)
    d = d[P.act.any(1)]
    P = Prob(d)
    M = P.M
    four = P.act.all(1)
    c = P.r * 0.70710678118654752440
    R = P.R
    sg = np.where(R[:, 2, 2] >= 0, -1.0, 1.0)
    u = -P.dist * np.where(P.dist > 0, 1 / P.dt, P.erp / P.dt)                     # target normal velocity per point
    sb = np.array([[c if not (k & 1) else -c, c if not (k & 2) else -c, 0.0] for k in range(4)])    # s_k in the body frame
    zb = R[:, 2, :]                                                                # world z in body coordinates
    S = np.einsum("mk,kj->mj", u, sb)
    wb = np.cross(S, zb) * np.array([1 / (4 * c * c), 1 / (4 * c * c), 1 / (8 * c * c)])
    w_ls = np.einsum("mij,mj->mi", R, wb)
    rbar = np.einsum("mij,mj->mi", R, np.stack([np.zeros(M), np.zeros(M), sg * P.hl], -1))
    v_ls = np.stack([np.zeros(M), np.zeros(M), u.mean(1)], -1) - np.cross(w_ls, rbar)
    # impulse the body needs for it, and a sufficient test that four contact forces inside their friction pyramids can deliver it
    lx, lz = 2 * (P.r + 0.002), 2 * (P.hl + 0.002)
    I0 = P.mass / 12.0 * (lx * lx + lz * lz); I2 = P.mass / 12.0 * (2 * lx * lx)
    Ib = np.array([I0, I0, I2])
    dP = P.mass * (v_ls - P.v)
    dLb = Ib * np.einsum("mji,mj->mi", R, (w_ls - P.w))
    dL = np.einsum("mij,mj->mi", R, dLb)
    Mc = dL - np.cross(rbar, dP)                                                   # moment about the cap centre
    fm = 0.25 * np.maximum(np.abs(dP[:, 0]), np.abs(dP[:, 1])) + np.abs(Mc[:, 2]) * (0.25 / P.r)
    s = 2 * P.r * (0.25 * dP[:, 2] - fm / P.mu)
    ok = four & (s > 0) & ((Mc[:, 0] ** 2 + Mc[:, 1] ** 2) * 1.05 < s * s)
    ref = pgs12(P, 400, tol=0, stag=2.0)[:2]
    dv10, dw10, used, _, _ = pgs12(P, 10)
    tilt = np.hypot(R[:, 2, 0], R[:, 2, 1])
    print("%d listed bodies; all four points active: %d; of those the sufficient test passes: %d" % (M, four.sum(), ok.sum()))
    e_ls = np.maximum(np.abs(v_ls - (P.v + ref[0])).max(1), np.abs(w_ls - (P.w + ref[1])).max(1))
    e_10 = np.maximum(np.abs(dv10 - ref[0]).max(1), np.abs(dw10 - ref[1]).max(1))
    q = lambda x: "median %.1e  99%% %.1e  max %.1e" % (np.median(x), np.quantile(x, 0.99), x.max()) if len(x) else "-"
    print("  closed form against 400 sweeps, bodies that pass:   " + q(e_ls[ok]))
    print("  capped sweeps (10) against 400 sweeps, same bodies:  " + q(e_10[ok]))
    print("  tilt of the bodies that pass: " + q(tilt[ok]))
    print("  sweeps used now, bodies that pass:   %s" % dict(zip(*np.unique(used[ok], return_counts=True))))
    print("  sweeps used now, bodies that remain: %s" % dict(zip(*np.unique(used[~ok], return_counts=True))))
    print("  four points active but the test fails: %d; their closed-form error: %s" % ((four & ~ok).sum(), q(e_ls[four & ~ok])))
    rng = np.random.default_rng(0)
    for name, m in (("now", np.ones(M, bool)), ("with the closed form", ~ok)):
        # solver waves of 5 listed bodies (a workgroup's share at the bench size): the sweeps the wave runs = its slowest body's
        idx = rng.permutation(np.where(m)[0]); k = int(5 * m.mean() + 0.5) or 1
        groups = [used[idx[i:i + k]].max() for i in range(0, len(idx) - k + 1, k)]
        print("  %-22s listed per workgroup %.1f -> sweeps per solver wave: mean %.2f, at the cap %.0f %%" % (name, 5 * m.mean(), np.mean(groups), 100 * np.mean(np.array(groups) >= 10)))


if len(sys.argv) > 1 and sys.argv[1] == "ls":
    ls_study()
