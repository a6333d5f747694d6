"""The leading open risk of SURVEY.md rows I/G, as an experiment (CPU oracle only; no GPU, no reference import).

    python tools/probes/readme_stability.py            -> profiles/r03_readme_stability.txt

The reference's README example (README.md:14-33): N_AGENTS = 3, ACTION_TYPE = set_target_vel with the constant target
[0.5, 0, 0], default START_POS (z in [1, 3]) and default START_ORI (roll = pitch = 0, yaw ~ U[-pi/2, pi/2], MRS.py:54).
On the oracle's restatement of the Bullet integrator a part of those quadcopters does NOT hold the target velocity: the
attitude loop goes unstable and the body ends on the ground.  The cause is inertia-independent (all three candidates of
SURVEY.md row I below): QuadControl.attitude_control feeds the WORLD-frame angular velocity into the body-rate D term
(Quadcopter.py:54 hands over get_angvel()), which is only right at yaw = 0.  Either the real reference (with pybullet)
crashes its own README demo for a part of its spawns, or the restated angular dynamics (damping form, gyroscopic term,
integration order -- all [BULLET-KNOWLEDGE], pybullet is absent) differ from Bullet's.  This cannot be settled here;
the table is what a pybullet-equipped machine has to compare against.

Survival = the share of quadcopters that are still above z = 0.8 after 1000 steps, over 64 envs x 3 agents per cell."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import oracle

E, N, STEPS = 64, 3, 1000
MASS, R, HL = 0.027, 0.06, 0.0125


def inertia_candidates():
    lx, lz = 2 * (R + 0.002), 2 * (HL + 0.002)
    aabb = (MASS / 12 * (lx * lx + lz * lz), MASS / 12 * (lx * lx + lz * lz), MASS / 12 * (2 * lx * lx))
    cyl = (MASS * (3 * R * R + (2 * HL) ** 2) / 12, MASS * (3 * R * R + (2 * HL) ** 2) / 12, MASS * R * R / 2)
    return [("hull AABB + margins (default)", aabb), ("URDF file (cf2x.urdf:12)", (1.4e-5, 1.4e-5, 2.17e-5)), ("implicit cylinder", cyl)]


def survival(yaw_max, inertia, ang_damp, use_gyro, seed=0):
    rng = np.random.default_rng(seed)
    # default spawn (MRS.py:69-78): xy ~ N(0,1) pulled into the unit disc, z ~ U[1,3]; agents kept >= 0.6 apart by resampling
    pos = np.zeros((E, N, 3))
    for e in range(E):
        while True:
            xy = rng.normal(size=(N, 2)); n = np.linalg.norm(xy, axis=1, keepdims=True); xy = np.where(n > 1, xy / n, xy)
            p = np.concatenate([xy, rng.uniform(1, 3, (N, 1))], 1)
            d = np.linalg.norm(p[:, None] - p[None], axis=-1) + 10 * np.eye(N)
            if d.min() >= 0.6:
                pos[e] = p; break
    eul = np.zeros((E, N, 3), np.float32); eul[..., 2] = rng.uniform(-yaw_max, yaw_max, (E, N))
    prm = oracle.default_params()
    prm.inertia[0], prm.inertia[1], prm.inertia[2] = inertia
    prm.ang_damp = ang_damp; prm.use_gyro = use_gyro
    sw = oracle.OracleSwarm(E, N, params=prm, nthreads=8)
    z = np.zeros((E, N, 3))
    sw.set_state(pos=pos, euler=eul, vel=z, angvel=z)
    act = np.broadcast_to(np.array([0.5, 0, 0], np.float32), (E, N, 3)).copy()
    for _ in range(STEPS):
        sw.step(act, "set_target_vel")
    alive = sw.pos[..., 2] > 0.8
    on_target = alive & (np.abs(sw.vel[..., 0] - 0.5) < 0.05)
    return float(alive.mean()), float(on_target.mean())


def main():
    out = []
    out.append("# tools/probes/readme_stability.py: README.md:14-33 on the CPU oracle, %d envs x %d agents, %d steps; survival (z > 0.8) / holding v_x = 0.5 +- 0.05" % (E, N, STEPS))
    out.append("%-32s %-9s %-5s | %-17s %-17s %-17s" % ("inertia", "ang_damp", "gyro", "|yaw| <= pi/2", "|yaw| <= 1.2", "|yaw| <= 0.8"))
    for name, I in inertia_candidates():
        for ad in (0.04, 0.0):
            for gy in (1, 0):
                cells = []
                for ym in (np.pi / 2, 1.2, 0.8):
                    a, t = survival(ym, I, ad, gy)
                    cells.append("%.2f / %.2f" % (a, t))
                out.append("%-32s %-9g %-5d | %-17s %-17s %-17s" % (name, ad, gy, *cells))
                print(out[-1], flush=True)
    path = os.path.join(ROOT, "profiles", "r03_readme_stability.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
