#!/usr/bin/env python3
"""Per-phase error quantiles of the teacher-forced runs (tests/util_teacher.py) on the literal BASELINE workloads:
`python tools/teacher_probe.py [--steps 1000] [--envs 32] [--solver-iters K]` on the GPU box -> the table of DESIGN.md section 5
(profiles/r04_teacher_forced.txt)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mrs-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import mrsgym_amd  # noqa: E402
import util_teacher as ut  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--configs", default="C2,C3,C4,C5")
    ap.add_argument("--solver-iters", type=int, default=0)
    ap.add_argument("--adj-every", type=int, default=1, help="compare adjacency / observation with the oracle every k-th step (large --envs)")
    ap.add_argument("--dump", default="", help="write the worst contact-phase cases (pre-state, both post-states, wrench) to this .npz")
    a = ap.parse_args()
    print("teacher-forced per-step error |GPU - oracle| (max over the 13 state words, relative above magnitude 1),")
    print("oracle re-seeded from the GPU state every step; E = %d envs, %d steps; phases by the body's pre-step state" % (a.envs, a.steps))
    for cfg in a.configs.split(","):
        params = None
        if a.solver_iters:
            params = mrsgym_amd.default_params()
            params.solver_iters = a.solver_iters
        t0 = time.time()
        dump = {"thr": float(os.environ.get("DUMP_THR", "5e-5")), "max": 400} if a.dump else None
        if dump and os.environ.get("DUMP_VMAX"):
            dump["vmax"] = float(os.environ["DUMP_VMAX"])
        r = ut.run(torch, mrsgym_amd, cfg, E=a.envs, steps=a.steps, params=params, dump=dump, unconstrained=True, check_adj_every=a.adj_every,
                   progress=lambda t: print("   ... %s step %d (%.0f s)" % (cfg, t, time.time() - t0), flush=True))
        print("%s  N=%d  %s  solver_iters=%d  grounded at the end %.0f %%  adjacency/observation mismatches %d  visited %s"
              % (cfg, r["N"], ut.CONFIGS[cfg]["atype"], int((params or mrsgym_amd.default_params()).solver_iters),
                 100 * r["grounded_share"], r["adj_bad"], json.dumps(r["visited"])))
        for ph in ut.PHASES:
            q = ut.quantiles(r["err"][ph])
            if q:
                qa = ut.quantiles(r["abs_err"][ph])
                print("   %-7s n=%9d  50%% %.2e  99%% %.2e  99.9%% %.2e  max %.2e  at (t, env, agent) %s   ABSOLUTE: 99%% %.2e  99.9%% %.2e  max %.2e"
                      % (ph, q["n"], q["q50"], q["q99"], q["q999"], q["max"], r["worst"][ph][1], qa["q99"], qa["q999"], qa["max"]))
                if ph != "free":        # split by what the contact solve was handed (the oracle's unconstrained velocity, largest word)
                    import numpy as np
                    x, vu = r["err"][ph], r["vunc"][ph]
                    for lo, hi in ((0, 5), (5, 50), (50, 1e9)):
                        m = (vu >= lo) & (vu < hi)
                        if m.any():
                            print("             unconstrained velocity in [%g, %g) m/s: n=%9d  99.9%% %.2e  max %.2e" % (lo, min(hi, 100), m.sum(), np.quantile(x[m], 0.999), x[m].max()))
        if dump and dump.get("cases"):
            import numpy as np
            c = dump["cases"]
            np.savez(a.dump.replace(".npz", "_%s.npz" % cfg), **{k: np.array([x[k] for x in c]) for k in c[0]})
        sys.stdout.flush()


if __name__ == "__main__":
    main()
