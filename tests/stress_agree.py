#!/usr/bin/env python3
"""One-off stress check (not a pytest file): the 4-envs-per-wave kernels (+ clean-up pass) against the wave-per-environment
kernels on thousands of environments, teacher-forced every Env.step, all control modes.  Prints the worst disagreement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cassierl_amd.vec_env import CassieVecEnv, WAVE_PER_ENV  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4099
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = np.load(os.path.join(ROOT, "tests", "golden", "traj2d.npz"))
rng = np.random.default_rng(0)
for kind, mode in (("walk", "PD"), ("stand", "Torque"), ("stand", "OSC"), ("stand", "Jacobian")):
    a = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    b = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV)
    for e in (a, b):
        e.set_trajectory(d["time"], d["qpos"])
    a.reset_host(); b.reset_host()
    lo, hi = a.action_space.low, a.action_space.high
    if mode == "OSC":
        lo, hi = np.array([-6, -6, -2, 0, -2, 0, -6.0]), np.array([6, 6, 2, 2, 2, 2, 6.0])
    if mode == "Jacobian":
        lo, hi = np.array([-60.0, 0.0, -25.0] * 2), np.array([60.0, 250.0, 25.0] * 2)
    worst, ndone, bad = 0.0, 0, 0
    for t in range(steps):
        acts = rng.uniform(lo, hi, (n, a.adim))
        b.set_full_state_host(a.get_full_state_host())
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        err = np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / (1.0 + np.abs(sb[:, :26]).max(axis=1))
        worst = max(worst, float(err.max()), float(np.abs(oa - ob).max()), float(np.abs(ra - rb).max()))
        bad += int((da != db).sum()) + int((~np.isfinite(sa)).any())
        ndone += int(da.sum())
    print("%-5s %-8s n=%d steps=%d  worst=%.2e  done-mismatch/nonfinite=%d  episodes=%d" % (kind, mode, n, steps, worst, bad, ndone), flush=True)
    a.close(); b.close()
