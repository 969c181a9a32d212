#!/usr/bin/env python3
"""Per-substep deviation of the HIP path from the oracle, teacher-forced (every substep starts from the oracle's state), PD and
torque mode, 1500 substeps each (not a test; the library is the one CASSIE2D_LIB points to, default the in-tree build).
r02: max 3.2e-14 / p99 2.1e-14 / median 3.9e-15 (PD), max 2.4e-14 / median 1.4e-15 (torque) -- the perturbation size the
shadowing test in tests/test_gpu_parity.py gives the oracle's twins."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_py as O
from cassierl_amd.vec_env import CassieVecEnv
PD_LO = np.array([-0.5, -1.0, 0.5, -1.0, 0.5, -1.0]); PD_HI = np.array([0.5, 0.0, 1.5, 0.0, 1.5, 0.0])
def state_vec(q, v, ws):
    s = np.zeros(88); s[:13] = q; s[13:26] = v; s[26:39] = ws; s[39:52] = q; s[52:65] = v; return s
rng = np.random.default_rng(5)
n = 4
for mode in ("PD", "Torque"):
    env = CassieVecEnv(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False)
    o = O.Oracle()
    errs = []
    for i in range(1500):
        a = rng.uniform(PD_LO, PD_HI) if mode == "PD" else rng.uniform(-1, 1, 6) * 20
        q, v = o.state()
        env.set_full_state_host(np.tile(state_vec(q, v, o.warmstart()), (n, 1)))
        env.substep_host(mode, np.tile(a, (n, 1)), 1)
        (o.step_pd if mode == "PD" else o.step_torque)(a)
        sg = env.get_full_state_host()
        q1, v1 = o.state()
        errs.append(max(np.abs(sg[0, :13] - q1).max(), np.abs(sg[0, 13:26] - v1).max() / (1 + np.abs(v1).max())))
    errs = np.array(errs)
    print(os.path.basename(os.environ.get("CASSIE2D_LIB", "default")), mode, "max %.2e  p99 %.2e  median %.2e" % (errs.max(), np.percentile(errs, 99), np.median(errs)))
    env.close()
