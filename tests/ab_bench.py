#!/usr/bin/env python3
"""A/B timing of alternative builds of the HIP extension (not a test): python tests/ab_bench.py lib1.so lib2.so ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from cassierl_amd.vec_env import CassieVecEnv
d = np.load(os.path.join(%r, "tests", "golden", "traj2d.npz"))
rng = np.random.default_rng(0)
lo, hi = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
for n in (4096, 65536):
    env = CassieVecEnv(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    env.set_trajectory(d["time"], d["qpos"])
    out = env.alloc(); env.reset(out)
    a = torch.as_tensor(rng.uniform(lo, hi, size=(n, 6)), device="cuda")
    env.time_steps(a, 5, out)
    ms = min(env.time_steps(a, 20, out) for _ in range(3))
    q, v = env.get_state_host()
    print("  n=%%6d  %%.3f ms/step  %%.3f M env-steps/s  finite=%%s" %% (n, ms, n / ms / 1e3, np.isfinite(q).all()))
    env.close()
''' % (ROOT, ROOT)
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ, CASSIE2D_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", CODE], env=env)
