#!/usr/bin/env python3
"""A/B timing of kernel variants (not a test).  python tests/ab_bench.py [lib.so ...]; runs each with CASSIE2D_G16=1 and 0."""
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
for kind, mode, n in (("walk", "PD", 4096), ("walk", "PD", 65536), ("stand", "Torque", 4096), ("stand", "Torque", 65536)):
    env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    env.set_trajectory(d["time"], d["qpos"])
    out = env.alloc(); env.reset(out)
    if mode == "PD":
        a = torch.as_tensor(rng.uniform(lo, hi, size=(n, 6)), device="cuda")
    else:
        a = torch.as_tensor(rng.uniform(-1, 1, size=(n, 6)) * np.array([12, 12, .9] * 2), device="cuda")
    env.time_steps(a, 30 if mode == "Torque" else 5, out)
    ms = min(env.time_steps(a, 20, out) for _ in range(3))
    q, v = env.get_state_host()
    print("  %%-5s %%-6s n=%%6d  %%.3f ms/step  %%.3f M env-steps/s  finite=%%s  zmean=%%.3f" %% (kind, mode, n, ms, n / ms / 1e3, np.isfinite(q).all(), q[:, 1].mean()))
    env.close()
''' % (ROOT, ROOT)
libs = sys.argv[1:] or [os.path.join(ROOT, "cassierl_amd", "lib", "libcassie2d.so")]
for lib in libs:
    for g16 in ("1", "0"):
        print(lib, "CASSIE2D_G16=" + g16, flush=True)
        env = dict(os.environ, CASSIE2D_LIB=os.path.abspath(lib), CASSIE2D_G16=g16)
        subprocess.run([sys.executable, "-c", CODE], env=env)
