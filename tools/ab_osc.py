#!/usr/bin/env python3
"""A/B of builds of the extension on configs[2] (not a test): per library, wall time of the OSC Env.step with random targets and of
the scripted standing controller at 65 536 envs.   usage: python tools/ab_osc.py libA.so [libB.so ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, numpy as np
sys.path.insert(0, %r)
import torch
from cassierl_amd import rollout as R
from cassierl_amd.vec_env import CassieVecEnv
n = 65536
env = CassieVecEnv(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
env.use_torch_stream()
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
lo, hi = np.array([-2, -2, -2, 0, -2, 0, -2.0]), np.full(7, 2.0)
acts = [R.random_actions(4, ids, t, lo, hi) for t in range(40)]
for t in range(10): env.step(acts[t], out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(10, 40): env.step(acts[t], out)
torch.cuda.synchronize(); ms_step = (time.perf_counter() - t0) / 30 * 1e3
zp = torch.full((n,), 0.9, dtype=torch.float64, device="cuda"); zv = torch.zeros(n, dtype=torch.float64, device="cuda")
env2 = CassieVecEnv(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
env2.use_torch_stream()
env2._chk(env2.L.CassieVecStandingStep(env2.h, 2, zp.data_ptr(), zv.data_ptr(), 20))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): env2._chk(env2.L.CassieVecStandingStep(env2.h, 2, zp.data_ptr(), zv.data_ptr(), 20))
torch.cuda.synchronize(); ms_sub = (time.perf_counter() - t0) / 200 * 1e3
q, v = env2.get_state_host()
print("AB " + json.dumps(dict(env_step_ms=ms_step, env_steps_per_s=n / ms_step * 1e3, standing_substep_ms=ms_sub, standing_substeps_per_s=n / ms_sub * 1e3, z=float(q[:, 1].mean()))))
''' % ROOT
for lib in sys.argv[1:]:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CASSIE2D_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
    print(lib, line[0][3:] if line else "FAILED " + p.stderr[-600:])
